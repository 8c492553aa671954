"""torch.autograd wrappers around the C-ABI kernels of libmonkeynet_hip.so.

Internal activation format ("act"): a contiguous fp32 tensor of shape (N, H, W, ld) -- NHWC with the reference's
time axis D folded into N (frame = b*D + d) and the channel stride `ld` rounded up to a multiple of 4 (pad channels
are zero).  The logical channel count travels next to the tensor.  PyTorch is used for memory, streams and the
autograd tape only; every arithmetic step is a kernel launch through `mnk._lib` (no CPU / eager fallback).
"""
import weakref

import numpy as np
import torch

from . import _lib
from . import knobs
from . import dist as mdist


def ceil4(c):
    return (c + 3) // 4 * 4


def ceil16(c):
    return (c + 15) // 16 * 16


def _p(t):
    return None if t is None else t.data_ptr()


# (the raw handle of the current stream without a torch.cuda.Stream object: 0.3 instead of 5 us per kernel launch -- the
# reference's own loop on these modules is bound by the host, not by the GPU: tools/dropin_profile.py)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    if not t.is_cuda:
        return 0
    if _RAW_STREAM is not None:
        return _RAW_STREAM(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


def _check_device(t):
    if not t.is_cuda and _lib.lib().is_device_build:
        raise _lib.MnkError("the MI355X hot path needs CUDA/HIP tensors; got a %s tensor (there is no CPU fallback)"
                            % t.device)


class _Scratch:
    """Grow-only scratch buffers, one per (purpose, device, stream): reuse between consecutive kernels of a stream is
    ordered by that stream, and a branch of work on a second stream gets buffers of its own."""

    def __init__(self):
        self.bufs = {}

    def get(self, key, nfloats, like):
        nfloats = max(int(nfloats), 1)
        k = (key, like.device, _stream(like))
        buf = self.bufs.get(k)
        if buf is None or buf.numel() < nfloats:
            buf = torch.empty(int(nfloats * 1.25) + 1024, dtype=torch.float32, device=like.device)
            self.bufs[k] = buf
        return buf


SCRATCH = _Scratch()


def _call(name, ref, *args):
    _lib.lib().call(name, *args, _stream(ref))


def _query(name, *args):
    return _lib.lib().query(name, *args)


class _EvalCtx:
    """What a Function.forward of this module sees as `ctx` when it runs OUTSIDE autograd (grad mode off): nothing is saved,
    no input needs a gradient; attributes a forward sets (ctx.meta = ...) land on this throw-away object."""
    __slots__ = ("__dict__", "needs_input_grad")

    def __init__(self, n):
        self.needs_input_grad = (False,) * n

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


class _Fn(torch.autograd.Function):
    """torch.autograd.Function whose apply() calls forward() directly when grad mode is off: the evaluation loops of the
    reference (reconstruction.py:52-62, transfer.py:65-79, demo.py) run the networks frame by frame at batch 1 under
    torch.no_grad(), ~95 Function calls per frame, and are bound by the host (tools/frame_loop_probe.py) -- the autograd
    machinery of apply() (argument scan, node construction, output wrapping) has nothing to record there."""

    @classmethod
    def apply(cls, *args):
        if torch.is_grad_enabled():
            return super().apply(*args)
        return cls.forward(_EvalCtx(len(args)), *args)


# ----------------------------------------------------------------------------------------------------------------
# layout
# ----------------------------------------------------------------------------------------------------------------
class ToActFn(_Fn):
    """(B,C,D,H,W) -> act, with the nearest down-scaling by an integer `step` (keypoint_detector.py:98-99)."""

    @staticmethod
    def forward(ctx, x5, step):
        _check_device(x5)
        x5 = x5.contiguous().float()
        b, c, d, h, w = x5.shape
        out = torch.empty(b * d, h // step, w // step, ceil4(c), dtype=torch.float32, device=x5.device)
        _call("mnk_ncdhw_to_nhwc", x5, _p(x5), _p(out), b, c, d, h, w, step, ceil4(c))
        ctx.meta = (b, c, d, h, w, step)
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, d, h, w, step = ctx.meta
        if step != 1:
            raise NotImplementedError("gradient w.r.t. a down-scaled input image is never consumed by the reference")
        g = g.contiguous()
        out = torch.empty(b, c, d, h, w, dtype=torch.float32, device=g.device)
        _call("mnk_nhwc_to_ncdhw", g, _p(g), g.shape[-1], _p(out), b, c, d, h, w)
        return out, None


class StackedBatch:
    """[a ; b] along the batch axis of two (B,C,D,H,W) tensors WITHOUT the concatenation: the only consumer of such a stack in
    the training iteration is the conversion to the kernels' layout (train.py:27 / :43-45 -> to_act), which writes the two
    halves of one act itself (ToActPairFn) -- torch.cat of contiguous tensors along dim 0 is two device-to-device memcpy
    launches in front of that conversion."""

    def __init__(self, a, b):
        assert a.shape == b.shape and a.dim() == 5
        self.parts = (a, b)
        self.shape = torch.Size((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]))
        self.device = a.device

    def __getitem__(self, idx):
        b = self.parts[0].shape[0]
        if isinstance(idx, slice) and idx.start in (None, 0) and idx.step is None and idx.stop == b:
            return self.parts[0]
        return torch.cat(self.parts, dim=0)[idx]


class ToActPairFn(_Fn):
    """ToActFn of StackedBatch(a, b): one act, its two halves written by one conversion launch each."""

    @staticmethod
    def forward(ctx, a5, b5, step):
        _check_device(a5)
        a5, b5 = a5.contiguous().float(), b5.contiguous().float()
        b, c, d, h, w = a5.shape
        out = torch.empty(2 * b * d, h // step, w // step, ceil4(c), dtype=torch.float32, device=a5.device)
        for i, src in enumerate((a5, b5)):
            _call("mnk_ncdhw_to_nhwc", src, _p(src), _p(out[i * b * d:]), b, c, d, h, w, step, ceil4(c))
        ctx.meta = (b, c, d, h, w, step)
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, d, h, w, step = ctx.meta
        if step != 1:
            raise NotImplementedError("gradient w.r.t. a down-scaled input image is never consumed by the reference")
        g = g.contiguous()
        outs = [None, None]
        for i in range(2):
            if ctx.needs_input_grad[i]:
                outs[i] = torch.empty(b, c, d, h, w, dtype=torch.float32, device=g.device)
                _call("mnk_nhwc_to_ncdhw", g, _p(g[i * b * d:]), g.shape[-1], _p(outs[i]), b, c, d, h, w)
        return outs[0], outs[1], None


class FromActFn(_Fn):
    """act -> (B,C,D,H,W)."""

    @staticmethod
    def forward(ctx, a, c, b):
        n, h, w, ld = a.shape
        d = n // b
        out = torch.empty(b, c, d, h, w, dtype=torch.float32, device=a.device)
        _call("mnk_nhwc_to_ncdhw", a, _p(a), ld, _p(out), b, c, d, h, w)
        ctx.meta = (b, c, d, h, w, ld)
        return out

    @staticmethod
    def backward(ctx, g):
        b, c, d, h, w, ld = ctx.meta
        g = g.contiguous()
        out = torch.empty(b * d, h, w, ld, dtype=torch.float32, device=g.device)
        _call("mnk_ncdhw_to_nhwc", g, _p(g), _p(out), b, c, d, h, w, 1, ld)
        return out, None, None


class Concat2PairFn(_Fn):
    """Concat2Fn for the batched discriminator pass [generated | real] (mnk.engine.fused_pair_losses): `a` holds 2B frames, `b`
    (the key-point embedding, the same for both halves) B frames: out[n] = [a[n] | b[n mod B]].  The reference's two calls
    embed the same key points twice (train.py:43-45); here the embedding is made once, and its gradient is the sum of the two
    halves -- no concatenated key-point tensors, none of their backward additions."""

    @staticmethod
    def forward(ctx, a, ca, b, cb):
        n, h, w, _ = a.shape
        assert b.shape[0] * 2 == n and b.shape[1:3] == a.shape[1:3]
        out = torch.empty(n, h, w, ceil4(ca + cb), dtype=torch.float32, device=a.device)
        _call("mnk_concat2_fwd", a, _p(a), a.shape[-1], ca, _p(b), b.shape[-1], cb, n // 2, _p(out), out.shape[-1], n, h * w)
        ctx.meta = (ca, cb, a.shape[-1], b.shape[-1])
        return out

    @staticmethod
    def backward(ctx, g):
        ca, cb, lda, ldb = ctx.meta
        g = g.contiguous()
        n, h, w, ld = g.shape
        ga = torch.empty(n, h, w, lda, dtype=torch.float32, device=g.device)
        gb = torch.empty(n // 2, h, w, ldb, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
        _call("mnk_concat2_bwd", g, _p(g), ld, ca, cb, n // 2, _p(ga), lda, _p(gb), ldb, n, h * w)
        return ga, None, gb, None


class Concat2Fn(_Fn):
    """torch.cat([a, b], dim=channel) on acts (modules/util.py:185 for the last decoder stage)."""

    @staticmethod
    def forward(ctx, a, ca, b, cb):
        n, h, w, _ = a.shape
        out = torch.empty(n, h, w, ceil4(ca + cb), dtype=torch.float32, device=a.device)       # one launch, pads included
        _call("mnk_concat2_fwd", a, _p(a), a.shape[-1], ca, _p(b), b.shape[-1], cb, n, _p(out), out.shape[-1], n, h * w)
        ctx.meta = (ca, cb, a.shape[-1], b.shape[-1])
        return out

    @staticmethod
    def backward(ctx, g):
        ca, cb, lda, ldb = ctx.meta
        g = g.contiguous()
        n, h, w, ld = g.shape
        ga = torch.empty(n, h, w, lda, dtype=torch.float32, device=g.device)
        gb = torch.empty(n, h, w, ldb, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[2] else None
        _call("mnk_concat2_bwd", g, _p(g), ld, ca, cb, n, _p(ga), lda, _p(gb), ldb, n, h * w)
        return ga, None, gb, None


# ----------------------------------------------------------------------------------------------------------------
# 3x3 convolution
# ----------------------------------------------------------------------------------------------------------------
_PACK_CACHE = {}
_PACK_EPOCH = [0]
_PACK_REG = {}                     # id(parameter) -> _PackEntry
_PACK_TABLE = {"dirty": True, "descs": None, "n": 0, "tiles": 0, "entries": (), "keep": []}
_PACK_REG_VERSION = [0]            # bumped whenever an entry (or one of its buffers) is created or dropped


def invalidate_packed_weights():
    """Parameters were written behind the library's back (`p.data.copy_`, a custom update): drop every packed copy --
    the inference-side cache (keyed by this epoch) and the freshness of the training-side registry.  Optimiser steps do
    this by themselves (hook below), for the parameters of that optimiser."""
    _PACK_EPOCH[0] += 1
    for e in _PACK_REG.values():
        e.stamp = None


def bump_inference_epoch():
    """the no-grad cache only (a replayed training iteration re-packs its own weights on the device)"""
    _PACK_EPOCH[0] += 1


def _after_optimizer_step(opt, args, kwargs):
    """Global post-step hook of torch.optim: the stepped optimiser's parameters changed (fused optimisers do not bump
    `Tensor._version`, so the version check alone cannot see it).  An optimiser that wrote the packed layouts itself
    (mnk.optim.MnkAdam) hands its entries over to be stamped fresh."""
    _PACK_EPOCH[0] += 1
    # hand-overs of a backward pass that nobody picked up (a gradient that autograd summed with another one first) hold GPU
    # tensors: a user-owned loop -- the reference's train.py on these modules -- never calls clear_dz_stats()
    _DZ_STATS.clear()
    for g in opt.param_groups:
        for p in g["params"]:
            e = _PACK_REG.get(id(p))
            if e is not None:
                e.stamp = None
    fresh = getattr(opt, "_mnk_fresh_entries", None)
    if fresh:
        for e in fresh:
            w = e.wref()
            if w is not None and e.wptr == w.data_ptr():
                e.stamp = w._version
        opt._mnk_fresh_entries = ()


try:  # any torch optimiser step anywhere invalidates what it touched
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_hook
    _reg_hook(_after_optimizer_step)
except Exception:  # pragma: no cover - older torch: TrainStep still invalidates explicitly
    pass


# ---- gradient sinks: parameters whose gradient a kernel writes straight into an optimiser-owned flat buffer ------------
_SINKS = {}                        # id(parameter) -> weakref to the owning mnk.optim.MnkAdam


def register_grad_sink(p, owner):
    _SINKS[id(p)] = weakref.ref(owner)


def unregister_grad_sinks(ids, owner_id):
    for i in ids:
        r = _SINKS.get(i)
        if r is not None and (r() is None or id(r()) == owner_id):
            _SINKS.pop(i, None)


def sink_owner(p):
    r = _SINKS.get(id(p))
    return r() if r is not None else None


def pack_registry_version():
    return _PACK_REG_VERSION[0]


def pack_entry_of(p):
    """the training-side packed-weight entry of a 3x3 convolution weight, if a training forward has created one"""
    e = _PACK_REG.get(id(p))
    if e is not None and e.wref() is p and e.wptr == p.data_ptr():
        return e
    return None


def subpixel(ups):
    """UpBlock3D convolutions run in their sub-pixel forms (mnk_conv3x3_up_*) -- 13.87 -> 12.62 ms per step in round 2; the 3x3
    convolution over the up-sampled view stays for the operands the sub-pixel kernels do not take (a residual input) and, for
    A/B runs, behind the library's "up_subpixel" tuning value (MNK_TUNING=up_subpixel=0 or mnk_set_tuning), whose LIVE value
    this function reads (mnk_get_tuning: the library's weight-gradient plans follow the same integer)."""
    if not ups:
        return False
    # read once per library handle and per mnk_set_tuning call made through it (a ctypes round trip per up-sampled
    # convolution of every eager forward, and a value that could change between a forward and its backward)
    lib = _lib.lib()
    epoch = getattr(lib, "tuning_epoch", 0)
    hit = getattr(lib, "_subpixel_cache", None)
    if hit is None or hit[0] != epoch:
        import ctypes
        v = ctypes.c_int(1)
        lib.call("mnk_get_tuning", b"up_subpixel", ctypes.byref(v))
        hit = lib._subpixel_cache = (epoch, bool(v.value))
    return hit[1]


# mnk.dropin.EvalRunner captures an evaluation forward with FROZEN weights: the cached packed weights / evaluation-mode norm
# coefficients are used (and their addresses recorded) instead of re-made inside the graph; the runner re-captures when a
# parameter, a buffer or the optimiser epoch changes
FROZEN_CAPTURE = [False]


def _packed_fwd_weight(weight, cout, c0, c1, up=False):
    """Packed [Cout][chunk][tap][16] copy of a conv weight for NO-GRAD forwards (inference loops), cached per parameter
    version, storage and optimiser epoch.  Training forwards never use this cache (they take the registry entry of
    _pack_entry: re-packed once per iteration by repack_registered(), or per call in a user-owned loop).
    Under hipGraph capture the cache is bypassed: the pack launch is recorded INTO the graph (its buffer lives in the
    graph's memory pool), so every replay re-packs from the live parameter -- a replay after load_state_dict / an
    in-place write can never combine old packed weights with new normalisation parameters."""
    n = _query("mnk_conv3x3_up_packed_floats" if up else "mnk_conv3x3_packed_floats", cout, c0, c1)
    pack = "mnk_conv3x3_up_pack_fwd" if up else "mnk_conv3x3_pack_fwd"
    if weight.is_cuda and torch.cuda.is_current_stream_capturing() and not FROZEN_CAPTURE[0]:
        wp = torch.empty(n, dtype=torch.float32, device=weight.device)
        _call(pack, weight, _p(weight), _p(wp), cout, c0, c1)
        return wp
    key = (id(weight), weight.device)
    ver = (weight._version, _PACK_EPOCH[0], weight.data_ptr())
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0] == ver and hit[1] == (cout, c0, c1, up) and hit[3]() is weight:
        return hit[2]
    wp = torch.empty(n, dtype=torch.float32, device=weight.device)
    _call(pack, weight, _p(weight), _p(wp), cout, c0, c1)
    if hit is None:
        weakref.finalize(weight, _PACK_CACHE.pop, key, None)      # no entry (and no packed buffer) outlives its parameter
    _PACK_CACHE[key] = (ver, (cout, c0, c1, up), wp, weakref.ref(weight))
    return wp


class _PackEntry:
    """Persistent packed copies (forward + data-gradient layouts) of one conv parameter used by training forwards."""
    __slots__ = ("wref", "wptr", "meta", "wp", "wd", "stamp", "__weakref__")

    def fresh(self, weight, meta, need):
        return (self.wref() is weight and self.wptr == weight.data_ptr() and self.meta == meta
                and self.stamp is not None and self.stamp == weight._version
                and all(self.wd[i] is not None for i in (0, 1) if need[i]))



def _pack_entry(weight, cout, c0, c1, need, up=False):
    """The (packed, up to date) registry entry of `weight`: packs with one per-layer launch unless the entry is fresh
    (repack_registered() ran since the last optimiser step).  up: the packs of the sub-pixel forms of an up-sampled
    convolution (mnk_conv3x3_up_fwd / _up_dgrad) instead of the 3x3 layouts."""
    meta = (cout, c0, c1, bool(up))
    e = _PACK_REG.get(id(weight))
    if e is not None and e.fresh(weight, meta, need):
        return e
    if (e is not None and e.wref() is weight and e.wptr == weight.data_ptr() and e.meta == meta
            and all(e.wd[i] is not None for i in (0, 1) if need[i])):
        # a known layer gone stale (an optimiser stepped): the first such layer of an iteration re-packs EVERY registered
        # parameter in one launch -- a user-owned loop (the reference's train.py) then pays one pack launch per iteration
        # instead of one (to three) per convolution, as mnk.engine.TrainStep does at the start of its iterations
        if repack_registered() and e.fresh(weight, meta, need):
            return e
    if e is None or e.wref() is not weight or e.wptr != weight.data_ptr() or e.meta != meta:
        e = _PackEntry()
        e.wref, e.wptr, e.meta, e.wd, e.stamp = weakref.ref(weight), weight.data_ptr(), meta, [None, None], None
        e.wp = torch.empty(_query("mnk_conv3x3_up_packed_floats" if up else "mnk_conv3x3_packed_floats", cout, c0, c1),
                           dtype=torch.float32, device=weight.device)
        key = id(weight)
        _PACK_REG[key] = e
        weakref.finalize(weight, _drop_pack_entry, key, weakref.ref(e))
        _PACK_TABLE["dirty"] = True
        _PACK_REG_VERSION[0] += 1
    for i, cc in enumerate((c0, c1)):
        if need[i] and e.wd[i] is None:
            nd = _query("mnk_conv3x3_up_dgrad_packed_floats", cout, cc) if up else _query("mnk_conv3x3_packed_floats", cc, cout, 0)
            e.wd[i] = torch.empty(nd, dtype=torch.float32, device=weight.device)
            _PACK_TABLE["dirty"] = True
            _PACK_REG_VERSION[0] += 1
    if up:
        _call("mnk_conv3x3_up_pack_fwd", weight, _p(weight), _p(e.wp), cout, c0, c1)
        for i, (cs, cc) in enumerate(((0, c0), (c0, c1))):
            if e.wd[i] is not None:
                _call("mnk_conv3x3_up_pack_dgrad", weight, _p(weight), _p(e.wd[i]), cout, c0 + c1, cs, cc)
    else:
        _call("mnk_conv3x3_pack_all", weight, _p(weight), _p(e.wp), _p(e.wd[0]), _p(e.wd[1]), cout, c0, c1)
    e.stamp = None            # only repack_registered() vouches for freshness: a user-owned loop packs on every call
    return e


def _drop_pack_entry(key, eref):
    if _PACK_REG.get(key) is eref():
        _PACK_REG.pop(key, None)
        _PACK_TABLE["dirty"] = True
        _PACK_REG_VERSION[0] += 1


def repack_registered(only_if_stale=False):
    """Re-pack EVERY registered conv parameter in one launch (mnk_conv3x3_pack_multi) -- the start of a training
    iteration (mnk.engine.TrainStep): ~40 per-layer pack launches become one.  Parameters first seen later in the
    iteration (and everything, when called under stream capture before the table exists) fall back to per-layer packs.
    only_if_stale: nothing to do when every entry is still fresh (the optimiser kernel of mnk.optim wrote the packs)."""
    t = _PACK_TABLE
    if only_if_stale and _PACK_REG:
        stale = False
        for e in _PACK_REG.values():
            w = e.wref()
            if w is not None and (e.stamp is None or e.stamp != w._version or e.wptr != w.data_ptr()):
                stale = True
                break
        if not stale:
            return False
    if t["dirty"]:
        entries = [e for e in _PACK_REG.values() if e.wref() is not None and e.wptr == e.wref().data_ptr()]
        if not entries:
            return False
        dev = entries[0].wp.device
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return False                     # building the table needs a host-to-device copy
        entries = [e for e in entries if e.wp.device == dev]
        rec = np.zeros(len(entries), dtype=np.dtype([("p", "<u8", 4), ("i", "<i4", 6)]))
        tiles = 0
        for k, e in enumerate(entries):
            cout, c0, c1, up = e.meta
            rec["p"][k] = (e.wptr, e.wp.data_ptr(), e.wd[0].data_ptr() if e.wd[0] is not None else 0,
                           e.wd[1].data_ptr() if e.wd[1] is not None else 0)
            rec["i"][k] = (cout, c0, c1, tiles, int(up), 0)
            tiles += ((ceil16(c0) + (ceil16(c1) if c1 else 0)) // 16) * ((cout + 15) // 16)
        descs = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev)
        t["keep"].append(descs)              # captured graphs hold raw pointers to earlier tables: never free them
        t.update(dirty=False, descs=descs, n=len(entries), tiles=tiles, entries=tuple(entries))
    if not t["n"]:
        return False
    _call("mnk_conv3x3_pack_multi", t["descs"], _p(t["descs"]), t["n"], t["tiles"])
    for e in t["entries"]:
        w = e.wref()
        if w is not None:
            e.stamp = w._version
    return True


# a split-K convolution whose partials a small-layer BatchNorm will sum: y.data_ptr() -> (ws, splits, phases, bias)
_SPLIT_PENDING = {}
_EVAL_DEFER = [False]                     # conv3x3(eval_bn=True) -> _conv_launch


def _small_sync(rows):
    """several ranks with the peer-to-peer exchange up: a small layer's backward is mnk_bn_small_bwd_sync (one launch)"""
    return mdist.active() and 1 < rows <= _query("mnk_bn_small_rows") and _sync_handle() is not None


def small_bn(rows, training=True, c=0):
    """the one-launch BatchNorm forms of small layers apply: training statistics, few pixel rows (per rank), and either a
    single rank or -- several ranks -- the peer-to-peer exchange up (mnk_bn_small_fwd_sync / _bwd_sync carry it inside the
    launch; a layer of c channels must fit a mailbox row)"""
    if not (training and 1 < rows <= _query("mnk_bn_small_rows")):
        return False
    return not mdist.active() or _sync_handle(c) is not None


def _conv_launch(x0, c0, x1, c1, ups, wp, bias, residual, n, h, w, cout, want_stats=False, up=False):
    """One conv launch.  With want_stats the BatchNorm sums of the output come out of the conv epilogue (finished by a
    tiny second-stage kernel) instead of a separate pass over y; returns (y, sums or None).  up: `wp` holds the
    sub-pixel packs of an up-sampled convolution and (h, w) is the up-sampled size."""
    y = torch.empty(n, h, w, ceil4(cout), dtype=torch.float32, device=x0.device)
    # a small layer whose BatchNorm follows at once (want_stats): that one-launch kernel makes its own statistics and,
    # when this convolution is split along K, also sums the partials -- no separate reduction, no epilogue statistics
    # (even sizes only: a pooling BatchNorm on an odd map -- 96 x 96 frames reach 3 x 3 -- takes the general kernels, and this
    # function does not know whether the BatchNorm that follows pools)
    small = want_stats and residual is None and small_bn(n * h * w, True, cout) and h % 2 == 0 and w % 2 == 0
    # evaluation mode, conv3x3(eval_bn=True): the norm layer that follows sums a split launch's partials itself (one-shot flag)
    evald, _EVAL_DEFER[0] = (_EVAL_DEFER[0] and not want_stats and residual is None and h % 2 == 0 and w % 2 == 0), False
    if _SPLIT_PENDING:
        # the BatchNorm of an earlier deferred split-K convolution never ran: an exception between the two launches (the entry
        # is popped by that BatchNorm; nothing else may run in between).  The partial sums are gone with the scratch buffer;
        # drop the entry -- a retry of the forward pass must not be poisoned by the failed one -- and say so
        import warnings
        warnings.warn("dropping the deferred split-K partial sums of %d convolution(s) whose BatchNorm never ran (an "
                      "exception between a convolution and its normalisation layer?)" % len(_SPLIT_PENDING))
        _SPLIT_PENDING.clear()
    if up:
        assert ups and residual is None
        hl, wl = h // 2, w // 2
        nws = _query("mnk_conv3x3_up_workspace_floats", n, hl, wl, c0, c1, cout)
        ws = SCRATCH.get("ws", nws, x0) if nws else None
        nst = _query("mnk_conv3x3_up_stats_floats", n, hl, wl, c0, c1, cout) if want_stats and not small else 0
        st = torch.empty(nst, dtype=torch.float32, device=x0.device) if nst else None
        defer = 4 if (small or evald) and nws else 0
        _call("mnk_conv3x3_up_fwd", x0, _p(x0), x0.shape[-1], c0, _p(x1), x1.shape[-1] if x1 is not None else 0, c1, defer,
              _p(wp), _p(bias), _p(y), y.shape[-1], n, hl, wl, cout, _p(ws), nws, _p(st))
        if defer:
            _SPLIT_PENDING[y.data_ptr()] = (ws, _query("mnk_conv3x3_up_splits", n, hl, wl, c0, c1, cout), 4, bias)
    else:
        nws = _query("mnk_conv3x3_workspace_floats", n, h, w, c0, c1, cout)
        ws = SCRATCH.get("ws", nws, x0) if nws else None
        nst = _query("mnk_conv3x3_stats_floats", n, h, w, c0, c1, cout) if want_stats and not small else 0
        st = torch.empty(nst, dtype=torch.float32, device=x0.device) if nst else None   # lives until the norm layer reads it
        defer = 4 if (small or evald) and nws else 0
        # flags: bit 0 = nearest x2 up-sampled view, bit 1 = MNK_CONV_CLEAN_PADS -- every act this module produces has zero
        # pad channels (tests/test_modules.py::test_pad_channels_are_written pins that), so the fast 3x3 loader applies
        _call("mnk_conv3x3_fwd", x0, _p(x0), x0.shape[-1], c0, _p(x1), x1.shape[-1] if x1 is not None else 0, c1,
              int(ups) | 2 | defer, _p(wp), _p(bias), _p(residual), residual.shape[-1] if residual is not None else 0, _p(y),
              y.shape[-1], n, h, w, cout, _p(ws), nws, _p(st))
        if defer:
            _SPLIT_PENDING[y.data_ptr()] = (ws, _query("mnk_conv3x3_splits", n, h, w, c0, c1, cout), 1, bias)
    if small:
        return y, y.new_empty(0)
    sums = None
    if want_stats:
        if nst and not mdist.active():
            sums = st                          # per-block partials: finished together with the finalisation (bn_act)
        elif nst:
            sums = torch.empty(2 * cout, dtype=torch.float32, device=x0.device)
            sh = _sync_handle(cout)
            if sh is not None:                 # second stage + exchange over the ranks in one launch
                _call("mnk_bn_stats_finish_sync", x0, sh, _p(st), nst // (2 * y.shape[-1]), y.shape[-1], cout, None, _p(sums),
                      mdist.P2P_TIMEOUT_MS)
                _mark_reduced(sums)
            else:
                _call("mnk_bn_stats_finish", x0, _p(st), nst // (2 * y.shape[-1]), y.shape[-1], cout, _p(sums))
        elif mdist.active():
            sums = channel_sums(y, cout, sync=True)       # split-K layers: statistics by a pass over y
        else:
            sums = y.new_empty(0)              # split-K layers, single process: bn_act makes its own pass over y
    return y, sums


# statistics vectors that already hold the sums over the ranks (see _sync_handle): data pointer -> weak reference to the
# tensor.  The entry dies WITH the tensor (its finaliser removes it), so an unconsumed entry -- an exception mid-forward, a
# norm layer whose .training differs from the block's -- can never mark whatever the allocator puts at that address next
# (ADVICE r4).
_REDUCED = {}


def _mark_reduced(t):
    key = t.data_ptr()
    _REDUCED[key] = weakref.ref(t, lambda _r, key=key: _REDUCED.pop(key, None))


def _take_reduced(t):
    return _REDUCED.pop(t.data_ptr(), None) is not None


def _sync_handle(c=0):
    """The connected peer-to-peer exchange (csrc/p2p.hip) when the BatchNorm statistics of this process are exchanged by it: the
    statistics' second stage then carries the exchange itself (mnk_bn_*_sync: one launch instead of second stage + collective).
    c: the layer's channel count -- 2c sums must fit a mailbox row (mnk_p2p_max_floats: 1056 channels), wider layers take the
    collective path."""
    if not mdist.active():
        return None
    h = mdist.p2p_comm()
    if h is None or 2 * c > mdist._P2P["max"]:
        return None
    import ctypes
    return ctypes.c_void_p(h)


def channel_sums(a, c, sync=False):
    """[sum over pixels of a[..., :c], sum of squares] -> tensor (2c,).  sync: over the ranks, when the peer-to-peer exchange is
    up (the result is then registered in _REDUCED: no all-reduce behind it)."""
    rows = a.numel() // a.shape[-1]
    ld = a.shape[-1]
    nws = _query("mnk_bn_workspace_floats", rows, ld)
    ws = SCRATCH.get("ws", nws, a)
    sums = torch.empty(2 * c, dtype=torch.float32, device=a.device)
    h = _sync_handle(c) if sync else None
    if h is not None:
        _call("mnk_bn_stats_sync", a, h, _p(a), ld, rows, c, None, _p(sums), _p(ws), nws, mdist.P2P_TIMEOUT_MS)
        _mark_reduced(sums)
        return sums
    _call("mnk_bn_stats", a, _p(a), ld, rows, c, _p(sums), _p(ws), nws)
    return sums


# (dy tensor, its column sums) handed from BNActFn.backward to the Conv3x3Fn.backward that receives this very tensor
_DY_SUMS = [None]

# ---- backward statistics of a norm layer from the data-gradient GEMM behind it (round 4) ---------------------------------------
# bn_act() remembers, per OUTPUT tensor z of a training-mode BatchNorm that does not pool, what its backward statistics need
# (y, mean, inv-std, scale, beta, activation).  A convolution that consumes z keeps that record; its data-gradient launch --
# whose output IS dz when z has no other consumer -- then also leaves the column sums of g and g * xhat (mnk_conv3x3_dgrad_bnstats:
# the dz tile is in the GEMM's registers, only y is read), and hands them to the norm layer's backward under the dx tensor it
# returns.  The norm layer's backward looks its incoming gradient up: the very tensor -> one second-stage launch instead of a
# pass over y and dz + second stage; anything else (a second consumer made autograd add two gradients) -> the ordinary pass.
import weakref  # noqa: E402

_BN_OF = {}                               # id(z) -> (weak reference to z, _BnRecord); z = the tensor object callers hold
#                                           (a WeakKeyDictionary would compare tensors with ==)


def _bn_of(t):
    e = _BN_OF.get(id(t)) if t is not None else None
    return e[1] if e is not None and e[0]() is t else None


def _bind_bn(z, rec):
    key = id(z)
    _BN_OF[key] = (weakref.ref(z, lambda _r, key=key: _BN_OF.pop(key, None)), rec)
_LAST_BN = [None]                         # BNActFn.forward -> bn_act(): the record of the launch that just ran
_DZ_STATS = {}                            # dx.data_ptr() -> (dx, partials, rows, y.data_ptr()); dx is held, so the key is unique
DZ_STATS_COUNT = [0, 0]                   # norm-layer backward passes that took the hand-over / that made their own pass (tests)


class _BnRecord:
    __slots__ = ("y", "mean", "invstd", "scale", "beta", "c", "slope")

    def __init__(self, y, mean, invstd, scale, beta, c, slope):
        self.y, self.mean, self.invstd, self.scale, self.beta, self.c, self.slope = y, mean, invstd, scale, beta, c, slope


def clear_dz_stats():
    """drop hand-overs that no norm layer picked up (their gradient was summed with another one first): start of an iteration"""
    _DZ_STATS.clear()
    _REDUCED.clear()


def handover_state():
    """The ONE-SHOT hand-overs between two adjacent calls of this module (producer -> the very next consumer), by name, with what
    is left in them.  Between two forward / backward passes every one of them is empty: anything left over is a producer whose
    consumer never ran (an exception in between, a caller that broke the `conv3x3(..., eval_bn=True)` -> `bn_act` contract) and
    would be picked up by an unrelated later call.  tests/test_step.py and tests/test_inference.py assert {} after whole passes.
    (Not listed: caches and registries keyed by tensor identity + version -- _PACK_*, _BN_EVAL, _BN_OF, _SINKS, _REDUCED,
    _DZ_STATS: entries die with their tensors or at clear_dz_stats().)"""
    slots = {
        "_SPLIT_PENDING (split-K convolution -> the norm layer that sums its partials)": len(_SPLIT_PENDING),
        "_EVAL_DEFER (conv3x3(eval_bn=True) -> _conv_launch)": _EVAL_DEFER[0],
        "_SRC_BN (conv3x3 -> Conv3x3Fn.forward)": _SRC_BN[0],
        "_LAST_BN (BNActFn.forward -> bn_act)": _LAST_BN[0],
        "_DY_SUMS (BNActFn.backward -> Conv3x3Fn.backward)": _DY_SUMS[0],
        "_SKIP_PARAM_GRADS (no_param_grads)": _SKIP_PARAM_GRADS[0],
        "_SKIP_LEAF_INPUT_GRADS (no_leaf_input_grads)": _SKIP_LEAF_INPUT_GRADS[0],
        "FROZEN_CAPTURE (mnk.dropin.EvalRunner._capture)": FROZEN_CAPTURE[0],
    }
    return {k: v for k, v in slots.items() if v}


_SRC_BN = [None]                          # conv3x3() -> Conv3x3Fn.forward: (record of x0's norm layer or None, of x1's)


class Conv3x3Fn(_Fn):
    """nn.Conv3d((1,3,3), padding (0,1,1)) over the channel concatenation [x0 | x1], optionally read through the
    nearest x2 up-sampling (UpBlock3D, modules/util.py:83-85), plus bias and residual add (ResBlock3D :66-67)."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, residual, c0, c1, ups, want_stats, track):
        _check_device(x0)
        cout = weight.shape[0]
        assert weight.shape[1] == c0 + c1 and weight.is_contiguous()
        n, hs, ws_, _ = x0.shape
        h, w = (hs * 2, ws_ * 2) if ups else (hs, ws_)
        up = subpixel(ups) and residual is None
        if track:
            # a backward will follow: the parameter changes every step -> packed per iteration (forward + the
            # data-gradient layouts the backward will need), by repack_registered() or one launch here
            need = [bool(cc and ctx.needs_input_grad[i]) for i, cc in enumerate((c0, c1))]
            e = _pack_entry(weight, cout, c0, c1, need, up)
            wp, ctx.wd = e.wp, list(e.wd)
        else:
            wp = _packed_fwd_weight(weight, cout, c0, c1, up)
        y, sums = _conv_launch(x0, c0, x1, c1, ups, wp, bias, residual, n, h, w, cout, want_stats, up)
        ctx.up = up
        ctx.src_bn, _SRC_BN[0] = (_SRC_BN[0] if track and _SRC_BN[0] is not None else (None, None)), None
        ctx.save_for_backward(x0, x1, weight)
        ctx.meta = (c0, c1, ups, cout, n, h, w, bias is not None, residual is not None)
        if sums is None:
            sums = y.new_empty(0)
        ctx.mark_non_differentiable(sums)
        ctx.set_materialize_grads(False)     # `sums` never has a gradient: no zero tensor is made for it per backward
        return y, sums

    @staticmethod
    def backward(ctx, dy, _dsums):
        return Conv3x3Fn._backward(ctx, dy, None)

    @staticmethod
    def _backward(ctx, dy, dskip):
        """dskip: a second gradient of x0 (Conv3x3SkipFn: x0's other consumer), added in the data gradient's epilogue."""
        x0, x1, weight = ctx.saved_tensors
        c0, c1, ups, cout, n, h, w, has_bias, has_res = ctx.meta
        if dy is None:                       # (materialize_grads is off) nothing flows into y
            return (dskip if ctx.needs_input_grad[0] else None,) + (None,) * 9
        if dskip is not None and not ctx.needs_input_grad[0]:
            dskip = None
        dy_in = dy
        dy = dy.contiguous()
        ld_dy = dy.shape[-1]
        cin = c0 + c1
        grads = [None, None]
        for i, (src, cs, cc) in enumerate(((x0, 0, c0), (x1, c0, c1))):
            if src is None or not ctx.needs_input_grad[i]:
                continue
            wp = ctx.wd[i]                       # packed in the forward (mnk_conv3x3_pack_all)
            # the source is the output of a norm layer (no pooling) and (presumably) has no other consumer: this launch also
            # leaves that layer's backward statistics (see _BN_OF); rows = the geometry both tensors share
            rec = ctx.src_bn[i]
            if rec is not None and (rec.c != cc or rec.y.shape[-1] != ceil4(cc) or small_bn(rec.y.numel() // rec.y.shape[-1], True, cc)
                                    or _small_sync(rec.y.numel() // rec.y.shape[-1])):
                rec = None               # (small layers make their backward statistics inside their own one-launch kernel)
            if ctx.up:
                # data gradient w.r.t. the low-resolution source: one 4x4 / stride 2 convolution over dy (no gradient of
                # the up-sampled view, no 2x2 sum-pool pass)
                dx = torch.empty(n, h // 2, w // 2, ceil4(cc), dtype=torch.float32, device=dy.device)
                nws = _query("mnk_conv3x3_up_dgrad_workspace_floats", n, h // 2, w // 2, cout, cc)
                ws = SCRATCH.get("ws", nws, dy) if nws else None
                nst = _query("mnk_conv3x3_up_dgrad_stats_floats", n, h // 2, w // 2, cout, cc) if (
                    rec is not None and tuple(rec.y.shape) == tuple(dx.shape)) else 0
                if nst:
                    st = torch.empty(nst, dtype=torch.float32, device=dy.device)
                    _call("mnk_conv3x3_up_dgrad_bnstats", dy, _p(dy), ld_dy, cout, _p(wp), _p(dx), dx.shape[-1], n, h // 2,
                          w // 2, cc, _p(ws), nws, _p(st), _p(rec.y), rec.y.shape[-1], _p(rec.mean), _p(rec.invstd),
                          _p(rec.scale), _p(rec.beta), float(rec.slope))
                    _DZ_STATS[dx.data_ptr()] = (dx, st, nst // (2 * dx.shape[-1]), rec.y.data_ptr())
                else:
                    _call("mnk_conv3x3_up_dgrad", dy, _p(dy), ld_dy, cout, _p(wp), _p(dx), dx.shape[-1], n, h // 2, w // 2, cc,
                          _p(ws), nws)
                grads[i] = dx
                continue
            # x0's second gradient rides as the residual operand of the data-gradient GEMM (its epilogue / split reduction)
            res = None
            if i == 0 and dskip is not None and not ups:
                res, dskip = dskip.contiguous(), None
                assert res.shape == (n, h, w, ceil4(cc))
            nst = _query("mnk_conv3x3_stats_floats", n, h, w, cout, 0, cc) if (
                rec is not None and not ups and tuple(rec.y.shape) == (n, h, w, ceil4(cc))) else 0
            if nst:
                dx = torch.empty(n, h, w, ceil4(cc), dtype=torch.float32, device=dy.device)
                nws = _query("mnk_conv3x3_workspace_floats", n, h, w, cout, 0, cc)
                ws = SCRATCH.get("ws", nws, dy) if nws else None
                st = torch.empty(nst, dtype=torch.float32, device=dy.device)
                _call("mnk_conv3x3_dgrad_bnstats", dy, _p(dy), ld_dy, cout, _p(wp), _p(res),
                      res.shape[-1] if res is not None else 0, _p(dx), dx.shape[-1], n, h, w, cc, _p(ws), nws, _p(st), _p(rec.y),
                      rec.y.shape[-1], _p(rec.mean), _p(rec.invstd), _p(rec.scale), _p(rec.beta), float(rec.slope))
                _DZ_STATS[dx.data_ptr()] = (dx, st, nst // (2 * dx.shape[-1]), rec.y.data_ptr())
                grads[i] = dx
                continue
            dx, _ = _conv_launch(dy, cout, None, 0, 0, wp, None, res, n, h, w, cc)
            if ups:
                dxs = torch.empty(n, h // 2, w // 2, ceil4(cc), dtype=torch.float32, device=dy.device)
                _call("mnk_sumpool2x2", dy, _p(dx), dx.shape[-1], _p(dxs), dxs.shape[-1], n, h, w, cc)
                dx = dxs
            grads[i] = dx
        dw = None
        if ctx.needs_input_grad[2]:
            # an optimiser of mnk.optim owns this parameter: the GEMM writes (the partials of) its gradient towards the
            # optimiser's flat buffer and the split reduction waits for the one launch that serves every layer
            owner = sink_owner(weight)
            taken = [False, False]
            if owner is not None:
                for i, (src, cs, cc) in enumerate(((x0, 0, c0), (x1, c0, c1))):
                    if src is not None:
                        taken[i] = owner.reducer.wgrad(weight, src, src.shape[-1], cc, int(ups) | 2, h, w, 3, 3, 1, dy,
                                                       ld_dy, cout, cin, cs, n, h, w)
            rest = [(src, cs, cc) for i, (src, cs, cc) in enumerate(((x0, 0, c0), (x1, c0, c1)))
                    if src is not None and not taken[i]]
            if rest:
                dw = torch.zeros_like(weight) if len(rest) < (1 + (x1 is not None)) else torch.empty_like(weight)
            for src, cs, cc in rest:
                nws = _query("mnk_conv3x3_up_wgrad_workspace_floats" if ups else "mnk_conv3x3_wgrad_workspace_floats", n, h, w,
                             cc, cout)
                ws = SCRATCH.get("ws", nws, dy) if nws else None
                _call("mnk_conv3x3_wgrad", dy, _p(src), src.shape[-1], cc, int(ups) | 2, _p(dy), ld_dy, cout, _p(dw), cin,
                      cs, n, h, w, _p(ws), nws)      # | 2: MNK_CONV_CLEAN_PADS (x and dy are acts of this module)
            if owner is not None and dw is not None:
                owner.add_to_sink(weight, dw)        # a further contribution before the step (slow path)
                dw = None
        db = None
        if has_bias and ctx.needs_input_grad[3]:
            slot, _DY_SUMS[0] = _DY_SUMS[0], None
            if slot is not None and slot[0] is dy_in and slot[1] is None:
                db = None                      # a training-mode BatchNorm follows: the bias gradient is exactly zero
            elif slot is not None and slot[0] is dy_in and slot[1].numel() == cout:
                db = slot[1]                   # column sums of dy came out of the norm layer's backward pass
            else:
                db = channel_sums(dy, cout)[:cout]
        dres = dy if has_res and ctx.needs_input_grad[4] else None
        if dskip is not None:                # forms without a residual operand (up-sampled sources): one add
            grads[0] = grads[0] + dskip if grads[0] is not None else dskip
        return grads[0], grads[1], dw, db, dres, None, None, None, None, None


class Conv3x3SkipFn(_Fn):
    """Conv3x3Fn that also hands x0 through: -> (y, sums, x0).  For a tensor with a second consumer (an hourglass level:
    the next down block's convolution AND the decoder's skip / a warp, util.py:142-152,184-188) the second consumer takes
    the handed-through tensor; this node is then the only consumer of the original and its backward adds the other
    gradient in the epilogue of its data-gradient GEMM (the `residual` operand) -- no accumulation pass by autograd."""

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, residual, c0, c1, ups, want_stats, track):
        y, sums = Conv3x3Fn.forward(ctx, x0, x1, weight, bias, residual, c0, c1, ups, want_stats, track)
        return y, sums, x0

    @staticmethod
    def backward(ctx, dy, _dsums, dskip):
        return Conv3x3Fn._backward(ctx, dy, dskip)


def conv3x3(x0, c0, weight, bias=None, x1=None, c1=0, ups=False, residual=None, want_stats=False, skip=False, eval_bn=False):
    """-> (y, sums): sums = fused BatchNorm statistics [sum, sum of squares] of y when want_stats, else None.
    skip: -> (y, sums, x0 handed through for x0's other consumer), see Conv3x3SkipFn.
    eval_bn: the caller vouches that y's ONLY consumer is the evaluation-mode norm layer it calls next (bn_act): under no_grad a
    split-K launch then leaves its partials to that layer (mnk_bn_eval_split_fwd: reduction + affine + ReLU + pool in one launch)
    and y is never written."""
    track = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (x0, x1, weight, bias, residual))
    _EVAL_DEFER[0] = bool(eval_bn and not want_stats and residual is None and not torch.is_grad_enabled()
                          and knobs.form("EVAL_SPLIT_FUSED"))
    if track:       # the norm layers whose outputs the sources are (see _BN_OF): picked up by Conv3x3Fn.forward
        _SRC_BN[0] = (_bn_of(x0), _bn_of(x1))
    try:
        if skip and track and x0.requires_grad and knobs.form("SKIP_GRAD_FUSED"):
            y, sums, through = Conv3x3SkipFn.apply(x0, x1, weight, bias, residual, c0, c1, bool(ups), bool(want_stats), track)
            return y, (sums if want_stats else None), through
        y, sums = Conv3x3Fn.apply(x0, x1, weight, bias, residual, c0, c1, bool(ups), bool(want_stats), track)
    finally:
        # one-shot hand-overs of THIS call: an exception in front of their consumer must not leave them for an unrelated launch
        # (a data-gradient launch of a later backward also goes through _conv_launch)
        _EVAL_DEFER[0] = False
        _SRC_BN[0] = None
    return (y, (sums if want_stats else None), x0) if skip else (y, (sums if want_stats else None))


# ----------------------------------------------------------------------------------------------------------------
# BatchNorm (+ReLU, +2x2 average pool)
# ----------------------------------------------------------------------------------------------------------------
# Evaluation-mode coefficients (mean, 1/sqrt(var + eps), gamma / sqrt(var + eps)) of a norm layer, kept between calls: the
# reference's evaluation loops run the networks frame by frame with constant weights, and the coefficient launch was 40 of
# the 135 launches per frame (tools/frame_loop_probe.py).  Valid while nothing wrote the layer's tensors: tensor versions
# (load_state_dict, in-place ops), storage addresses, the optimiser-step / replay epoch (_PACK_EPOCH: fused optimisers and
# captured iterations do not bump versions) and the training-forward epoch (kernels update running statistics through raw
# pointers) are all part of the key; never used under stream capture (a replay must read the live buffers).
_BN_EVAL = {}
_BN_EVAL_EPOCH = [0]


def _bn_eval_coeffs(y, gamma, running_mean, running_var, eps, c):
    capturing = y.is_cuda and torch.cuda.is_current_stream_capturing() and not FROZEN_CAPTURE[0]
    if not capturing:
        key = (0 if gamma is None else gamma.data_ptr(), 0 if gamma is None else gamma._version, running_mean.data_ptr(),
               running_mean._version, running_var.data_ptr(), running_var._version, float(eps), c, _stream(y),
               _PACK_EPOCH[0], _BN_EVAL_EPOCH[0])
        hit = _BN_EVAL.get(id(running_mean))
        if hit is not None and hit[0] == key and hit[2]() is running_mean:
            return hit[1]
    mean = torch.empty(c, dtype=torch.float32, device=y.device)
    invstd, scale = torch.empty_like(mean), torch.empty_like(mean)
    _call("mnk_bn_eval_coeffs", y, _p(gamma), _p(running_mean), _p(running_var), float(eps), c, _p(mean), _p(invstd), _p(scale))
    if not capturing:
        if id(running_mean) not in _BN_EVAL:
            weakref.finalize(running_mean, _BN_EVAL.pop, id(running_mean), None)
        _BN_EVAL[id(running_mean)] = (key, (mean, invstd, scale), weakref.ref(running_mean))
    return mean, invstd, scale


class BNActFn(_Fn):
    """SynchronizedBatchNorm3d (sync_batchnorm/batchnorm.py:48-78) fused with the ReLU and the AvgPool3d((1,2,2))
    that follow it in DownBlock3D / UpBlock3D / SameBlock3D / ResBlock3D (modules/util.py).  In training mode under
    torch.distributed the sufficient statistics are all-reduced over the ranks (SyncBN over RCCL)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, pre_sums, c, training, relu, pool, momentum, eps):
        _check_device(y)
        n, h, w, ld = y.shape
        rows = n * h * w
        dev = y.device
        if training:
            _BN_EVAL_EPOCH[0] += 1          # kernels are about to write running statistics through raw pointers
        if training:
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            invstd = torch.empty_like(mean)
            scale = torch.empty_like(mean)
        count = float(rows)
        pending = _SPLIT_PENDING.pop(y.data_ptr(), None)
        ctx.small = False
        if small_bn(rows, training, c) and ld == ceil4(c) and h % (2 if pool else 1) == 0 and w % (2 if pool else 1) == 0:
            # the whole layer in one launch (csrc/batchnorm.hip: bn_small_fwd_kernel; several ranks: bn_small_fwd_sync_kernel,
            # the exchange of the sums inside the launch, statistics over the rows of ALL ranks)
            ho, wo = (h // 2, w // 2) if pool else (h, w)
            z = torch.empty(n, ho, wo, ceil4(c), dtype=torch.float32, device=dev)
            ws_, splits, phases, bias = pending if pending is not None else (None, 0, 1, None)
            if mdist.active():
                count *= mdist.world_size()
                _call("mnk_bn_small_fwd_sync", y, _sync_handle(c), _p(ws_), splits, ld, phases, _p(bias), _p(y), ld, n, h, w, c,
                      _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(momentum), float(eps), _p(mean), _p(invstd),
                      _p(scale), _p(z), z.shape[-1], int(relu), int(pool), mdist.P2P_TIMEOUT_MS)
            else:
                _call("mnk_bn_small_fwd", y, _p(ws_), splits, ld, phases, _p(bias), _p(y), ld, n, h, w, c, _p(gamma), _p(beta),
                      _p(running_mean), _p(running_var), float(momentum), float(eps), _p(mean), _p(invstd), _p(scale), _p(z),
                      z.shape[-1], int(relu), int(pool))
            ctx.save_for_backward(y, mean, invstd, scale, beta)
            ctx.meta = (c, training, relu, pool, count)
            ctx.small = True
            return z
        if pending is not None and not training and ld == ceil4(c) and h % 2 == 0 and w % 2 == 0:
            # evaluation mode behind a split-K convolution (conv3x3(eval_bn=True), no_grad): reduction + affine + ReLU + pool in
            # one launch; y was never written and nothing is saved for a backward pass
            mean, invstd, scale = _bn_eval_coeffs(y, gamma, running_mean, running_var, eps, c)
            ws_, splits, phases, bias = pending
            ho, wo = (h // 2, w // 2) if pool else (h, w)
            z = torch.empty(n, ho, wo, ceil4(c), dtype=torch.float32, device=dev)
            _call("mnk_bn_eval_split_fwd", y, _p(ws_), splits, ld, phases, _p(bias), _p(mean), _p(scale), _p(beta), _p(z),
                  z.shape[-1], n, h, w, c, int(relu), int(pool))
            return z
        if pending is not None:
            raise RuntimeError("split-K partials were deferred to a BatchNorm that does not take the small-layer path")
        if training:
            if rows * mdist.world_size() <= 1:
                raise ValueError("BatchNorm needs more than one value per channel in training mode "
                                 "(sync_batchnorm/batchnorm.py:116)")
            if mdist.active():
                sums = pre_sums if pre_sums is not None and pre_sums.numel() == 2 * c else channel_sums(y, c, sync=True)
                if _take_reduced(sums):                  # the statistics' second stage carried the exchange
                    pass
                else:
                    sums = mdist.all_reduce_sum(sums) if sums is pre_sums else mdist.all_reduce_sum_(sums)
                count *= mdist.world_size()
                # finalisation inside the apply pass (mnk_bn_act_fwd_sums): one launch less per norm layer on the SyncBN path
                ho, wo = (h // 2, w // 2) if pool else (h, w)
                z = torch.empty(n, ho, wo, ceil4(c), dtype=torch.float32, device=dev)
                _call("mnk_bn_act_fwd_sums", y, _p(y), ld, _p(sums), count, _p(gamma), _p(beta), _p(running_mean),
                      _p(running_var), float(momentum), float(eps), 1, _p(mean), _p(invstd), _p(scale), _p(z), z.shape[-1], 0, n, h,
                      w, c, int(relu), int(pool))
                ctx.save_for_backward(y, mean, invstd, scale, beta)
                ctx.meta = (c, training, relu, pool, count)
                if not pool and ld == z.shape[-1] and knobs.form("DGRAD_BN_STATS"):
                    _LAST_BN[0] = _BnRecord(y, mean, invstd, scale, beta, c, 0.0 if relu else -1.0)
                return z
            else:
                # one launch for second stage + finalisation; partials from the conv epilogue when it produced them
                part = pre_sums if pre_sums is not None and pre_sums.numel() > 2 * c else None
                if pre_sums is not None and part is None and pre_sums.numel() == 2 * c:
                    _call("mnk_bn_finalize", y, _p(pre_sums), count, _p(gamma), _p(running_mean), _p(running_var),
                          float(momentum), float(eps), c, 1, _p(mean), _p(invstd), _p(scale))
                else:
                    nws = 0 if part is not None else _query("mnk_bn_workspace_floats", rows, ld)
                    ws = SCRATCH.get("ws", nws, y) if nws else None
                    _call("mnk_bn_stats_finalize", y, _p(y), ld, rows, c, _p(part),
                          part.numel() // (2 * ld) if part is not None else 0, count, _p(gamma), _p(running_mean),
                          _p(running_var), float(momentum), float(eps), 1, None, _p(mean), _p(invstd), _p(scale),
                          _p(ws), nws)
        else:
            mean, invstd, scale = _bn_eval_coeffs(y, gamma, running_mean, running_var, eps, c)
        ho, wo = (h // 2, w // 2) if pool else (h, w)
        z = torch.empty(n, ho, wo, ceil4(c), dtype=torch.float32, device=dev)     # the kernel zeroes the pad channels
        _call("mnk_bn_act_fwd", y, _p(y), ld, _p(mean), _p(scale), _p(beta), _p(z), z.shape[-1], 0, n, h, w, c, int(relu),
              int(pool))
        ctx.save_for_backward(y, mean, invstd, scale, beta)
        ctx.meta = (c, training, relu, pool, count)
        if training and not pool and ld == z.shape[-1] and knobs.form("DGRAD_BN_STATS"):
            _LAST_BN[0] = _BnRecord(y, mean, invstd, scale, beta, c, 0.0 if relu else -1.0)
        return z

    @staticmethod
    def backward(ctx, dz):
        return BNActFn._backward(ctx, dz, None)

    @staticmethod
    def _backward(ctx, dz, dskip):
        """dskip: a second gradient of y (BNActSkipFn: the skip path of a residual block), added inside the dy pass."""
        y, mean, invstd, scale, beta = ctx.saved_tensors
        c, training, relu, pool, count = ctx.meta
        dz = dz.contiguous()
        n, h, w, ld = y.shape
        rows = n * h * w
        if dskip is not None:
            dskip = dskip.contiguous()
            assert dskip.shape == y.shape, "the skip gradient of a residual block has the shape of the block's input"
        if ctx.small and not mdist.active():
            sums = torch.empty(2 * c, dtype=torch.float32, device=y.device)
            dy = torch.empty(n, h, w, ld, dtype=torch.float32, device=y.device)
            _call("mnk_bn_small_bwd", y, _p(y), ld, _p(dz), dz.shape[-1], _p(mean), _p(invstd), _p(scale), _p(beta), count, n, h,
                  w, c, int(relu), int(pool), _p(sums), _p(dy), ld)
            if dskip is not None:
                dy = dy + dskip                # few-pixel layers: the one-launch kernel has no addend operand
                _DY_SUMS[0] = None             # ... and the sum's column sums are not zero: the convolution in front makes them
            else:
                _DY_SUMS[0] = (dy, None)
            return dy, sums[c:], sums[:c], None, None, None, None, None, None, None, None, None
        sync = _sync_handle(c) if training else None      # several ranks: the second stage carries the exchange of the sums
        if (sync is not None and _small_sync(rows) and ld == ceil4(c) and 2 * c <= mdist._P2P["max"]
                and (not pool or (h % 2 == 0 and w % 2 == 0))):
            # a small layer on one rank of several: statistics, their exchange and the apply pass in ONE launch, as the
            # single-process path does it (mnk_bn_small_bwd) -- the general path spends three to four launches here
            _DZ_STATS.pop(dz.data_ptr(), None)
            local = torch.empty(2 * c, dtype=torch.float32, device=y.device)
            dy = torch.empty(n, h, w, ld, dtype=torch.float32, device=y.device)
            _call("mnk_bn_small_bwd_sync", y, sync, _p(y), ld, _p(dz), dz.shape[-1], _p(mean), _p(invstd), _p(scale), _p(beta),
                  count, n, h, w, c, int(relu), int(pool), _p(local), _p(dy), ld, mdist.P2P_TIMEOUT_MS)
            if dskip is not None:
                dy = dy + dskip
                _DY_SUMS[0] = None
            else:
                _DY_SUMS[0] = (dy, None)
            return dy, local[c:], local[:c], None, None, None, None, None, None, None, None, None
        nws = _query("mnk_bn_workspace_floats", rows, ceil4(c))
        ws = SCRATCH.get("ws", nws, y)
        sums = torch.empty(2 * c, dtype=torch.float32, device=y.device)
        pre = _DZ_STATS.pop(dz.data_ptr(), None)
        local = torch.empty(2 * c, dtype=torch.float32, device=y.device) if sync is not None else sums
        if pre is not None and pre[0].shape == dz.shape and pre[3] == y.data_ptr() and not pool:
            # this gradient is the data-gradient GEMM's own output: its epilogue left the statistics' first stage
            if sync is not None:
                _call("mnk_bn_stats_finish_sync", y, sync, _p(pre[1]), pre[2], ld, c, _p(local), _p(sums), mdist.P2P_TIMEOUT_MS)
            else:
                _call("mnk_bn_stats_finish", y, _p(pre[1]), pre[2], ld, c, _p(sums))
            DZ_STATS_COUNT[0] += 1
        else:
            DZ_STATS_COUNT[1] += 1
            if sync is not None:
                _call("mnk_bn_act_bwd_stats_sync", y, sync, _p(y), ld, _p(dz), dz.shape[-1], 0, _p(mean), _p(invstd), _p(scale),
                      _p(beta), n, h, w, c, int(relu), int(pool), _p(local), _p(sums), _p(ws), nws, mdist.P2P_TIMEOUT_MS)
            else:
                _call("mnk_bn_act_bwd_stats", y, _p(y), ld, _p(dz), dz.shape[-1], 0, _p(mean), _p(invstd), _p(scale), _p(beta), n,
                      h, w, c, int(relu), int(pool), _p(sums), _p(ws), nws)
        dbeta, dgamma = local[:c], local[c:]     # local contributions (averaged later together with all gradients)
        if training and mdist.active() and sync is None:
            sums = mdist.all_reduce_sum(sums)
        dy = torch.empty(n, h, w, ld, dtype=torch.float32, device=y.device)
        if dskip is not None:
            # dy = BatchNorm backward + skip gradient in one pass, with the column sums of the SUM: the bias gradient of the
            # convolution that produced y (the previous residual block's second convolution) -- not zero here
            dy_sums = torch.empty(c, dtype=torch.float32, device=y.device)
            _call("mnk_bn_act_bwd_apply_add_colsum", y, _p(y), ld, _p(dz), dz.shape[-1], 0, _p(mean), _p(invstd), _p(scale),
                  _p(beta), _p(sums), count, int(training), _p(dskip), dskip.shape[-1], _p(dy), ld, n, h, w, c, int(relu),
                  int(pool), _p(dy_sums), _p(ws), nws)
            _DY_SUMS[0] = (dy, dy_sums)
        elif training:
            # training-mode statistics: sum over pixels of dy = scale * (sum g - N * mean(g) - k2 * sum xhat) == 0 exactly, so
            # the bias gradient of the convolution in front (util.py:54-56,80-81,99-100) is analytically zero; the reference
            # computes its rounding noise.  No reduction is spent on it: that bias simply receives no gradient (an optimiser
            # then leaves it alone -- BatchNorm cancels any bias anyway).  MNK_BN_ZERO_BIAS_GRAD=0 computes the noise.
            _call("mnk_bn_act_bwd_apply", y, _p(y), ld, _p(dz), dz.shape[-1], 0, _p(mean), _p(invstd), _p(scale), _p(beta),
                  _p(sums), count, 1, _p(dy), ld, n, h, w, c, int(relu), int(pool))
            _DY_SUMS[0] = (dy, None)
        else:
            dy_sums = torch.empty(c, dtype=torch.float32, device=y.device)
            _call("mnk_bn_act_bwd_apply_colsum", y, _p(y), ld, _p(dz), dz.shape[-1], 0, _p(mean), _p(invstd), _p(scale),
                  _p(beta), _p(sums), count, int(training), _p(dy), ld, n, h, w, c, int(relu), int(pool), _p(dy_sums), _p(ws),
                  nws)
            _DY_SUMS[0] = (dy, dy_sums)      # picked up by Conv3x3Fn.backward when it receives this very tensor
        return dy, dgamma, dbeta, None, None, None, None, None, None, None, None, None


class BNActSkipFn(_Fn):
    """BNActFn for the first norm layer of a residual block (util.py:58-67): returns (z, y) -- y handed through for the
    block's `out += x`.  Both consumers of the block's input are then this one node, and its backward adds the gradient of
    the skip path inside the BatchNorm dy pass (mnk_bn_act_bwd_apply_add_colsum) instead of leaving a separate accumulation
    pass over both gradients to autograd."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, pre_sums, c, training, relu, pool, momentum, eps):
        z = BNActFn.forward(ctx, y, gamma, beta, running_mean, running_var, pre_sums, c, training, relu, pool, momentum, eps)
        return z, y

    @staticmethod
    def backward(ctx, dz, dskip):
        if dz is None:          # only the skip path was used
            return (dskip,) + (None,) * 11
        return BNActFn._backward(ctx, dz, dskip)


def bn_act(y, c, norm, relu=True, pool=False, sums=None, skip=False):
    """`norm` is a sync_batchnorm.SynchronizedBatchNorm3d parameter holder; `sums` = statistics of y that a conv
    epilogue already produced (training mode).  skip: -> (z, y handed through for a residual add), see BNActSkipFn."""
    fn = BNActSkipFn if skip and knobs.form("RES_SKIP_FUSED") else BNActFn
    _LAST_BN[0] = None
    out = fn.apply(y, norm.weight, norm.bias, norm.running_mean, norm.running_var,
                   sums if norm.training else None, c, norm.training, relu, pool, norm.momentum, norm.eps)
    rec, _LAST_BN[0] = _LAST_BN[0], None
    if rec is not None and torch.is_grad_enabled():
        z = out[0] if isinstance(out, tuple) else out
        if z.requires_grad:
            _bind_bn(z, rec)
    return (out, y) if skip and fn is BNActFn else out


# ----------------------------------------------------------------------------------------------------------------
# grouped 1x1 conv, 1x1 conv + sigmoid
# ----------------------------------------------------------------------------------------------------------------
_SKIP_PARAM_GRADS = [False]


class no_param_grads:
    """Context: backward passes of the discriminator's convolutions skip their weight / bias gradients (the caller
    asked torch.autograd.grad for input gradients only -- a custom Function cannot see that restriction)."""

    def __enter__(self):
        self.prev, _SKIP_PARAM_GRADS[0] = _SKIP_PARAM_GRADS[0], True

    def __exit__(self, *exc):
        _SKIP_PARAM_GRADS[0] = self.prev


_SKIP_LEAF_INPUT_GRADS = [False]


class no_leaf_input_grads:
    """Context: a ConvKxKFn marked `leaf_input` (the discriminator's first convolution) skips its data gradient -- the
    caller's backward only asks for parameter gradients and nothing below that convolution will run (the custom
    Function cannot see torch.autograd.backward(inputs=...))."""

    def __enter__(self):
        self.prev, _SKIP_LEAF_INPUT_GRADS[0] = _SKIP_LEAF_INPUT_GRADS[0], True

    def __exit__(self, *exc):
        _SKIP_LEAF_INPUT_GRADS[0] = self.prev


class ConvKxKFn(_Fn):
    """nn.Conv3d((1,k,k)), stride 1, zero padding `pad`, single source -- the discriminator's 4x4 convolutions without
    padding (modules/discriminator.py:17-18,28) on the same implicit-GEMM kernels; the data gradient is the same
    kernel on dy with pad k-1-pad and the flipped / transposed pack."""

    @staticmethod
    def forward(ctx, x, weight, bias, cin, kh, kw, pad, leaf_input=False):
        _check_device(x)
        ctx.leaf_input = bool(leaf_input)
        cout = weight.shape[0]
        n, hi, wi, ld = x.shape
        ho, wo = hi + 2 * pad - kh + 1, wi + 2 * pad - kw + 1
        nt = kh * kw
        wp = SCRATCH.get("pack", _query("mnk_conv2d_packed_floats", cout, cin, 0, nt), x)
        ctx.wd = None
        if ctx.needs_input_grad[0] and nt <= 16:
            # forward and data-gradient layouts in ONE launch; the latter serves every backward pass through this node (the
            # generator pass and the discriminator pass of one iteration read the same, not yet updated, weights)
            ctx.wd = torch.empty(_query("mnk_conv2d_packed_floats", cin, cout, 0, nt), dtype=torch.float32, device=x.device)
            _call("mnk_conv2d_pack_all", x, _p(weight), _p(wp), _p(ctx.wd), None, cout, cin, 0, nt)
        else:
            _call("mnk_conv2d_pack_fwd", x, _p(weight), _p(wp), cout, cin, 0, nt)
        y = torch.empty(n, ho, wo, ceil4(cout), dtype=torch.float32, device=x.device)
        nws = _query("mnk_conv2d_workspace_floats", n, ho, wo, cin, 0, cout, nt)
        ws = SCRATCH.get("ws", nws, x) if nws else None
        # flags = MNK_CONV_CLEAN_PADS: x is an act of this module (zero pad channels) -> the K x K buffer-load loader
        _call("mnk_conv2d_fwd", x, _p(x), ld, cin, None, 0, 0, 2, hi, wi, kh, kw, pad, _p(wp), _p(bias), None, 0, _p(y),
              y.shape[-1], n, ho, wo, cout, _p(ws), nws, None)
        ctx.save_for_backward(x, weight)
        ctx.meta = (cin, cout, kh, kw, pad, n, hi, wi, ho, wo, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        cin, cout, kh, kw, pad, n, hi, wi, ho, wo, has_bias = ctx.meta
        dy = dy.contiguous()
        nt = kh * kw
        dx = dw = db = None
        if ctx.needs_input_grad[0] and not (ctx.leaf_input and _SKIP_LEAF_INPUT_GRADS[0]):
            wp = ctx.wd
            if wp is None:
                wp = SCRATCH.get("pack", _query("mnk_conv2d_packed_floats", cin, cout, 0, nt), dy)
                _call("mnk_conv2d_pack_dgrad", dy, _p(weight), _p(wp), cout, cin, 0, cin, nt)
            dx = torch.empty(n, hi, wi, ceil4(cin), dtype=torch.float32, device=dy.device)
            nws = _query("mnk_conv2d_workspace_floats", n, hi, wi, cout, 0, cin, nt)
            ws = SCRATCH.get("ws", nws, dy) if nws else None
            _call("mnk_conv2d_fwd", dy, _p(dy), dy.shape[-1], cout, None, 0, 0, 2, ho, wo, kh, kw, kh - 1 - pad, _p(wp), None,
                  None, 0, _p(dx), dx.shape[-1], n, hi, wi, cin, _p(ws), nws, None)
        if ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS[0]:
            owner = sink_owner(weight)
            if owner is None or not owner.reducer.wgrad(weight, x, x.shape[-1], cin, 0, hi, wi, kh, kw, pad, dy,
                                                        dy.shape[-1], cout, cin, 0, n, ho, wo):
                dw = torch.empty_like(weight)
                nws = _query("mnk_conv2d_wgrad_workspace_floats", n, ho, wo, cin, cout, kh, kw, pad)
                ws = SCRATCH.get("ws", nws, dy) if nws else None
                _call("mnk_conv2d_wgrad", dy, _p(x), x.shape[-1], cin, 0, hi, wi, kh, kw, pad, _p(dy), dy.shape[-1], cout,
                      _p(dw), cin, 0, n, ho, wo, _p(ws), nws)
                if owner is not None:
                    owner.add_to_sink(weight, dw)
                    dw = None
        if has_bias and ctx.needs_input_grad[2] and not _SKIP_PARAM_GRADS[0]:
            db = channel_sums(dy, cout)[:cout]
        return dx, dw, db, None, None, None, None, None


class InstNormActFn(_Fn):
    """[InstanceNorm3d (affine)] -> LeakyReLU(slope) -> avg_pool (1,2,2) of the discriminator's DownBlock3D
    (modules/discriminator.py:26-33): per-frame statistics (the time axis is 1 on every call path), one fused pass."""

    @staticmethod
    def forward(ctx, y, gamma, beta, c, slope, pool, eps):
        _check_device(y)
        n, h, w, ld = y.shape
        dev = y.device
        has_norm = gamma is not None
        if has_norm:
            nws = _query("mnk_norm_workspace_floats", h * w, n, ld)
            ws = SCRATCH.get("ws", nws, y)
            sums = torch.empty(2 * n * c, dtype=torch.float32, device=dev)
            _call("mnk_norm_stats", y, _p(y), ld, h * w, n, c, _p(sums), _p(ws), nws)
            mean = torch.empty(n * c, dtype=torch.float32, device=dev)
            invstd, scale = torch.empty_like(mean), torch.empty_like(mean)
            _call("mnk_norm_finalize", y, _p(sums), float(h * w), _p(gamma), None, None, 0.0, float(eps), c, n, 0, _p(mean),
                  _p(invstd), _p(scale))
            bt = beta
        else:   # first block: no normalisation -> identity affine
            mean, invstd = _identity_affine(c, dev)          # constants, made once per (channels, device)
            scale, bt = invstd, mean
        ho, wo = (h // 2, w // 2) if pool else (h, w)
        z = torch.empty(n, ho, wo, ceil4(c), dtype=torch.float32, device=dev)
        _call("mnk_norm_act_fwd", y, _p(y), ld, _p(mean), _p(scale), _p(bt), int(has_norm), _p(z), z.shape[-1], 0, n, h, w, c,
              float(slope), int(pool))
        ctx.save_for_backward(y, mean, invstd, scale, bt)
        ctx.meta = (c, float(slope), pool, has_norm)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, mean, invstd, scale, bt = ctx.saved_tensors
        c, slope, pool, has_norm = ctx.meta
        dz = dz.contiguous()
        n, h, w, ld = y.shape
        dy = torch.empty_like(y)
        dgamma = dbeta = None
        sums = None
        if has_norm:
            nws = _query("mnk_norm_workspace_floats", h * w, n, ceil4(c))
            ws = SCRATCH.get("ws", nws, y)
            sums = torch.empty(2 * n * c, dtype=torch.float32, device=y.device)
            _call("mnk_norm_act_bwd_stats", y, _p(y), ld, _p(dz), dz.shape[-1], 0, _p(mean), _p(invstd), _p(scale), _p(bt), 1,
                  n, h, w, c, slope, int(pool), _p(sums), _p(ws), nws)
            if not _SKIP_PARAM_GRADS[0]:
                per = sums.view(2, n, c).sum(dim=1)
                dbeta, dgamma = per[0], per[1]
        _call("mnk_norm_act_bwd_apply", y, _p(y), ld, _p(dz), dz.shape[-1], 0, _p(mean), _p(invstd), _p(scale), _p(bt),
              int(has_norm), _p(sums), float(h * w), int(has_norm), _p(dy), ld, n, h, w, c, slope, int(pool))
        return dy, dgamma, dbeta, None, None, None, None


_IDENTITY_AFFINE = {}


def _identity_affine(c, dev):
    t = _IDENTITY_AFFINE.get((c, dev))
    if t is None:
        t = _IDENTITY_AFFINE[(c, dev)] = (torch.zeros(c, dtype=torch.float32, device=dev),
                                          torch.ones(c, dtype=torch.float32, device=dev))
    return t


class PairL1Fn(_Fn):
    """weight * mean_batch(|generated - real|) (modules/losses.py:8-12) of one discriminator feature map, read from the
    act of the batched pass [generated | real] (2B frames) -> tensor (B,).  One launch forward, one backward; the
    feature maps never take their NCDHW form."""

    @staticmethod
    def forward(ctx, act, c, b, weight):
        _check_device(act)
        n, h, w, ld = act.shape
        assert n == 2 * b and act.is_contiguous()
        out = torch.empty(b, dtype=torch.float32, device=act.device)
        _call("mnk_pair_l1_fwd", act, _p(act), ld, h * w, c, b, float(weight), _p(out))
        ctx.save_for_backward(act)
        ctx.meta = (c, b, float(weight))
        return out

    @staticmethod
    def backward(ctx, g):
        act, = ctx.saved_tensors
        c, b, weight = ctx.meta
        n, h, w, ld = act.shape
        g = g.contiguous()
        da = torch.empty_like(act)
        _call("mnk_pair_l1_bwd", act, _p(act), ld, h * w, c, b, weight, _p(g), _p(da))
        return da, None, None, None


class PairL1TapFn(_Fn):
    """PairL1Fn as a tap on the discriminator's forward pass: (the map itself for the next block, the loss vector).  The map
    then has ONE consumer in the autograd graph, and the gradient of the next block is added inside the L1 backward kernel
    (autograd's own accumulation: one more pass over every feature map of the batched [generated | real] pass)."""

    @staticmethod
    def forward(ctx, act, c, b, weight):
        _check_device(act)
        n, h, w, ld = act.shape
        assert n == 2 * b and act.is_contiguous()
        out = torch.empty(b, dtype=torch.float32, device=act.device)
        _call("mnk_pair_l1_fwd", act, _p(act), ld, h * w, c, b, float(weight), _p(out))
        ctx.save_for_backward(act)
        ctx.meta = (c, b, float(weight))
        ctx.set_materialize_grads(False)
        return act.view_as(act), out

    @staticmethod
    def backward(ctx, g_act, g):
        if g is None:
            return g_act, None, None, None
        act, = ctx.saved_tensors
        c, b, weight = ctx.meta
        n, h, w, ld = act.shape
        if g_act is not None:
            g_act = g_act.contiguous()
        da = torch.empty_like(act)
        _call("mnk_pair_l1_bwd_add", act, _p(act), ld, h * w, c, b, weight, _p(g.contiguous()), _p(g_act), _p(da))
        return da, None, None, None


class L1MeanFn(_Fn):
    """weight * mean_batch(|prediction - target|) (modules/losses.py:8-12) of two equally shaped contiguous tensors
    (the frames themselves: reconstruction_deformed, map 0 of 'reconstruction') -> tensor (B,).  One launch each way
    instead of sub / abs / mean / mul and their four backward nodes."""

    @staticmethod
    def forward(ctx, prediction, target, weight):
        _check_device(prediction)
        assert prediction.shape == target.shape
        prediction, target = prediction.contiguous(), target.contiguous()
        b = prediction.shape[0]
        out = torch.empty(b, dtype=torch.float32, device=prediction.device)
        _call("mnk_l1_mean_fwd", prediction, _p(prediction), _p(target), prediction.numel() // b, b, float(weight), _p(out))
        ctx.save_for_backward(prediction, target)
        ctx.weight = float(weight)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if da is None and db is None:
            return None, None, None
        n = a.shape[0]
        _call("mnk_l1_mean_bwd", a, _p(a), _p(b), a.numel() // n, n, ctx.weight, _p(g.contiguous()), _p(da), _p(db))
        return da, db, None


class GanTermsFn(_Fn):
    """(generator_gan_loss, discriminator_gan_loss) of modules/losses.py:15-21 from the score maps of the batched
    discriminator pass [generated | real] (2B samples) -> two tensors (B,).  One launch each way; the score tensor is
    not sliced (a slice's backward is a zero fill and a copy)."""

    @staticmethod
    def forward(ctx, score, b, w_gen, w_disc):
        _check_device(score)
        score = score.contiguous()
        assert score.shape[0] == 2 * b
        gen = torch.empty(b, dtype=torch.float32, device=score.device)
        disc = torch.empty(b, dtype=torch.float32, device=score.device)
        _call("mnk_gan_terms_fwd", score, _p(score), score.numel() // (2 * b), b, float(w_gen), float(w_disc), _p(gen), _p(disc))
        ctx.save_for_backward(score)
        ctx.meta = (b, float(w_gen), float(w_disc))
        ctx.set_materialize_grads(False)
        return gen, disc

    @staticmethod
    def backward(ctx, ggen, gdisc):
        score, = ctx.saved_tensors
        b, w_gen, w_disc = ctx.meta
        if ggen is None and gdisc is None:
            return None, None, None, None
        ds = torch.empty_like(score)
        _call("mnk_gan_terms_bwd", score, _p(score), score.numel() // (2 * b), b, w_gen, w_disc,
              _p(ggen.contiguous()) if ggen is not None else None, _p(gdisc.contiguous()) if gdisc is not None else None, _p(ds))
        return ds, None, None, None


class LossMeansFn(_Fn):
    """[v.mean() for v in vecs] and the sum of these means (train.py:114,116) for per-sample loss vectors of one length:
    (means (n,), total ()).  One launch each way instead of stack / mean / sum and their expand / div backward passes."""

    @staticmethod
    def forward(ctx, *vecs):
        import numpy as np
        ref = vecs[0]
        _check_device(ref)
        vecs = [v.contiguous().float() for v in vecs]
        n, ln = len(vecs), vecs[0].numel()
        assert all(v.numel() == ln for v in vecs)
        out = torch.empty(n + 1, dtype=torch.float32, device=ref.device)
        ptrs = np.array([v.data_ptr() for v in vecs], dtype=np.uint64)
        _call("mnk_vec_means_fwd", ref, ptrs.ctypes.data, n, ln, _p(out))
        ctx.meta = (n, ln, [v.shape for v in vecs])
        ctx.set_materialize_grads(False)
        return out[:n], out[n]

    @staticmethod
    def backward(ctx, gmeans, gtotal):
        n, ln, shapes = ctx.meta
        if gmeans is None and gtotal is None:
            return (None,) * n
        ref = gmeans if gmeans is not None else gtotal
        gv = torch.empty(n, ln, dtype=torch.float32, device=ref.device)
        _call("mnk_vec_means_bwd", ref, _p(gmeans.contiguous()) if gmeans is not None else None,
              _p(gtotal.contiguous()) if gtotal is not None else None, n, ln, _p(gv))
        return tuple(g.view(shp) for g, shp in zip(gv.unbind(0), shapes))


class GConv1x1Fn(_Fn):
    """nn.Conv3d(kernel (1,1,1), groups=num_kp+1) of SameBlock3D (dense_motion_module.py:24-28)."""

    @staticmethod
    def forward(ctx, x, weight, bias, groups):
        _check_device(x)
        c = weight.shape[0]
        s = weight.shape[1]
        assert c == groups * s
        n, h, w, ld = x.shape
        y = torch.empty(n, h, w, ceil4(c), dtype=torch.float32, device=x.device)
        _call("mnk_gconv1x1_fwd", x, _p(x), ld, _p(weight), _p(bias), _p(y), y.shape[-1], n * h * w, groups, s)
        ctx.save_for_backward(x, weight)
        ctx.meta = (groups, s, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        groups, s, has_bias = ctx.meta
        dy = dy.contiguous()
        n, h, w, ld = x.shape
        rows = n * h * w
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _call("mnk_gconv1x1_bwd_data", dy, _p(dy), dy.shape[-1], _p(weight), _p(dx), ld, rows, groups, s)
        dw = torch.empty_like(weight)
        db = torch.empty(groups * s, dtype=torch.float32, device=x.device)
        nws = _query("mnk_gconv1x1_workspace_floats", rows, groups, s)
        ws = SCRATCH.get("ws", nws, x)
        _call("mnk_gconv1x1_bwd_weight", x, _p(x), ld, _p(dy), dy.shape[-1], _p(dw), _p(db), rows, groups, s, _p(ws), nws)
        return dx, dw, db if has_bias else None, None


class Conv1x1SigmoidFn(_Fn):
    """1x1 conv writing (B,C,D,H,W) directly: refinement_module['conv-last'] + torch.sigmoid (generator.py:48,79-80;
    act=1) or the discriminator's linear score head (discriminator.py:59,77; act=0)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cin, b, act=1):
        _check_device(x)
        n, h, w, ld = x.shape
        d = n // b
        cout = weight.shape[0]
        out = torch.empty(b, cout, d, h, w, dtype=torch.float32, device=x.device)
        _call("mnk_conv1x1_fwd", x, _p(x), ld, cin, _p(weight), _p(bias), _p(out), b, d, h, w, cout, int(act))
        ctx.save_for_backward(x, weight, out)
        ctx.meta = (cin, b, d, bias is not None, int(act))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        cin, b, d, has_bias, act = ctx.meta
        dout = dout.contiguous()
        n, h, w, ld = x.shape
        cout = weight.shape[0]
        need_w = ctx.needs_input_grad[1] and not _SKIP_PARAM_GRADS[0]     # (no_param_grads: input gradients only)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if need_w else None
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if need_w else None
        if dx is None and dw is None:
            return None, None, None, None, None, None
        nws = _query("mnk_conv1x1_workspace_floats", n * h * w, cin, cout) if need_w else 0
        ws = SCRATCH.get("ws", nws, x) if need_w else None
        _call("mnk_conv1x1_bwd", x, _p(x), ld, cin, _p(weight), _p(out), _p(dout), _p(dx), ld, _p(dw), _p(db), b, d,
              h, w, cout, act, _p(ws), nws)
        return dx, dw, db if has_bias else None, None, None, None


# ----------------------------------------------------------------------------------------------------------------
# key-points
# ----------------------------------------------------------------------------------------------------------------
class SoftmaxKPFn(_Fn):
    """F.softmax(heatmap / T over H*W) + gaussian2kp 'matrix' (keypoint_detector.py:43-60,103-107)."""

    @staticmethod
    def forward(ctx, heat, k, temperature):
        _check_device(heat)
        n, h, w, ld = heat.shape
        dev = heat.device
        mean = torch.empty(n, k, 2, dtype=torch.float32, device=dev)
        var = torch.empty(n, k, 2, 2, dtype=torch.float32, device=dev)
        stat = torch.empty(n, k, 2, dtype=torch.float32, device=dev)
        _call("mnk_softmax_kp_fwd", heat, _p(heat), ld, n, h, w, k, float(temperature), _p(mean), _p(var), _p(stat))
        ctx.save_for_backward(heat, mean, stat)
        ctx.meta = (k, float(temperature))
        ctx.set_materialize_grads(False)     # backward makes the missing one itself (kp_variance given as a constant)
        return mean, var

    @staticmethod
    def backward(ctx, dmean, dvar):
        heat, mean, stat = ctx.saved_tensors
        k, temperature = ctx.meta
        n, h, w, ld = heat.shape
        dmean = dmean.contiguous() if dmean is not None else torch.zeros_like(mean)
        dvar = dvar.contiguous() if dvar is not None else torch.zeros(n, k, 2, 2, dtype=torch.float32, device=heat.device)
        dheat = torch.empty_like(heat)
        _call("mnk_softmax_kp_bwd", heat, _p(heat), ld, n, h, w, k, temperature, _p(mean), _p(stat), _p(dmean), _p(dvar),
              _p(dheat), ld)
        return dheat, None, None


def heatmap_argmax(heat, k):
    """(frames, K) int32: linear pixel index h * W + w of the largest heat-map logit per key point, first occurrence -- the
    integer form of the soft-argmax (keypoint_detector.py:103-104); `heat` is the key-point detector's logit act."""
    _check_device(heat)
    n, h, w, ld = heat.shape
    idx = torch.empty(n, k, dtype=torch.int32, device=heat.device)
    _call("mnk_heatmap_argmax", heat, _p(heat), ld, n, h, w, k, _p(idx))
    return idx


def kp_pixel_index(mean, size):
    """floor(size * (mean + 1) / 2) per axis, size = (W, H): the pixel the reference's Visualizer draws a key point at
    (logger.py:99-100).  mean (..., 2) fp32 -> (..., 2) int32."""
    _check_device(mean)
    m = mean.detach().contiguous().float()
    out = torch.empty(m.shape, dtype=torch.int32, device=m.device)
    _call("mnk_kp_pixel_index", m, _p(m), m.numel() // 2, int(size[0]), int(size[1]), _p(out))
    return out


class ClipVarianceFn(_Fn):
    """var * max(clip, sigma_min(var)) / sigma_min(var) on (..., 2, 2) covariances (keypoint_detector.py:62-65,
    modules/util.py:244-255) in one kernel instead of ~25 element-wise launches (+ ~60 in the backward)."""

    @staticmethod
    def forward(ctx, var, clip, mode="stable"):
        """mode: "stable" (default; sigma_min = |det| / sigma_max) | "reference" (the reference's own fp32 closed form
        sqrt((s1 - s2) / 2), modules/util.py:244-255 -- for parity work; it returns NaN on nearly singular covariances)"""
        _check_device(var)
        if mode not in ("stable", "reference"):
            raise ValueError("clip_variance_mode must be 'stable' or 'reference', got %r" % (mode,))
        var = var.contiguous().float()
        out = torch.empty_like(var)
        ctx.ref_mode = int(mode == "reference")
        _call("mnk_kp_clip_variance_fwd", var, _p(var), float(clip), var.numel() // 4, _p(out), ctx.ref_mode)
        ctx.save_for_backward(var)
        ctx.clip = float(clip)
        return out

    @staticmethod
    def backward(ctx, dout):
        var, = ctx.saved_tensors
        dout = dout.contiguous()
        dvar = torch.empty_like(var)
        _call("mnk_kp_clip_variance_bwd", var, _p(var), ctx.clip, var.numel() // 4, _p(dout), _p(dvar), ctx.ref_mode)
        return dvar, None, None


class MovementEmbeddingFn(_Fn):
    """MovementEmbeddingModule.forward (movement_embedding.py:42-92) -> act with kp-major channel order."""

    @staticmethod
    def forward(ctx, img, mean_d, var_d, mean_s, var_s, cfg):
        (b, d, h, w, k, cimg, add_bg, use_heatmap, use_difference, use_deformed, heatmap_diff, norm_const,
         const_var) = cfg
        ref = mean_d
        _check_device(ref)
        dev = ref.device
        mean_d = mean_d.contiguous().float()
        mean_s = mean_s.contiguous().float()
        var_d = var_d.contiguous().float() if var_d is not None else None
        var_s = var_s.contiguous().float() if var_s is not None else None
        slots = k + int(add_bg)
        per = int(use_heatmap) + 2 * int(use_difference) + (cimg if use_deformed else 0)
        ld_out = ceil4(slots * per)
        norm_d = norm_s = None
        norm_c = float(norm_const) if norm_const != "sum" else 0.0
        if norm_const == "sum" and use_heatmap:
            norm_d = torch.empty(b * d * k, dtype=torch.float32, device=dev)
            norm_s = torch.empty(b * k, dtype=torch.float32, device=dev)
            _call("mnk_gaussian_sums", ref, _p(mean_d), _p(var_d), const_var, b * d * k, h, w, _p(norm_d))
            _call("mnk_gaussian_sums", ref, _p(mean_s), _p(var_s), const_var, b * k, h, w, _p(norm_s))
        out = torch.empty(b * d, h, w, ld_out, dtype=torch.float32, device=dev)
        args = (_p(img), img.shape[-1] if img is not None else 0, cimg, _p(mean_d), _p(var_d), _p(mean_s), _p(var_s),
                float(const_var), b, d, h, w, k, int(add_bg), int(use_heatmap), int(use_difference), int(use_deformed),
                int(heatmap_diff), norm_c, _p(norm_d), _p(norm_s))
        _call("mnk_movement_embedding_fwd", ref, *args, _p(out), ld_out)
        ctx.save_for_backward(img, mean_d, var_d, mean_s, var_s, norm_d, norm_s)
        ctx.cfg = cfg
        ctx.norm_c = norm_c
        return out

    @staticmethod
    def backward(ctx, dout):
        img, mean_d, var_d, mean_s, var_s, norm_d, norm_s = ctx.saved_tensors
        (b, d, h, w, k, cimg, add_bg, use_heatmap, use_difference, use_deformed, heatmap_diff, norm_const,
         const_var) = ctx.cfg
        dout = dout.contiguous()
        dev = dout.device
        gmd = torch.empty(b * d, k, 2, dtype=torch.float32, device=dev)
        gms = torch.empty(b * d, k, 2, dtype=torch.float32, device=dev)
        gvd = gvs = None
        if var_d is not None:
            gvd = torch.empty(b * d, k, 4, dtype=torch.float32, device=dev)
            gvs = torch.empty(b * d, k, 4, dtype=torch.float32, device=dev)
        args = (_p(img), img.shape[-1] if img is not None else 0, cimg, _p(mean_d), _p(var_d), _p(mean_s), _p(var_s),
                float(const_var), b, d, h, w, k, int(add_bg), int(use_heatmap), int(use_difference), int(use_deformed),
                int(heatmap_diff), ctx.norm_c, _p(norm_d), _p(norm_s))
        _call("mnk_movement_embedding_bwd", dout, *args, _p(dout), dout.shape[-1], _p(gmd), _p(gvd), _p(gms), _p(gvs))
        g_mean_d = gmd.view(b, d, k, 2)
        g_mean_s = gms.view(b, d, k, 2).sum(dim=1, keepdim=True) if d > 1 else gms.view(b, 1, k, 2)
        g_var_d = g_var_s = None
        if gvd is not None:
            g_var_d = gvd.view(b, d, k, 2, 2)
            g_var_s = gvs.view(b, d, k, 2, 2).sum(dim=1, keepdim=True) if d > 1 else gvs.view(b, 1, k, 2, 2)
            if not use_heatmap:
                g_var_d = g_var_s = None
        return None, g_mean_d, g_var_d, g_mean_s, g_var_s, None


class MotionFieldKPFn(_Fn):
    """MotionFieldFn's mask form with kp_source.mean - kp_driving.mean formed inside the kernels (one driving frame per
    video): no subtraction / concatenation / zero-slot launches, and the two key-point gradients come out of the backward
    kernel.  mean_s, mean_d: (B,1,K,2)."""

    @staticmethod
    def forward(ctx, pred, mean_s, mean_d, k, use_corr):
        _check_device(pred)
        n, h, w, ld = pred.shape
        ms, md = mean_s.contiguous().float(), mean_d.contiguous().float()
        assert ms.numel() == n * k * 2 and md.numel() == n * k * 2
        field = torch.empty(n, h, w, 2, dtype=torch.float32, device=pred.device)
        _call("mnk_motion_field_kp_fwd", pred, _p(pred), ld, _p(ms), _p(md), n, h, w, k, int(use_corr), _p(field))
        ctx.save_for_backward(pred, ms, md)
        ctx.meta = (k, use_corr, mean_s.shape)
        return field

    @staticmethod
    def backward(ctx, dfield):
        pred, ms, md = ctx.saved_tensors
        k, use_corr, shape = ctx.meta
        n, h, w, ld = pred.shape
        dfield = dfield.contiguous()
        dpred = torch.empty_like(pred)
        g = torch.empty(2, *shape, dtype=torch.float32, device=pred.device)
        _call("mnk_motion_field_kp_bwd", pred, _p(pred), ld, _p(ms), _p(md), _p(dfield), n, h, w, k, int(use_corr), _p(dpred),
              ld, _p(g[0]), _p(g[1]))
        return dpred, g[0], g[1], None, None


class MotionFieldFn(_Fn):
    """mask softmax, sum_k m_k * delta_k + correction + identity grid (dense_motion_module.py:52-73) -> (N,h,w,2)."""

    @staticmethod
    def forward(ctx, pred, delta, k, use_mask, use_corr):
        _check_device(pred)
        n, h, w, ld = pred.shape
        delta = delta.contiguous().float() if delta is not None else None
        field = torch.empty(n, h, w, 2, dtype=torch.float32, device=pred.device)
        _call("mnk_motion_field_fwd", pred, _p(pred), ld, _p(delta), n, h, w, k, int(use_mask), int(use_corr), _p(field))
        ctx.save_for_backward(pred, delta)
        ctx.meta = (k, use_mask, use_corr)
        return field

    @staticmethod
    def backward(ctx, dfield):
        pred, delta = ctx.saved_tensors
        k, use_mask, use_corr = ctx.meta
        n, h, w, ld = pred.shape
        dfield = dfield.contiguous()
        dpred = torch.empty_like(pred)
        ddelta = torch.empty(n, k + 1, 2, dtype=torch.float32, device=pred.device)
        _call("mnk_motion_field_bwd", pred, _p(pred), ld, _p(delta), _p(dfield), n, h, w, k, int(use_mask), int(use_corr),
              _p(dpred), ld, _p(ddelta))
        return dpred, (ddelta if delta is not None else None), None, None, None


class WarpSkipFn(_Fn):
    """deform_input (generator.py:51-58) of one skip tensor, with the nearest-resized key-point embedding written
    behind it in the same buffer (generator.py:72-73): out = [warp(inp, field) | resize(emb)]."""

    @staticmethod
    def forward(ctx, inp, field, emb, c, ke, mode):
        _check_device(inp)
        n, h, w, ld_in = inp.shape
        _, hf, wf, _ = field.shape
        # c a multiple of 4: the warp writes whole channel quads and the resized embedding, copied with its (zero) pad
        # channels, the rest of every pixel row -- no zero fill of the output
        whole = c % 4 == 0 and (emb is None or emb.shape[-1] == ceil4(ke))
        out = (torch.empty if whole else torch.zeros)(n, h, w, ceil4(c + ke), dtype=torch.float32, device=inp.device)
        _call("mnk_deform_fwd", inp, _p(inp), ld_in, c, h, w, _p(field), hf, wf, mode, _p(out), out.shape[-1], 0, n)
        if emb is not None:
            _call("mnk_resize_nearest" if mode == 0 else "mnk_resize_bilinear", inp, _p(emb), emb.shape[-1], emb.shape[1], emb.shape[2], _p(out), out.shape[-1], c,
                  h, w, n, ceil4(ke) if whole else ke)
        ctx.save_for_backward(inp, field, emb)
        ctx.meta = (c, ke, mode)
        return out

    @staticmethod
    def backward(ctx, dout):
        inp, field, emb = ctx.saved_tensors
        c, ke, mode = ctx.meta
        dout = dout.contiguous()
        n, h, w, ld_in = inp.shape
        _, hf, wf, _ = field.shape
        dinp = torch.empty_like(inp) if ctx.needs_input_grad[0] else None       # written by the gather pass
        dfield = torch.zeros_like(field) if ctx.needs_input_grad[1] else None    # added to
        if dinp is not None or dfield is not None:
            nws = _query("mnk_deform_bwd_workspace_floats", c, h, w, n)
            _call("mnk_deform_bwd", inp, _p(inp), ld_in, c, h, w, _p(field), hf, wf, mode, _p(dout), dout.shape[-1], 0,
                  _p(dinp), _p(dfield), n, _p(SCRATCH.get("ws", nws, inp)), nws)
        demb = None
        if emb is not None and ctx.needs_input_grad[2]:
            # nearest: the adjoint is a gather that writes every source pixel; with c a multiple of 4 it also reads the
            # (zero) pad channels of dout and so writes demb's pads itself
            whole = mode == 0 and c % 4 == 0 and emb.shape[-1] == ceil4(ke) and dout.shape[-1] == c + emb.shape[-1]
            demb = torch.empty_like(emb) if whole else torch.zeros_like(emb)
            _call("mnk_resize_nearest_bwd" if mode == 0 else "mnk_resize_bilinear_bwd", inp, _p(dout), dout.shape[-1], c, h, w, _p(demb), emb.shape[-1], emb.shape[1],
                  emb.shape[2], n, emb.shape[-1] if whole else ke)
        return dinp, dfield, demb, None, None, None


WARP_LEVEL = np.dtype([("inp", "<u8"), ("out", "<u8"), ("dout", "<u8"), ("dinp", "<u8"), ("ld_in", "<i4"), ("C", "<i4"),
                       ("h", "<i4"), ("w", "<i4"), ("ld_out", "<i4"), ("ke", "<i4"), ("emb_off", "<i4"), ("reserved", "<i4")])


class WarpAllFn(_Fn):
    """Every deform_input of one generator forward (generator.py:66-73 the skips, :78 the source frame) as ONE autograd node.
    They all read the same deformation field, so as separate nodes each backward zero-fills its own field gradient and
    autograd adds the eight of them up; here one gather per field texel sums the levels in order (mnk_warp_levels_bwd:
    deterministic, nothing is zero-filled).  forward(field, emb, mode, specs, *inps) with specs = ((c, ke), ...) per input; returns one
    [warp(inp) | resize(emb)] act per input (ke = 0: no embedding behind it)."""

    @staticmethod
    def forward(ctx, field, emb, mode, specs, *inps):
        _check_device(field)
        _, hf, wf, _ = field.shape
        outs = []
        ctx.multi = len(inps) <= 12 and knobs.form("WARP_LEVELS")
        if ctx.multi:         # one launch for the warps and embedding copies of all levels (mnk_warp_levels_fwd)
            lv = np.zeros(len(inps), dtype=WARP_LEVEL)
            for i, (inp, (c, ke)) in enumerate(zip(inps, specs)):
                n, h, w, ld_in = inp.shape
                e = emb if ke else None
                # (the launch writes every channel of every row -- warp, embedding, pad channels: no zero fill)
                out = torch.empty(n, h, w, ceil4(c + ke), dtype=torch.float32, device=inp.device)
                lv[i] = (inp.data_ptr(), out.data_ptr(), 0, 0, ld_in, c, h, w, out.shape[-1], ke if e is not None else 0, c, 0)
                outs.append(out)
            _call("mnk_warp_levels_fwd", field, lv.ctypes.data, len(inps), _p(field), hf, wf, mode, _p(emb),
                  emb.shape[-1] if emb is not None else 0, emb.shape[1] if emb is not None else 0,
                  emb.shape[2] if emb is not None else 0, inps[0].shape[0])
            ctx.save_for_backward(field, emb, *inps)
            ctx.meta = (mode, tuple(specs))
            ctx.set_materialize_grads(False)
            return tuple(outs)
        for inp, (c, ke) in zip(inps, specs):
            n, h, w, ld_in = inp.shape
            e = emb if ke else None
            whole = c % 4 == 0 and (e is None or e.shape[-1] == ceil4(ke))      # see WarpSkipFn.forward
            out = (torch.empty if whole else torch.zeros)(n, h, w, ceil4(c + ke), dtype=torch.float32, device=inp.device)
            _call("mnk_deform_fwd", inp, _p(inp), ld_in, c, h, w, _p(field), hf, wf, mode, _p(out), out.shape[-1], 0, n)
            if e is not None:
                _call("mnk_resize_nearest" if mode == 0 else "mnk_resize_bilinear", inp, _p(e), e.shape[-1], e.shape[1],
                      e.shape[2], _p(out), out.shape[-1], c, h, w, n, ceil4(ke) if whole else ke)
            outs.append(out)
        ctx.save_for_backward(field, emb, *inps)
        ctx.meta = (mode, tuple(specs))
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        field, emb = ctx.saved_tensors[:2]
        inps = ctx.saved_tensors[2:]
        mode, specs = ctx.meta
        _, hf, wf, _ = field.shape
        # the gradients of all warps (d input of every level, d field) are slices of ONE buffer.  The acts come first (their
        # sizes are multiples of 4 floats: every slice stays 16-byte aligned for the float4 readers downstream), the field
        # last.  The one-launch form writes every element; the per-level form adds the levels' field gradients up.
        want_field = ctx.needs_input_grad[0] and any(d is not None for d in douts)
        want = [ctx.needs_input_grad[4 + i] and douts[i] is not None for i in range(len(inps))]
        sizes = [inp.numel() if wnt else 0 for inp, wnt in zip(inps, want)] + [field.numel() if want_field else 0]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=field.device) if sum(sizes) else None
        offs = [sum(sizes[:k]) for k in range(len(sizes))]
        dfield = flat[offs[-1]:offs[-1] + sizes[-1]].view_as(field) if want_field else None
        demb = None
        dinps = []
        if ctx.multi and any(d is not None for d in douts):
            live = [i for i, d in enumerate(douts) if d is not None]        # (a level whose output went nowhere: no gradient)
            lv = np.zeros(len(live), dtype=WARP_LEVEL)
            keep = []
            want_emb = emb is not None and ctx.needs_input_grad[1] and any(specs[i][1] for i in live)
            dinps = [None] * len(inps)
            for j, i in enumerate(live):
                inp, (c, ke) = inps[i], specs[i]
                dout = douts[i].contiguous()
                keep.append(dout)
                n, h, w, ld_in = inp.shape
                dinps[i] = flat[offs[i]:offs[i] + sizes[i]].view_as(inp) if want[i] else None
                lv[j] = (inp.data_ptr(), 0, dout.data_ptr(), dinps[i].data_ptr() if dinps[i] is not None else 0, ld_in, c, h, w,
                         dout.shape[-1], ke if want_emb else 0, c, 0)
            if want_emb:
                demb = torch.empty_like(emb)
            nws = _query("mnk_warp_levels_bwd_workspace_floats", lv.ctypes.data, len(live), inps[0].shape[0])
            _call("mnk_warp_levels_bwd", field, lv.ctypes.data, len(live), _p(field), hf, wf, mode, _p(dfield), _p(demb),
                  emb.shape[-1] if emb is not None else 0, emb.shape[1] if emb is not None else 0,
                  emb.shape[2] if emb is not None else 0, inps[0].shape[0], _p(SCRATCH.get("ws", nws, field)), nws)
            return (dfield, demb, None, None) + tuple(dinps)
        if dfield is not None:
            dfield.zero_()
        for i, (inp, (c, ke), dout) in enumerate(zip(inps, specs, douts)):
            if dout is None:
                dinps.append(None)
                continue
            dout = dout.contiguous()
            n, h, w, ld_in = inp.shape
            dinp = flat[offs[i]:offs[i] + sizes[i]].view_as(inp) if want[i] else None
            if dinp is not None or dfield is not None:
                nws = _query("mnk_deform_bwd_workspace_floats", c, h, w, n)
                _call("mnk_deform_bwd", inp, _p(inp), ld_in, c, h, w, _p(field), hf, wf, mode, _p(dout), dout.shape[-1], 0,
                      _p(dinp), _p(dfield), n, _p(SCRATCH.get("ws", nws, inp)), nws)
            dinps.append(dinp)
            if ke and emb is not None and ctx.needs_input_grad[1]:
                # the first contribution writes demb (nearest: a gather over the source pixels that, with c a multiple of 4,
                # also copies the zero pad channels of dout), the others are added by the kernels
                whole = mode == 0 and c % 4 == 0 and emb.shape[-1] == ceil4(ke) and dout.shape[-1] == c + emb.shape[-1]
                first = demb is None
                if first:
                    demb = torch.empty_like(emb) if whole else torch.zeros_like(emb)
                name = ("mnk_resize_nearest_bwd" if first else "mnk_resize_nearest_bwd_accumulate") if mode == 0 \
                    else "mnk_resize_bilinear_bwd"                  # (bilinear: added to the zeroed buffer by a gather)
                _call(name, inp, _p(dout), dout.shape[-1], c, h, w, _p(demb), emb.shape[-1], emb.shape[1], emb.shape[2], n,
                      emb.shape[-1] if whole else ke)
        return (dfield, demb, None, None) + tuple(dinps)


# convenience wrappers ---------------------------------------------------------------------------------------------
def to_act(x5, step=1):
    if isinstance(x5, StackedBatch):
        return ToActPairFn.apply(x5.parts[0], x5.parts[1], step)
    return ToActFn.apply(x5, int(step))


def from_act(a, c, b):
    return FromActFn.apply(a, c, b)


def step_from_scale(scale_factor):
    if scale_factor == 1:
        return 1
    step = int(round(1.0 / scale_factor))
    if abs(step * scale_factor - 1.0) > 1e-6:
        raise NotImplementedError("only scale factors 1/integer are used by the reference configs; got %r" % scale_factor)
    return step
