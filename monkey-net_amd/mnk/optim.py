"""The optimiser side of a training iteration on the gfx950 kernels: what
torch.optim.Adam(params, lr, betas=(0.5, 0.999)) + the implicit gradient reductions are in train.py:81-83,118-136.

`MnkAdam` is a torch.optim.Optimizer (param_groups, state_dict, lr schedulers keep working) whose step is ONE kernel
launch over a descriptor table (mnk_adam_multi) that also writes the packed GEMM layouts of every 3x3 convolution weight
for the next iteration.  It owns three flat fp32 buffers -- gradients, exp_avg, exp_avg_sq -- with one 16-byte aligned
slice per parameter:

* gradients land in their slice ("sink") without copies: the weight-gradient GEMMs of the convolutions run with
  MNK_WGRAD_DEFER and leave their pixel-split partials in per-layer buffers; `materialize_grads()` reduces the partials
  of ALL layers in one launch (mnk_wgrad_reduce_multi) straight into the sinks.  The small tensors that autograd hands to
  `p.grad` (normalisation scales, biases, 1x1 weights) are gathered with one multi-tensor copy;
* the flat gradient buffer is what the data-parallel exchange all-reduces (RCCL over xGMI): no bucket concatenation, no
  copy back; the division by the world size is folded into the update (`grad_scale`);
* learning rate, step count and bias corrections live in device memory, so a captured hipGraph follows a scheduler.
"""
import ctypes
import weakref

import numpy as np
import torch

from . import _lib
from . import dist as mdist
from . import ops as mops

ADAM_DESC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("wp_fwd", "<u8"),
                      ("wp_d0", "<u8"), ("wp_d1", "<u8"), ("Cout", "<i4"), ("C0", "<i4"), ("C1", "<i4"),
                      ("block_begin", "<i4"), ("flags", "<i4"), ("reserved", "<i4"), ("gt0", "<u8"), ("gt1", "<u8"),
                      ("gt_splits0", "<i4"), ("gt_splits1", "<i4")])
REDUCE_DESC = np.dtype([("part", "<u8"), ("dw", "<u8"), ("layout", "<i4"), ("splits", "<i4"), ("ntaps", "<i4"),
                        ("Cout", "<i4"), ("C", "<i4"), ("Cin_total", "<i4"), ("c_start", "<i4"), ("accumulate", "<i4"),
                        ("block_begin", "<i4"), ("reserved", "<i4")])


JOB = np.dtype([("x", "<u8"), ("dy", "<u8"), ("part", "<u8"), ("part_floats", "<u8"), ("ld_x", "<i4"), ("C", "<i4"),
                ("flags", "<i4"), ("ld_dy", "<i4"), ("Cout", "<i4"), ("N", "<i4"), ("Ho", "<i4"), ("Wo", "<i4"), ("Hi", "<i4"),
                ("Wi", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("pad", "<i4"), ("variant", "<i4"), ("splits", "<i4"),
                ("reserved", "<i4")])


class _Plan(ctypes.Structure):
    _fields_ = [("layout", ctypes.c_int), ("splits", ctypes.c_int), ("part_floats", ctypes.c_size_t)]


def _device_table(rec, device, keep):
    t = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(device)
    keep.append(t)          # captured graphs hold raw pointers to earlier tables: never free them
    return t


_BG_STREAMS = {}


def _background_stream(device):
    s = _BG_STREAMS.get(device)
    if s is None:
        s = _BG_STREAMS[device] = torch.cuda.Stream(device=device)
    return s


def _capturing(device):
    return device.type == "cuda" and torch.cuda.is_current_stream_capturing()


class DeferredReducer:
    """The weight-gradient GEMMs of one optimiser's convolutions, run as late and as together as possible:
    * tap-major shapes (every layer wider than one 64-channel tile) are only RECORDED during backward and launched
      together at flush time (mnk_wgrad_grouped_*: one launch per tile shape; the tiles of all layers fill the chip, so
      a layer is split along pixels only where its pixel range is long) -- `x` and `dy` stay alive until then;
    * the other shapes (nine-tap 16x16 kernel, LDS-halo, gather) launch at once with MNK_WGRAD_DEFER;
    * ONE mnk_wgrad_reduce_multi launch then reduces the partials of all layers into the optimiser's flat buffer.
    Memory: every layer keeps its partial buffer between iterations (captured graphs hold their addresses): 0.87 GB for the
    generator + key-point detector of BASELINE configs[1] at batch 32, on top of the 3 x 4 B per parameter of the flat
    gradient / exp_avg / exp_avg_sq buffers; the operands (x, dy) of the recorded GEMMs stay allocated until flush() -- the
    activations of a backward pass are not freed layer by layer as autograd would free them (a few hundred MB at batch 32 @ 64^2;
    sized for the 288 GB of an MI355X, not for a small card)."""

    def __init__(self, owner):
        self.owner = owner
        self.recs = {}            # (id(weight), c_start) -> dict(shape, part, row of REDUCE_DESC without block_begin, ...)
        self.pending = []         # keys with partials waiting for the reduction, in launch order
        self.jobs = []            # recorded grouped jobs: (key, x, dy, flags)
        self.done = set()         # keys launched / recorded since zero_grad
        self.tables = {}          # tuple(pending keys) -> (device table, n, blocks)
        self.keep = []
        # background launches (MNK_WGRAD_BG = giga-MACs per launch group; 0: everything at flush time) of an EAGER
        # backward: the recorded GEMMs go to a second stream as soon as enough of them are there, and run under the
        # launch-bound normalisation / element-wise passes of the layers further down the chain (MI355X: 12.54 -> 11.40 ms
        # per eager iteration).  Not while capturing: a hipGraph with a second branch replays 0.6 ms SLOWER than the
        # linear one (11.4-11.6 against 10.78 ms, profiles/r03_knob_ab_log.txt), so captured iterations stay linear
        self.bg_macs = float(mops.knobs.get("MNK_WGRAD_BG") or 0) * 1e9
        self.job_macs = 0.0
        self.inflight = []        # operands / tables of launches on the background stream since the last join
        self.direct = ()          # keys (since the last drop) whose partials the optimiser kernel reads itself (MnkAdam.tap_direct)
        self.multi = set()        # keys of weights that were seen to receive two contributions before one step: never direct

    def _record(self, key, weight, sink, shape, flags):
        n, ho, wo, c, cout, kh, kw, pad, ld_x, cin_total, hi, wi, ld_dy, c_start = shape
        if _capturing(weight.device):
            raise RuntimeError("a convolution shape first seen during hipGraph capture (run one eager iteration first)")
        lib = _lib.lib()
        job = np.zeros(1, dtype=JOB)
        job[0] = (0, 0, 0, 0, ld_x, c, flags, ld_dy, cout, n, ho, wo, hi, wi, kh, kw, pad, 0, 0, 0)
        grouped = True
        if grouped:
            if lib.query("mnk_wgrad_grouped_plan", job.ctypes.data, 1) != 0:
                raise _lib.MnkError("mnk_wgrad_grouped_plan failed: %s" % lib.cdll.mnk_last_error().decode())
            grouped = int(job["variant"][0]) >= 0
        if grouped:       # variant % 4 == 3: the sub-pixel form of an up-sampled layer (16 pseudo taps, layout 2);
            v = int(job["variant"][0])                  # variants >= 16: the nine-tap 16x16 kernel (tap-major partials too)
            layout = 2 if (v < 16 and v % 4 == 3) else 0
            splits, nfloats = int(job["splits"][0]), int(job["part_floats"][0])
        else:
            plan = _Plan()
            if lib.query("mnk_conv2d_wgrad_plan2", n, ho, wo, c, cout, kh, kw, pad, ld_x, flags, ctypes.byref(plan)) != 0:
                raise _lib.MnkError("mnk_conv2d_wgrad_plan2 failed: %s" % lib.cdll.mnk_last_error().decode())
            layout, splits, nfloats = int(plan.layout), int(plan.splits), int(plan.part_floats)
        part = torch.empty(max(nfloats, 1), dtype=torch.float32, device=weight.device)
        rec = {"shape": shape, "part": part, "splits": splits, "nfloats": nfloats, "grouped": grouped, "job": job,
               "direct_ok": mops.pack_entry_of(weight) is not None and c_start in (0, mops.pack_entry_of(weight).meta[1]),
               "row": (part.data_ptr(), sink.data_ptr(), layout, splits, kh * kw, cout, c, cin_total, c_start, 0),
               "blocks": lib.query("mnk_wgrad_reduce_blocks", splits, cout, c) if splits > 0 else 0}
        self.recs[key] = rec
        self.tables.clear()
        return rec

    def wgrad(self, weight, x, ld_x, c, flags, hi, wi, kh, kw, pad, dy, ld_dy, cout, cin_total, c_start, n, ho, wo):
        """d(weight)[:, c_start:c_start+c] towards the owner's sink of `weight`; True when taken.
        INVARIANT (background launches, MNK_WGRAD_BG): the recorded operands `x` and `dy` are read by GEMMs that may run on a
        second stream while the backward pass continues on the main one -- nothing later in that backward pass may write
        these two buffers in place (no kernel of this package does: every epilogue writes a fresh tensor), and they stay
        referenced until _join() has made the main stream wait for the side stream (flush() / drop()).
        tests/test_step.py::test_background_weight_gradients_equal_the_in_order_ones pins bit-equality of the two orders."""
        own = self.owner
        sink = own.sink(weight)
        key = (id(weight), c_start)
        if key in self.done:
            return False                                   # a second contribution before the step: the caller accumulates
        if x.data_ptr() % 16 or dy.data_ptr() % 16:
            return False
        shape = (n, ho, wo, c, cout, kh, kw, pad, ld_x, cin_total, hi, wi, ld_dy, c_start)
        rec = self.recs.get(key)
        if rec is None or rec["shape"] != shape:
            rec = self._record(key, weight, sink, shape, int(flags))
        if rec["grouped"]:
            self.jobs.append((key, x, dy))                 # launched by flush(); the operands stay alive until then
            self.job_macs += float(n * ho * wo) * c * cout * kh * kw
            if self.bg_macs > 0 and self.job_macs >= self.bg_macs and x.is_cuda and not _capturing(x.device):
                self._launch_grouped(background=True)
        else:
            mops._call("mnk_conv2d_wgrad", dy, mops._p(x), ld_x, c, int(flags) | 4, hi, wi, kh, kw, pad, mops._p(dy), ld_dy,
                       cout, mops._p(sink), cin_total, c_start, n, ho, wo, mops._p(rec["part"]), rec["nfloats"])
        if rec["splits"] > 0:
            self.pending.append(key)
        self.done.add(key)
        own._written.add(id(weight))
        return True

    def is_direct(self, key):
        """The owner's optimiser kernel reads this layer's gradient straight from its tap-major partials (MnkAdam.tap_direct:
        few-split layers of a grouped GEMM whose weight the kernel updates tile by tile) -- no reduction for it."""
        if not self.owner.tap_direct:
            return False
        r = self.recs.get(key)
        # C % 4 == 0: only the layers mnk_wgrad_reduce_multi sums with its flat map -- (s0 + s1) + s2 in split order, the order
        # the optimiser kernel's direct read uses; the tile map of the other layers adds (s0 + s2) + s1, an ulp apart
        return (r is not None and r["grouped"] and 1 <= r["splits"] <= 3 and r["row"][2] in (0, 2) and r["row"][4] == 9
                and r["row"][6] % 4 == 0 and r["direct_ok"] and key not in self.multi)

    def _launch_grouped(self, background=False):
        jobs, self.jobs = self.jobs, []
        self.job_macs = 0.0
        lib = _lib.lib()
        dev = jobs[0][1].device
        arr = np.zeros(len(jobs), dtype=JOB)
        for i, (key, x, dy) in enumerate(jobs):
            rec = self.recs[key]
            arr[i] = rec["job"][0]
            arr["x"][i], arr["dy"][i], arr["part"][i] = x.data_ptr(), dy.data_ptr(), rec["part"].data_ptr()
        nbytes = lib.query("mnk_wgrad_grouped_table_bytes", len(jobs))
        host = np.zeros((nbytes + 15) // 16 * 16, dtype=np.uint8)
        if lib.query("mnk_wgrad_grouped_build", arr.ctypes.data, len(jobs), host.ctypes.data, nbytes) != 0:
            raise _lib.MnkError("mnk_wgrad_grouped_build failed: %s" % lib.cdll.mnk_last_error().decode())
        # the table travels as kernel arguments: capturable as it is (frozen into the graph), no page-locked staging
        table = torch.empty(host.size, dtype=torch.uint8, device=dev)
        if _capturing(dev):
            self.keep.append(table)
        if background:
            main, side = torch.cuda.current_stream(dev), _background_stream(dev)
            side.wait_stream(main)                       # the operands of the recorded jobs are complete on `main`
            self.inflight.append((jobs, table))          # ... and stay allocated until `main` has waited for `side`
            with torch.cuda.stream(side):
                mops._call("mnk_table_upload", table, host.ctypes.data, mops._p(table), host.size)
                mops._call("mnk_wgrad_grouped_launch", table, mops._p(table), host.ctypes.data)
            return
        if dev.type == "cuda":
            mops._call("mnk_table_upload", table, host.ctypes.data, mops._p(table), host.size)
        else:
            table.copy_(torch.from_numpy(host))
        mops._call("mnk_wgrad_grouped_launch", table, mops._p(table), host.ctypes.data)

    def _join(self):
        if self.inflight:
            dev = self.inflight[0][1].device
            torch.cuda.current_stream(dev).wait_stream(_background_stream(dev))
            self.inflight = []

    def flush(self):
        """The recorded GEMMs in a few grouped launches, then one launch reducing every pending layer's partials."""
        if self.jobs:
            self._launch_grouped()
        self._join()
        # (a second flush before the step -- TrainStep materialises the gradients, then steps -- finds nothing pending and must
        # leave the set alone: it lives until drop())
        if self.owner.tap_direct and self.pending:
            now = tuple(k for k in self.pending if self.is_direct(k))
            self.direct = self.direct + now
            self.pending = [k for k in self.pending if k not in now]
        if not self.pending:
            return 0
        # the layers with the longest per-block chains (most splits) first: the blocks of a launch start in table order, and a
        # 497-split layer placed last (the discriminator's first convolution: 64 blocks of 31 sequential partial sums) was the
        # tail of the whole launch.  Every layer's sum is its own: the order of the table changes no result
        keys = tuple(sorted(self.pending, key=lambda k: -self.recs[k]["splits"]))
        tab = self.tables.get(keys)
        dev = self.recs[keys[0]]["part"].device
        if tab is None:
            if _capturing(dev):
                raise RuntimeError("the set of deferred weight gradients changed during hipGraph capture")
            rec = np.zeros(len(keys), dtype=REDUCE_DESC)
            blocks = 0
            for i, k in enumerate(keys):
                r = self.recs[k]
                rec[i] = r["row"] + (blocks, 0)
                blocks += r["blocks"]
            tab = (_device_table(rec, dev, self.keep), len(keys), blocks)
            self.tables[keys] = tab
        mops._call("mnk_wgrad_reduce_multi", tab[0], mops._p(tab[0]), tab[1], tab[2])
        n = len(self.pending)
        self.pending = []
        return n

    def undirect(self, weight):
        """A second contribution to `weight` arrives before the step (MnkAdam.add_to_sink): the first one must be in the sink
        after all -- reduce the partials the optimiser kernel was going to read itself, now and from now on."""
        keys = tuple(k for k in self.direct if k[0] == id(weight))
        self.multi.update(k for k in self.recs if k[0] == id(weight))
        if not keys:
            return
        self.direct = tuple(k for k in self.direct if k not in keys)
        rec = np.zeros(len(keys), dtype=REDUCE_DESC)
        blocks = 0
        for i, k in enumerate(keys):
            r = self.recs[k]
            rec[i] = r["row"] + (blocks, 0)
            blocks += r["blocks"]
        dev = self.recs[keys[0]]["part"].device
        tab = _device_table(rec, dev, self.keep)
        mops._call("mnk_wgrad_reduce_multi", tab, mops._p(tab), len(keys), blocks)

    def drop(self):
        self._join()
        self.direct = ()
        self.pending = []
        self.jobs = []
        self.job_macs = 0.0
        self.done.clear()


class FlatGrads:
    """The gradient side of a parameter set without an optimiser attached: one flat fp32 buffer with a 16-byte aligned slice
    ("sink") per parameter, the DeferredReducer that lands the convolution weight gradients there, and `materialize_grads()`,
    after which every `p.grad` IS its slice.  MnkAdam is this plus the update; mnk.dropin.TrainPairRunner owns one per network
    when the loop's optimisers are stock torch.optim objects (the reference's train.py:81-83)."""

    def _init_sinks(self, ps):
        ps = [p for p in ps if p.requires_grad]
        if not ps:
            raise ValueError("no trainable parameters")
        dev = ps[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise ValueError("fp32 parameters on one device are needed")
        mops._check_device(ps[0])
        self._params = ps
        self.device = dev
        off, self._off = 0, {}
        for p in ps:
            self._off[id(p)] = off
            off += (p.numel() + 3) // 4 * 4                  # 16-byte aligned slices: float4 path of the kernel
        self.numel = off
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self._sinks = {id(p): self.flat_grad[self._off[id(p)]:self._off[id(p)] + p.numel()].view_as(p) for p in ps}
        self.reducer = DeferredReducer(self)
        self._written = set()            # ids of parameters whose sink was written by a kernel since zero_grad
        # True: the step kernel takes the gradients of the few-split tap-major layers (the deep levels: 330 of the generator's
        # 690 MB of partials ARE the gradient) straight from the partials, and mnk_wgrad_reduce_multi skips them -- `p.grad` of
        # those parameters is then NOT valid.  mnk.engine.TrainStep switches it on for a captured iteration of one process
        # (nobody can look at p.grad between the launches of a replay; several ranks exchange the flat buffer)
        self.tap_direct = False
        for p in ps:
            mops.register_grad_sink(p, self)
        weakref.finalize(self, mops.unregister_grad_sinks, [id(p) for p in ps], id(self))

    def sink(self, p):
        return self._sinks[id(p)]

    def add_to_sink(self, p, grad):
        """slow path: a further contribution to a parameter whose sink already holds one (a parameter used by two
        backward passes before one step, e.g. train_params['detach_kp_discriminator'] = False)."""
        self.materialize_grads()
        self.reducer.undirect(p)
        self._sinks[id(p)].add_(grad)
        self._written.add(id(p))

    def materialize_grads(self):
        """After backward: every gradient of this set's parameters in its slice of the flat buffer, `p.grad`
        pointing at it.  Idempotent (a second call finds nothing pending and every p.grad already in place)."""
        self.reducer.flush()
        dsts, srcs = [], []
        for p in self._params:
            s = self._sinks[id(p)]
            gr = p.grad
            if gr is None:
                if id(p) in self._written:
                    p.grad = s
                continue
            if gr.data_ptr() == s.data_ptr():
                continue
            if id(p) in self._written:       # kernel-written part + an autograd part (never in the shipped configs)
                s.add_(gr)
            else:
                dsts.append(s)
                srcs.append(gr.reshape(s.shape) if gr.shape != s.shape else gr)
                self._written.add(id(p))
            p.grad = s
        if dsts:
            torch._foreach_copy_(dsts, srcs)

    def begin_pass(self):
        """a new backward pass will write the sinks: forget what the last one recorded (what zero_grad() of an owning optimiser does,
        without touching p.grad -- a stock optimiser's zero_grad() has already set it to None)"""
        self.reducer.drop()
        self._written.clear()


class GradSinks(FlatGrads):
    """FlatGrads of a network whose optimiser is not ours (see FlatGrads)."""

    def __init__(self, params):
        self._init_sinks(list(params))
        self._mnk_owns_exchange = False


class MnkAdam(torch.optim.Optimizer, FlatGrads):
    """torch.optim.Adam(params, lr, betas, eps) semantics (no amsgrad / weight decay), one kernel launch per step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super(MnkAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError("MnkAdam takes one parameter group (train.py:81-83 builds one optimiser per network)")
        self._init_sinks(self.param_groups[0]["params"])
        ps, dev, off = self._params, self.device, self.numel
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self.hyper = torch.tensor([g["lr"], b1, b2, g["eps"], 0.0, 0.0, 1.0, 0.0, 1.0 - b1, 1.0 - b2],
                                  dtype=torch.float64).float().to(dev)
        self._lr_on_device = float(g["lr"])
        self._gscale_on_device = 1.0
        self.steps_taken = 0
        self._mnk_owns_exchange = True   # mnk.dist's generic pre-step gradient averaging skips this optimiser
        self._table = None               # (key, device table, n, blocks, entries)
        self._keep = []
        self._mnk_fresh_entries = ()
        self._exchange = None            # a gradient exchange started by begin_exchange(), finished by step()
        self._exchanged = False          # ... or already finished by exchange_end(): step() must not sum again

    def zero_grad(self, set_to_none=True):
        if self._exchange is not None:       # (never in TrainStep: an exchange is always consumed by the step)
            mdist.all_reduce_flat_end(self._exchange)
            self._exchange = None
        for p in self._params:
            p.grad = None
        self._exchanged = False
        self.reducer.drop()
        self._written.clear()

    # ---- the step ----------------------------------------------------------------------------------------------------
    def sync_scalars(self, grad_scale=None):
        """push a changed learning rate / gradient scale to the device scalars (outside a captured graph)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:
            self.hyper[0:1].copy_(torch.tensor([lr], dtype=torch.float32))
            self._lr_on_device = lr
        if grad_scale is not None and float(grad_scale) != self._gscale_on_device:
            self.hyper[6:7].copy_(torch.tensor([float(grad_scale)], dtype=torch.float32))
            self._gscale_on_device = float(grad_scale)

    def _build_table(self, active):
        rows, blocks, entries = [], 0, []
        for p in active:
            o = self._off[id(p)]
            g, m, v = self.flat_grad[o:], self.flat_m[o:], self.flat_v[o:]
            e = mops.pack_entry_of(p)
            if e is not None:
                cout, c0, c1, up = e.meta
                nb = _lib.lib().query("mnk_adam_blocks", 0, cout, c0, c1, 1)
                gt, flags = [(0, 0), (0, 0)], int(up)
                for si, cs in enumerate((0, c0) if c1 else (0,)):
                    key = (id(p), cs)
                    if key in self.reducer.direct:
                        r = self.reducer.recs[key]
                        gt[si] = (r["part"].data_ptr(), r["splits"])
                        flags |= (2 << si) if r["row"][2] == 2 else 0
                rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), e.wp.data_ptr(),
                             e.wd[0].data_ptr() if e.wd[0] is not None else 0,
                             e.wd[1].data_ptr() if e.wd[1] is not None else 0, cout, c0, c1, blocks, flags, 0,
                             gt[0][0], gt[1][0], gt[0][1], gt[1][1]))
                entries.append(e)
            else:
                nb = _lib.lib().query("mnk_adam_blocks", p.numel(), 0, 0, 0, 0)
                rows.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), 0, 0, 0, 0, 0, 0, blocks, 0, 0,
                             0, 0, 0, 0))
            blocks += nb
        rec = np.array(rows, dtype=ADAM_DESC)
        return _device_table(rec, self.device, self._keep), len(rows), blocks, tuple(entries)

    @torch.no_grad()
    def begin_exchange(self):
        """After backward: put every gradient in place and START the sum over the ranks without blocking the kernels' stream;
        step() waits for it.  Work that does not depend on this optimiser's update (the discriminator-loss backward of
        mnk.engine.TrainStep) runs in between and hides the exchange."""
        self.materialize_grads()
        if self._exchange is None and mdist.grads_active() and any(p.grad is not None for p in self._params):
            self._exchange = mdist.all_reduce_flat_begin(self.flat_grad)

    def exchange_begin(self):
        """The bare start of the exchange (no look at the parameters: this is also what runs between two captured
        hipGraphs at replay time, when no Python-side gradient state exists); exchange_end() is its other half."""
        if mdist.grads_active():
            self._exchange = mdist.all_reduce_flat_begin(self.flat_grad)

    def exchange_end(self):
        if self._exchange is not None:
            mdist.all_reduce_flat_end(self._exchange)
            self._exchange = None
            self._exchanged = True            # the next step() finds the sums in place

    def replay_step(self):
        """The launches of the last step() again, from its descriptor table: the update of a captured iteration whose
        optimiser step is NOT part of a hipGraph (mnk.engine.TrainStep with an overlapped exchange)."""
        if self._table is None:
            raise RuntimeError("replay_step() before any step()")
        _, tab, n, blocks, entries = self._table
        mops._call("mnk_adam_tick", self.hyper, mops._p(self.hyper))
        mops._call("mnk_adam_multi", self.hyper, mops._p(tab), n, blocks, mops._p(self.hyper))
        self.steps_taken += 1

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used by train.py")
        self.materialize_grads()
        active = [p for p in self._params if p.grad is not None]
        if not active:
            return None
        ws = mdist.world_size() if mdist.grads_active() else 1
        if self._exchange is not None:
            mdist.all_reduce_flat_end(self._exchange)
            self._exchange = None
        elif self._exchanged:
            pass
        elif mdist.grads_active():
            mdist.all_reduce_flat_(self.flat_grad)          # sum over ranks; the mean is folded into the update
        if not _capturing(self.device):
            self.sync_scalars(1.0 / ws)
        key = (tuple(id(p) for p in active), tuple(p.data_ptr() for p in active), mops.pack_registry_version(),
               self.reducer.direct)
        if self._table is None or self._table[0] != key:
            if _capturing(self.device):
                raise RuntimeError("the optimiser's tensor set changed during hipGraph capture (run one eager iteration first)")
            self._table = (key,) + self._build_table(active)
        _, tab, n, blocks, entries = self._table
        mops._call("mnk_adam_tick", self.hyper, mops._p(self.hyper))
        mops._call("mnk_adam_multi", self.hyper, mops._p(tab), n, blocks, mops._p(self.hyper))
        self.steps_taken += 1
        self._mnk_fresh_entries = entries          # re-stamped by the global post-step hook (mnk.ops)
        return None

    # ---- torch.optim.Adam's state_dict format (logger.py:43-66 saves / restores the three optimisers) --------------
    def _views(self, p):
        o = self._off[id(p)]
        return self.flat_m[o:o + p.numel()].view_as(p), self.flat_v[o:o + p.numel()].view_as(p)

    def state_dict(self):
        if mdist._P2P["handle"] is not None:
            mdist.check_p2p()          # never checkpoint moments that were made from NaN-poisoned statistics
        step = float(self.hyper[7].item())
        for p in self._params:
            m, v = self._views(p)
            self.state[p] = {"step": torch.tensor(step), "exp_avg": m, "exp_avg_sq": v}
        sd = super(MnkAdam, self).state_dict()
        self.state.clear()
        for st in sd["state"].values():
            st["exp_avg"] = st["exp_avg"].clone()
            st["exp_avg_sq"] = st["exp_avg_sq"].clone()
        return sd

    def load_state_dict(self, state_dict):
        super(MnkAdam, self).load_state_dict(state_dict)
        step = 0.0
        for p in self._params:
            st = self.state.get(p)
            m, v = self._views(p)
            if not st:          # no entry in the checkpoint (a parameter that never received a gradient): fresh state
                m.zero_()
                v.zero_()
                continue
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            step = max(step, float(st["step"]))
        self.state.clear()
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self.hyper.copy_(torch.tensor([g["lr"], b1, b2, g["eps"], 0.0, 0.0, self._gscale_on_device, step, 1.0 - b1,
                                       1.0 - b2], dtype=torch.float64).float())
        self._lr_on_device = float(g["lr"])


# ---- a stock torch.optim.Adam stepped by the library's kernel (round 6) ------------------------------------------------------
# The reference's train.py builds three torch.optim.Adam objects (train.py:81-83).  On the drop-in path
# (sync_batchnorm.DataParallelWithCallback -> mnk.dropin.TrainPairRunner) the gradients of a network live in ONE flat buffer
# (GradSinks), and a stock optimiser's step would walk it with ~10 multi-tensor passes (7.8 GB of traffic for the 81 M parameters
# of BASELINE configs[1]: 1.5 ms of a 10 ms iteration) and leave the GEMM layouts of the weights stale (one more pack launch that
# reads every weight again).  A global pre-step hook therefore performs the update of such an optimiser with mnk_adam_multi ON
# THE OPTIMISER'S OWN STATE TENSORS -- exp_avg / exp_avg_sq / step exactly where torch.optim.Adam keeps them, so state_dict(),
# load_state_dict() and lr schedulers see nothing unusual -- and hands the stepped parameters' gradients back after the
# (then empty) step.  Only the plain configuration is taken over (one group, no amsgrad / weight decay / maximize /
# capturable / differentiable / fused); anything else steps the stock way.  MNK_ADOPT_ADAM=0 switches it off.
_ADOPT = {"installed": False, "by_opt": weakref.WeakKeyDictionary()}


class AdoptedAdam:
    def __init__(self, opt, owner):
        self.opt = weakref.ref(opt)
        self.owner = weakref.ref(owner)
        g = opt.param_groups[0]
        b1, b2 = g["betas"]
        self.device = owner.device
        self.hyper = torch.tensor([g["lr"], b1, b2, g["eps"], 0.0, 0.0, 1.0, 0.0, 1.0 - b1, 1.0 - b2],
                                  dtype=torch.float64).float().to(self.device)
        self._host = (float(g["lr"]), float(b1), float(b2), float(g["eps"]))
        self._step = 0.0                 # the step count the device scalars hold
        self._table = None
        self._keep = []
        self._stash = None
        self.steps_taken = 0

    @staticmethod
    def eligible(opt):
        """the FlatGrads that owns exactly this optimiser's parameters, or None"""
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return None
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("weight_decay", 0) != 0 or g.get("maximize") or g.get("capturable") or \
                g.get("differentiable") or g.get("fused") or g.get("decoupled_weight_decay"):
            return None
        if not isinstance(g["lr"], float) or any(not isinstance(b, float) for b in g["betas"]):
            return None              # tensor hyper-parameters: the stock path
        ps = [p for p in g["params"] if p.requires_grad]
        owners = {id(mops.sink_owner(p)) for p in ps}
        if len(owners) != 1 or not ps:
            return None
        owner = mops.sink_owner(ps[0])
        if not isinstance(owner, GradSinks) or {id(p) for p in ps} != {id(p) for p in owner._params}:
            return None
        return owner

    def _sync(self, opt, active):
        """device scalars <- the optimiser's hyper-parameters and step count (a scheduler's new lr, a loaded checkpoint)"""
        g = opt.param_groups[0]
        host = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
        step = float(opt.state[active[0]]["step"])
        if host == self._host and step == self._step:
            return True
        if step != self._step and any(float(opt.state[p]["step"]) != step for p in active):
            return False             # parameters at different step counts: one bias correction cannot serve them
        if host[1:] != self._host[1:] or step != self._step:
            self.hyper.copy_(torch.tensor([host[0], host[1], host[2], host[3], 0.0, 0.0, 1.0, step, 1.0 - host[1], 1.0 - host[2]],
                                          dtype=torch.float64).float())
        else:
            self.hyper[0:1].copy_(torch.tensor([host[0]], dtype=torch.float32))
        self._host, self._step = host, step
        return True

    @torch.no_grad()
    def pre_step(self):
        opt, owner = self.opt(), self.owner()
        if opt is None or owner is None or _capturing(self.device):
            return False
        active = [p for p in opt.param_groups[0]["params"] if p.grad is not None]
        if not active:
            return False
        if any(p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.device != p.device for p in active):
            return False
        for p in active:                 # torch.optim.Adam._init_group's lazy state, where it keeps it
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.get_default_dtype())
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not (torch.is_tensor(st.get("step")) and st["step"].device.type == "cpu" and st["exp_avg"].is_contiguous()
                      and st["exp_avg_sq"].is_contiguous() and st["exp_avg"].dtype == torch.float32):
                return False
        if not self._sync(opt, active):
            return False
        key = (tuple((p.data_ptr(), p.grad.data_ptr(), opt.state[p]["exp_avg"].data_ptr(), opt.state[p]["exp_avg_sq"].data_ptr())
                     for p in active), mops.pack_registry_version())
        if self._table is None or self._table[0] != key:
            rows, blocks, entries = [], 0, []
            for p in active:
                st = opt.state[p]
                e = mops.pack_entry_of(p)
                if e is not None:
                    cout, c0, c1, up = e.meta
                    nb = _lib.lib().query("mnk_adam_blocks", 0, cout, c0, c1, 1)
                    rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                 e.wp.data_ptr(), e.wd[0].data_ptr() if e.wd[0] is not None else 0,
                                 e.wd[1].data_ptr() if e.wd[1] is not None else 0, cout, c0, c1, blocks, int(up), 0, 0, 0, 0, 0))
                    entries.append(e)
                else:
                    nb = _lib.lib().query("mnk_adam_blocks", p.numel(), 0, 0, 0, 0)
                    rows.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                 0, 0, 0, 0, 0, 0, blocks, 0, 0, 0, 0, 0, 0))
                blocks += nb
            self._keep = []              # (nothing captures these tables: the previous one may go)
            tab = _device_table(np.array(rows, dtype=ADAM_DESC), self.device, self._keep)
            self._table = (key, tab, len(rows), blocks, tuple(entries))
        _, tab, n, blocks, entries = self._table
        mops._call("mnk_adam_tick", self.hyper, mops._p(self.hyper))
        mops._call("mnk_adam_multi", self.hyper, mops._p(tab), n, blocks, mops._p(self.hyper))
        torch._foreach_add_([opt.state[p]["step"] for p in active], 1.0)
        self._step += 1.0
        self.steps_taken += 1
        opt._mnk_fresh_entries = entries         # stamped fresh by the global post-step hook of mnk.ops
        # the stock step that follows must find nothing to do; post_step() hands the gradients back
        self._stash = [(p, p.grad) for p in active]
        for p in active:
            p.grad = None
        return True

    def post_step(self):
        if self._stash is not None:
            for p, g in self._stash:
                if p.grad is None:
                    p.grad = g
            self._stash = None


def install_adam_adoption():
    if _ADOPT["installed"]:
        return False
    from torch.optim.optimizer import register_optimizer_step_pre_hook, register_optimizer_step_post_hook

    def pre_step(opt, args, kwargs):
        # (args = (optimiser, closure?): a step with a closure is the stock step's business)
        if type(opt) is not torch.optim.Adam or not mops.knobs.on("MNK_ADOPT_ADAM") or (len(args) > 1 and args[1] is not None) \
                or kwargs.get("closure") is not None:
            return
        ad = _ADOPT["by_opt"].get(opt)
        # (a refusal is remembered together with the size of the sink registry: a runner created later asks again)
        if ad is None or (isinstance(ad, int) and ad != len(mops._SINKS)) or (isinstance(ad, AdoptedAdam) and ad.owner() is None):
            owner = AdoptedAdam.eligible(opt)
            ad = AdoptedAdam(opt, owner) if owner is not None else len(mops._SINKS)
            _ADOPT["by_opt"][opt] = ad
        if isinstance(ad, AdoptedAdam):
            ad.pre_step()

    def post_step(opt, args, kwargs):
        ad = _ADOPT["by_opt"].get(opt)
        if isinstance(ad, AdoptedAdam):
            ad.post_step()

    register_optimizer_step_pre_hook(pre_step)
    register_optimizer_step_post_hook(post_step)
    _ADOPT["installed"] = True
    return True


def adopted(opt):
    """the AdoptedAdam of a stock optimiser (tests, bench.py), or None"""
    ad = _ADOPT["by_opt"].get(opt)
    return ad if isinstance(ad, AdoptedAdam) else None
