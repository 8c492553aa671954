"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in the
CPU tests).  Replaces the reference's single-process DataParallel + SyncMaster threads
(sync_batchnorm/replicate.py:50-67, comm.py, batchnorm.py:90-111):

* BatchNorm sufficient statistics  [sum x, sum x^2]  (forward) and  [sum g, sum g*xhat]  (backward) are summed over
  ranks with one small all-reduce per layer (`all_reduce_sum_`); every rank then finalises mean / inv-std itself,
  so the reference's reduce-to-master + broadcast pair (batchnorm.py:102,105) collapses into one collective.
  Ranks are assumed to hold equally sized shards (count = local count * world size).
* Gradients are averaged with a bucketed flat all-reduce after backward (`GradAverager`), replacing the implicit
  reduce-add of DataParallel's backward and its per-forward parameter broadcast.
"""

import torch
import torch.distributed as tdist

from . import knobs

_SYNC_BN = True


def set_sync_bn(flag):
    """Disable to get per-rank BatchNorm statistics (plain DDP semantics)."""
    global _SYNC_BN
    _SYNC_BN = bool(flag)


def initialized():
    return tdist.is_available() and tdist.is_initialized()


def world_size():
    return tdist.get_world_size() if initialized() else 1


def rank():
    return tdist.get_rank() if initialized() else 0


_FORCE = knobs.on("MNK_DIST_FORCE")   # exercise the collectives even with a single rank (tests)


def active():
    """True when BatchNorm statistics must be exchanged."""
    return _SYNC_BN and initialized() and (tdist.get_world_size() > 1 or _FORCE)


# ---- the library's own RCCL communicator (include/monkeynet_hip.h: mnk_comm_*, mnk_allreduce_*) ------------------------
# With the nccl (= RCCL) backend the two collectives of the hot path are issued by libmonkeynet_hip.so on the stream the
# kernels run on: in order with producer and consumer, no event hand-over to torch.distributed's communication stream
# (two event waits per collective, 84 small SyncBN collectives per iteration), capturable as they are.  torch.distributed
# stays the rendezvous (it carries the 128-byte id) and the transport of everything else (gloo in the CPU tests).
_COMMS = {}


def direct_comm(kind="main"):
    """The communicator handle (an int) or None: gloo / CPU builds / RCCL not loadable / MNK_RCCL_DIRECT=0.
    kind "main": the communicator of the kernels' stream (SyncBN sums, in-order gradient exchange); "grads": a second
    communicator for the gradient exchange that runs on its own stream next to the discriminator's backward pass -- one
    communicator must not have operations in flight on two streams."""
    _COMM = _COMMS.setdefault(kind, {"tried": False, "handle": None})
    if _COMM["tried"]:
        return _COMM["handle"]
    _COMM["tried"] = True
    try:
        if not (knobs.on("MNK_RCCL_DIRECT") and initialized() and tdist.get_backend() == "nccl"):
            return None
        import ctypes
        from . import _lib
        lib = _lib.lib()
        if not lib.is_device_build or not lib.cdll.mnk_comm_available():
            return None
        dev = torch.device("cuda", torch.cuda.current_device())
        ident = torch.zeros(128, dtype=torch.uint8)
        if tdist.get_rank() == 0:
            lib.call("mnk_comm_unique_id", ident.data_ptr())
        on_dev = ident.to(dev)
        tdist.broadcast(on_dev, 0)
        ident = on_dev.cpu()
        handle = ctypes.c_void_p()
        lib.call("mnk_comm_init", ident.data_ptr(), tdist.get_rank(), tdist.get_world_size(), ctypes.byref(handle))
        _COMM["handle"] = handle.value
    except Exception as e:      # the torch.distributed path below is always there
        import sys
        sys.stderr.write("mnk.dist: direct RCCL communicator not available (%s: %s); using torch.distributed\n"
                         % (type(e).__name__, e))
        _COMM["handle"] = None
    return _COMM["handle"]


# ---- the SyncBN exchange of one node as the library's own peer-to-peer kernel (csrc/p2p.hip, mnk_p2p_*) -----------------
# One 256-thread kernel per norm layer and direction instead of one RCCL all-reduce: every rank pushes its <= 8 KB vector into
# every rank's IPC-mapped mailbox and adds the rows in rank order (bit-identical sums on all ranks).  torch.distributed only
# carries the 64-byte IPC handles once.  Used when every rank of the group runs on this host (one process per GPU of one node:
# the deployment BASELINE.json names); otherwise -- or with MNK_SYNCBN_P2P=0 -- the RCCL path below.
_P2P = {"tried": False, "handle": None, "max": 0}
# How long a rank's kernel polls for a peer's word (wall clock).  Long: a rank that is merely late -- a checkpoint or the
# visualiser on rank 0, the first iteration's capture -- must not be given up on; bounded: a dead peer must not hang the
# GPU for good.  A give-up is never silent: the exchanged sums become NaN (csrc/p2p.h: P2P_POISON), every later exchange
# gives up at once (the error word is set), and TrainStep / check_p2p() raise.
P2P_TIMEOUT_MS = int(knobs.get("MNK_P2P_TIMEOUT_MS"))


def p2p_comm(force=False):
    """The peer-to-peer exchange handle (an int) or None.  force: also with a gloo group (the GPU test runs several processes
    on ONE device, which RCCL does not allow)."""
    if _P2P["tried"]:
        return _P2P["handle"]
    _P2P["tried"] = True
    try:
        if not (knobs.on("MNK_SYNCBN_P2P") and initialized() and (force or tdist.get_backend() == "nccl")):
            return None
        import ctypes
        import socket
        from . import _lib
        lib = _lib.lib()
        if not lib.is_device_build or not torch.cuda.is_available():
            return None
        world, me = tdist.get_world_size(), tdist.get_rank()
        handle = ctypes.c_void_p()
        mine = (ctypes.c_ubyte * 64)()
        ok = True
        try:
            lib.call("mnk_p2p_create", me, world, ctypes.byref(handle))
            lib.call("mnk_p2p_export", handle, mine)
        except Exception as e:
            ok = False
            why = str(e)
        # every rank takes part in the gather whatever happened locally: the decision must be the same on all ranks
        gathered = [None] * world
        tdist.all_gather_object(gathered, (socket.gethostname(), bytes(mine) if ok else None))
        if not all(g[1] is not None for g in gathered) or len({g[0] for g in gathered}) != 1:
            if handle:
                lib.cdll.mnk_p2p_destroy(handle)
            return None
        blob = b"".join(g[1] for g in gathered)
        ok2 = True
        try:
            lib.call("mnk_p2p_connect", handle, ctypes.c_char_p(blob))
        except Exception as e:
            ok2 = False
            why = str(e)
        flags = [None] * world
        tdist.all_gather_object(flags, ok2)
        if not all(flags):
            lib.cdll.mnk_p2p_destroy(handle)
            if not ok2:
                raise RuntimeError(why)
            return None
        # self-test before the exchange carries any statistics: one all-reduce of rank-dependent values over the mapped
        # mailboxes, bounded by a short timeout; every rank must see the right sums and no give-up flag, or ALL ranks stay on
        # the collective path (the decision is gathered, so it is the same everywhere)
        dev = torch.device("cuda", torch.cuda.current_device())
        probe = torch.arange(1, 65, dtype=torch.float32, device=dev) * float(me + 1)
        got = torch.empty_like(probe)
        fine = True
        try:
            lib.call("mnk_p2p_allreduce", handle, probe.data_ptr(), got.data_ptr(), 64, 3000,
                     torch.cuda.current_stream(dev).cuda_stream)
            torch.cuda.synchronize(dev)
            flag = ctypes.c_int(0)
            lib.call("mnk_p2p_error", handle, ctypes.byref(flag))
            want = torch.arange(1, 65, dtype=torch.float32) * float(world * (world + 1) // 2)
            fine = flag.value == 0 and torch.equal(got.cpu(), want)
        except Exception:
            fine = False
        verdicts = [None] * world
        tdist.all_gather_object(verdicts, bool(fine))
        if not all(verdicts):
            import sys
            sys.stderr.write("mnk.dist: the peer-to-peer exchange failed its self-test on rank(s) %s; using the collective path\n"
                             % [i for i, v in enumerate(verdicts) if not v])
            return None               # (the mailboxes stay mapped: a rank may still be inside the probe kernel of a slow peer)
        _P2P["handle"] = handle.value
        _P2P["max"] = int(lib.query("mnk_p2p_max_floats"))
    except Exception as e:      # the RCCL / torch.distributed path below is always there
        import sys
        sys.stderr.write("mnk.dist: peer-to-peer SyncBN exchange not available (%s: %s); using the collective path\n"
                         % (type(e).__name__, e))
        _P2P["handle"] = None
    return _P2P["handle"]


def p2p_error():
    """0, or 1 + the rank whose contribution an exchange gave up waiting for (synchronises the device)"""
    if _P2P["handle"] is None:
        return 0
    import ctypes
    from . import _lib
    flag = ctypes.c_int(0)
    _lib.lib().call("mnk_p2p_error", ctypes.c_void_p(_P2P["handle"]), ctypes.byref(flag))
    return int(flag.value)


def disable_p2p():
    """Leave the peer-to-peer exchange for the rest of the process (every rank must call it at the same point): the SyncBN
    sums travel through the collective path from now on.  The mailboxes stay mapped -- a peer may still be inside a kernel."""
    _P2P["tried"] = True
    _P2P["handle"] = None
    _P2P["max"] = 0


def check_p2p():
    """Raise if a peer-to-peer exchange gave a rank up (synchronises the device; TrainStep calls it every few iterations, a
    user-owned loop may call it before it writes a checkpoint)."""
    code = p2p_error()
    if code:
        raise RuntimeError("SyncBN peer-to-peer exchange: rank %d did not deliver its statistics within %d ms; the sums of "
                           "this rank were poisoned with NaN (MNK_P2P_TIMEOUT_MS, MNK_SYNCBN_P2P=0 for the RCCL path)"
                           % (code - 1, P2P_TIMEOUT_MS))


def _p2p_sum(t, out):
    h = p2p_comm() if t.is_cuda else None
    if h is None or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() > _P2P["max"]:
        return False
    from . import _lib
    import ctypes
    _lib.lib().call("mnk_p2p_allreduce", ctypes.c_void_p(h), t.data_ptr(), out.data_ptr(), t.numel(), P2P_TIMEOUT_MS, _stream_of(t))
    return True


def _stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def all_reduce_sum_(t):
    """SyncBN exchange: sum a small fp32 vector over the ranks in place."""
    if _p2p_sum(t, t):
        return t
    h = direct_comm() if t.is_cuda else None
    if h is not None and t.dtype == torch.float32 and t.is_contiguous():
        from . import _lib
        _lib.lib().call("mnk_allreduce_bnstats", h, t.data_ptr(), t.numel(), _stream_of(t))
        return t
    tdist.all_reduce(t, op=tdist.ReduceOp.SUM)
    return t


def all_reduce_sum(t):
    """-> a new tensor: the sum of `t` over the ranks; `t` keeps the local values (no copy launch in front of the collective)."""
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
        out = torch.empty_like(t)
        if _p2p_sum(t, out):
            return out
    h = direct_comm() if t.is_cuda else None
    if h is not None and t.dtype == torch.float32 and t.is_contiguous():
        from . import _lib
        out = torch.empty_like(t)
        _lib.lib().call("mnk_allreduce_bnstats_to", h, t.data_ptr(), out.data_ptr(), t.numel(), _stream_of(t))
        return out
    return all_reduce_sum_(t.clone())


def grads_active():
    """True when gradients must be exchanged (several ranks, or the forced single-rank exercise)."""
    return initialized() and (tdist.get_world_size() > 1 or _FORCE)


def all_reduce_flat_(flat, chunk_mb=64.0):
    """Sum a flat fp32 buffer over the ranks in place, as a few large collectives (xGMI rings are per-link bound: fewer,
    larger messages) launched back to back; the caller's stream waits for them, the host does not."""
    step = max(int(chunk_mb * 1024 * 1024 / 4), 1)
    h = direct_comm() if flat.is_cuda else None
    if h is not None and flat.dtype == torch.float32 and flat.is_contiguous():
        from . import _lib
        _lib.lib().call("mnk_allreduce_grads", h, flat.data_ptr(), flat.numel(), 0, step, _stream_of(flat))
        return flat
    works = [tdist.all_reduce(flat[i:i + step], op=tdist.ReduceOp.SUM, async_op=True) for i in range(0, flat.numel(), step)]
    for w in works:
        w.wait()
    return flat


_COMM_STREAMS = {}


def all_reduce_flat_begin(flat, chunk_mb=64.0):
    """Start the sum of a flat fp32 buffer over the ranks WITHOUT making the kernels' stream wait: the exchange of the
    generator's (and key-point detector's) 265-379 MB of gradients runs next to the discriminator-loss backward pass, which
    depends on neither (what DataParallel's backward hides inside itself, train.py:116-118).  -> handle for
    all_reduce_flat_end.  RCCL: on a communication stream of its own, with a communicator of its own; torch.distributed
    (gloo in the CPU tests): asynchronous works."""
    step = max(int(chunk_mb * 1024 * 1024 / 4), 1)
    h = direct_comm("grads") if flat.is_cuda else None
    if h is not None and flat.dtype == torch.float32 and flat.is_contiguous():
        from . import _lib
        side = _COMM_STREAMS.get(flat.device)
        if side is None:
            side = _COMM_STREAMS[flat.device] = torch.cuda.Stream(flat.device)
        side.wait_stream(torch.cuda.current_stream(flat.device))         # the gradients are complete
        _lib.lib().call("mnk_allreduce_grads", h, flat.data_ptr(), flat.numel(), 0, step, side.cuda_stream)
        return ("stream", side, flat.device)
    return ("works", [tdist.all_reduce(flat[i:i + step], op=tdist.ReduceOp.SUM, async_op=True)
                      for i in range(0, flat.numel(), step)])


def all_reduce_flat_end(handle):
    """the consumer (the optimiser step) waits for an exchange started by all_reduce_flat_begin"""
    if handle[0] == "stream":
        torch.cuda.current_stream(handle[2]).wait_stream(handle[1])
    else:
        for w in handle[1]:
            w.wait()


def average_grads_(params, bucket_mb=64.0):
    """Generic (user-owned loop) gradient averaging: flat buckets, one all-reduce each, p.grad re-pointed at its slice of
    the averaged bucket (no copy back)."""
    ws = float(world_size())
    limit = int(bucket_mb * 1024 * 1024 / 4)
    cur, size, buckets = [], 0, []
    for p in params:
        if p.grad is None:
            continue
        cur.append(p)
        size += p.grad.numel()
        if size >= limit:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    launched = []
    for ps in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        launched.append((tdist.all_reduce(flat, op=tdist.ReduceOp.SUM, async_op=True), flat, ps))
    for work, flat, ps in launched:
        work.wait()
        flat.div_(ws)
        off = 0
        for p in ps:
            k = p.grad.numel()
            p.grad = flat[off:off + k].view_as(p.grad)
            off += k
    return sum(len(ps) for ps in buckets)


_AUTO = {"installed": False}


def install_grad_averaging():
    """A torch.optim pre-step hook that averages the stepped optimiser's gradients over the ranks: the reference's
    unmodified train.py (`loss.backward(); optimizer.step()`, train.py:117-118,131-132) then trains data-parallel under
    torch.distributed.run without an edit -- what DataParallel's backward did implicitly (train.py:104-105).  Installed by
    sync_batchnorm.DataParallelWithCallback when a process group exists.  mnk.optim.MnkAdam exchanges its own flat buffer
    and is skipped."""
    if _AUTO["installed"]:
        return False
    from torch.optim.optimizer import register_optimizer_step_pre_hook

    def pre_step(opt, args, kwargs):
        if not grads_active() or getattr(opt, "_mnk_owns_exchange", False):
            return
        average_grads_([p for g in opt.param_groups for p in g["params"]])

    register_optimizer_step_pre_hook(pre_step)
    _AUTO["installed"] = True
    return True


def check_equal_shards(n):
    """SyncBN finalises with count = local count * world size and gradients are averaged with 1 / world size: both assume that
    every rank holds the same number of samples.  Raises on the ranks' first disagreement (one tiny collective; callers cache
    the checked value)."""
    if not initialized() or tdist.get_world_size() == 1:
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if tdist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(n), -float(n)], dtype=torch.float32, device=dev)
    tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    hi, lo = int(t[0].item()), -int(t[1].item())
    if hi != lo:
        raise ValueError("ranks hold different numbers of samples (%d ... %d; this rank %d): the SyncBN counts and the gradient "
                         "average of the data-parallel step assume equal shards" % (lo, hi, n))


def combine_bn_stats(sums, count):
    """Host-side statement of the SyncBN exchange for tests: returns (global sums, global count)."""
    if active():
        sums = sums.clone()
        all_reduce_sum_(sums)
        count = count * world_size()
    return sums, count


class GradAverager:
    """Bucketed gradient averaging over ranks, overlapped with backward.

    Parameters are assigned (in reverse registration order ~ the order backward produces them) to flat fp32 buckets of
    ~`bucket_mb` MB -- fewer, larger collectives: xGMI rings are per-link bound.  `arm()` before backward; a
    post-accumulate-grad hook launches a bucket's asynchronous all-reduce as soon as its last gradient has been
    written, so the exchange of early buckets runs under the rest of backward; `average()` after backward launches
    whatever is left (parameters that received no gradient are skipped), waits, divides by the world size and points
    every `p.grad` at its slice of the averaged bucket (no copy back).  Launch order is a pure function of the autograd graph, hence identical on all ranks.
    Replaces DataParallel's implicit reduce-add + per-forward parameter broadcast (train.py:104-105)."""

    def __init__(self, params, bucket_mb=64.0, overlap=None):
        self.params = [p for p in params if p.requires_grad]
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel()
            if size >= limit:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._pending = [0] * len(self.buckets)
        self._launched = [None] * len(self.buckets)
        self._armed = False
        if overlap is None:
            overlap = knobs.on("MNK_GRAD_OVERLAP")
        self.overlap = overlap
        self._hooks = []
        if overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _enabled(self):
        return initialized() and (world_size() > 1 or _FORCE)

    def arm(self):
        """Call right before backward()."""
        self._armed = self._enabled() and self.overlap
        self._pending = [len(b) for b in self.buckets]
        self._launched = [None] * len(self.buckets)

    def _launch(self, i):
        ps = [p for p in self.buckets[i] if p.grad is not None]
        if not ps:
            self._launched[i] = ()
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        work = tdist.all_reduce(flat, op=tdist.ReduceOp.SUM, async_op=True)
        self._launched[i] = (work, flat, ps)

    def _on_grad(self, p):
        if not self._armed:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0 and self._launched[i] is None:
            self._launch(i)

    def average(self):
        """Call after backward(), before optimizer.step().  Returns the number of averaged parameters."""
        self._armed = False
        if not self._enabled():
            return 0
        ws = float(world_size())
        n = 0
        for i in range(len(self.buckets)):
            if self._launched[i] is None:
                self._launch(i)
        for item in self._launched:
            if not item:
                continue
            work, flat, ps = item
            work.wait()
            flat.div_(ws)
            off = 0
            for p in ps:
                k = p.grad.numel()
                p.grad = flat[off:off + k].view_as(p.grad)       # no copy back: the gradient now lives in the bucket
                off += k
            n += len(ps)
        self._launched = [None] * len(self.buckets)
        return n


def shard_batch(x, dim=0):
    """This rank's equal slice of a global batch tensor (the DataParallel scatter, replicate.py:64-67)."""
    ws, r = world_size(), rank()
    if ws == 1:
        return x
    n = x.shape[dim]
    assert n % ws == 0, "global batch %d is not divisible by world size %d" % (n, ws)
    per = n // ws
    return x.narrow(dim, r * per, per)
