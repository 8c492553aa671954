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


class MnkAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) semantics (no amsgrad / weight decay), one kernel launch per step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super(MnkAdam, self).__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) != 1:
            raise ValueError("MnkAdam takes one parameter group (train.py:81-83 builds one optimiser per network)")
        ps = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        if not ps:
            raise ValueError("no trainable parameters")
        dev = ps[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in ps):
            raise ValueError("MnkAdam needs fp32 parameters on one device")
        mops._check_device(ps[0])
        self._params = ps
        self.device = dev
        off, self._off = 0, {}
        for p in ps:
            self._off[id(p)] = off
            off += (p.numel() + 3) // 4 * 4                  # 16-byte aligned slices: float4 path of the kernel
        self.numel = off
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=dev)
        self._sinks = {id(p): self.flat_grad[self._off[id(p)]:self._off[id(p)] + p.numel()].view_as(p) for p in ps}
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        self.hyper = torch.tensor([g["lr"], b1, b2, g["eps"], 0.0, 0.0, 1.0, 0.0, 1.0 - b1, 1.0 - b2],
                                  dtype=torch.float64).float().to(dev)
        self._lr_on_device = float(g["lr"])
        self._gscale_on_device = 1.0
        self.steps_taken = 0
        self.reducer = DeferredReducer(self)
        self._written = set()            # ids of parameters whose sink was written by a kernel since zero_grad
        self._mnk_owns_exchange = True   # mnk.dist's generic pre-step gradient averaging skips this optimiser
        self._table = None               # (key, device table, n, blocks, entries)
        self._keep = []
        self._mnk_fresh_entries = ()
        self._exchange = None            # a gradient exchange started by begin_exchange(), finished by step()
        # True: the step kernel takes the gradients of the few-split tap-major layers (the deep levels: 330 of the generator's
        # 690 MB of partials ARE the gradient) straight from the partials, and mnk_wgrad_reduce_multi skips them -- `p.grad` of
        # those parameters is then NOT valid.  mnk.engine.TrainStep switches it on for a captured iteration of one process
        # (nobody can look at p.grad between the launches of a replay; several ranks exchange the flat buffer)
        self.tap_direct = False
        self._exchanged = False          # ... or already finished by exchange_end(): step() must not sum again
        for p in ps:
            mops.register_grad_sink(p, self)
        weakref.finalize(self, mops.unregister_grad_sinks, [id(p) for p in ps], id(self))

    # ---- gradient sinks ----------------------------------------------------------------------------------------------
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
        """After backward: every gradient of this optimiser's parameters in its slice of the flat buffer, `p.grad`
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
