"""The reference's OWN training loop (train.py:104-136) served from captured hipGraphs -- without the user adopting
mnk.engine.TrainStep.

train.py wraps its two "full models" in DataParallelWithCallback (train.py:104-105) and then runs, per batch,

    out = generator_full_par(x); loss = sum(v.mean() for v in out[:-2]); loss.backward(...)        # train.py:110-117
    optimizer_generator.step() ... optimizer_kp_detector.step() ...                                # :118-123
    loss_values = discriminator_full_par(x, kp_joined, generated); ...; loss.backward()            # :127-131
    optimizer_discriminator.step() ...                                                             # :132-136

Every statement of that loop is Python that launches a few hundred small kernels: on these modules the loop is bound by the
host (14.3 ms per iteration where the captured iteration of TrainStep takes 10.3).  sync_batchnorm.DataParallelWithCallback
recognises the two wrapped modules (attributes kp_extractor / generator / discriminator / train_params) and hands their calls
to ONE TrainPairRunner per network triple, which splits the iteration into three phases, each captured once as a hipGraph
(one memory pool, the make_graphed_callables pattern with the library's own capture):

  A  forward: key-point detector on [source | driving], generator, ONE discriminator pass on [generated; real] that serves the
     generator loss AND the discriminator loss (the reference's second discriminator forward recomputes the first one's values:
     same weights, same inputs -- mnk.engine.TrainStep._eager_step_shared), the per-sample loss vectors;
  B  `loss.backward()` of the generator pass: a whole-model autograd Function hands the loss vectors out; its backward copies the
     incoming gradients into static buffers and replays: discriminator (no weight gradients: the reference throws them away,
     train.py:120) -> generator + key-point detector, weight-gradient GEMMs grouped per tile shape, one reduction launch into a
     flat gradient buffer per network (mnk.optim.GradSinks); p.grad of every parameter is a view of it;
  C  `loss.backward()` of the discriminator pass through the retained discriminator graph (the generated frames are leaves:
     the reference's `.detach()`), discriminator weight gradients into their flat buffer.

The optimiser steps stay the caller's statements.  Stock torch.optim.Adam objects over these networks are stepped by
mnk_adam_multi on their own state tensors (mnk.optim.AdoptedAdam: one launch that also leaves the packed GEMM layouts);
mnk.optim.MnkAdam objects own the flat buffers themselves.

Anything unexpected -- evaluation mode, no_grad, gradients that were not zeroed, a discriminator call on tensors that are not
this iteration's outputs, changed discriminator weights in between, several driving frames, a process group -- falls back
to calling the wrapped module as it is.  MNK_DROPIN_GRAPH=0 switches the runner off, =phases runs the three phases as eager
launches (what the CPU-emulator tests exercise; also the form used when the device has no graph support).

Returned tensors of the graph form (loss vectors, `generated`, `kp_joined`) are static buffers that the next
generator_full_par(x) call overwrites -- the loop's `.cpu()` copies and logger calls read them before that.
"""
import gc
import weakref

import torch

from . import dist as mdist
from . import knobs
from . import ops as mops


def _detached(obj):
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, dict):
        return {k: _detached(v) for k, v in obj.items()}
    return obj


class _PhaseFn(torch.autograd.Function):
    """The loss vectors of phase A as outputs of ONE autograd node whose backward runs phase B ("g") or C ("d")."""

    @staticmethod
    def forward(ctx, anchor, runner, which, epoch):
        ctx.runner, ctx.which, ctx.epoch = runner, which, epoch
        ctx.set_materialize_grads(False)
        return tuple(v.detach() for v in runner._vectors(which))

    @staticmethod
    def backward(ctx, *grads):
        ctx.runner._backward(ctx.which, ctx.epoch, grads)
        return None, None, None, None


class _Program:
    """one captured iteration: three hipGraphs over one memory pool and their static tensors"""
    __slots__ = ("x", "gA", "gB", "gC", "g_vec", "d_vec", "g_grads", "d_grads", "generated", "kp_joined", "grads_g", "grads_d",
                 "reg_version", "keep")


class TrainPairRunner:
    def __init__(self, kp_extractor, generator, discriminator, train_params):
        # weak references: the runner lives in a registry keyed by the networks and must not keep them alive (the wrappers that call
        # it do); it dies with the generator (runner_for)
        self._kp, self._gen, self._disc = weakref.ref(kp_extractor), weakref.ref(generator), weakref.ref(discriminator)
        self.tp = train_params
        self.device = next(generator.parameters()).device
        mode = knobs.get("MNK_DROPIN_GRAPH")
        self.use_graph = mode == "1" and self.device.type == "cuda"
        self.g_params = [p for p in generator.parameters() if p.requires_grad]
        self.k_params = [p for p in kp_extractor.parameters() if p.requires_grad]
        self.d_params = [p for p in discriminator.parameters() if p.requires_grad]
        self.owners = None                # {"g": FlatGrads, "k": ..., "d": ...}, made at the first call
        self.programs = {}                # input shapes -> _Program
        self.epoch = 0
        self.cur = None                   # this iteration: phase state (eager phases) or the _Program (graph form)
        self._done = {"g": True, "d": True}
        self._last = None
        self._anchor = torch.zeros((), dtype=torch.float32, device=self.device, requires_grad=True)
        self.stats = {"graph_calls": 0, "phase_calls": 0, "captures": 0, "fallbacks": 0, "d_fallbacks": 0}

    kp = property(lambda self: self._kp())
    gen = property(lambda self: self._gen())
    disc = property(lambda self: self._disc())

    # ---- eligibility ---------------------------------------------------------------------------------------------------------
    def _owners(self):
        if self.owners is not None:
            return self.owners
        from . import optim as moptim
        out = {}
        for name, ps in (("g", self.g_params), ("k", self.k_params), ("d", self.d_params)):
            have = {id(mops.sink_owner(p)): mops.sink_owner(p) for p in ps}
            if len(have) == 1 and None not in have.values():
                owner = next(iter(have.values()))           # the caller's mnk.optim.MnkAdam
                if {id(p) for p in owner._params} != {id(p) for p in ps}:
                    return None
            elif set(have.values()) == {None}:
                owner = moptim.GradSinks(ps)
            else:
                return None
            out[name] = owner
        moptim.install_adam_adoption()
        self.owners = out
        return out

    def _ready(self, x):
        if not torch.is_grad_enabled() or mdist.initialized():
            return False
        if not (self.kp.training and self.gen.training and self.disc.training):
            return False
        if not (isinstance(x, dict) and torch.is_tensor(x.get("source")) and torch.is_tensor(x.get("video"))):
            return False
        s, v = x["source"], x["video"]
        if s.dim() != 5 or v.dim() != 5 or s.shape[2] != 1 or v.shape[2] != 1 or s.shape != v.shape:
            return False
        if s.dtype != torch.float32 or v.dtype != torch.float32:
            return False
        if not (knobs.form("DISC_SHARED") and knobs.form("FUSED_FM_LOSS") and hasattr(self.disc, "forward_acts")):
            return False
        if self._owners() is None:
            return False
        # gradients that were not zeroed would have to be accumulated into: the sinks are written, not added to
        return all(p.grad is None for p in self.g_params) and all(p.grad is None for p in self.k_params) and \
            all(p.grad is None for p in self.d_params)

    # ---- the three phases (shared by the eager-phase form and the capture) --------------------------------------------------
    def _phase_a(self, x):
        from .engine import joined_kp, split_kp, fused_pair_losses
        tp = self.tp
        for o in self.owners.values():
            o.begin_pass()
        mops.clear_dz_stats()
        kp_joined = joined_kp(self.kp, x)
        generated = self.gen(x["source"], **split_kp(kp_joined, tp["detach_kp_generator"]))
        fake = generated["video_prediction"]
        fake_leaf = fake.detach().requires_grad_(True)
        names = list(kp_joined.keys())
        kp_leaf = {k: kp_joined[k].detach().requires_grad_(True) for k in names}
        g_vec, d_vec = fused_pair_losses(self.disc, fake_leaf, x["video"], split_kp(kp_leaf, False), generated["video_deformed"],
                                         tp["loss_weights"])
        generated.update(split_kp(kp_joined, False))
        return {"kp_joined": kp_joined, "generated": generated, "fake": fake, "fake_leaf": fake_leaf, "names": names,
                "kp_leaf": kp_leaf, "g_vec": list(g_vec), "d_vec": list(d_vec)}

    def _phase_b(self, st, grads):
        """dL_G / d(parameters of generator and key-point detector) for the gradients `grads` of the generator-loss vectors"""
        tp = self.tp
        names = st["names"]
        leaves = [st["fake_leaf"]] + [st["kp_leaf"][k] for k in names]
        with mops.no_param_grads():          # through the discriminator only (its own weight gradients are thrown away, train.py:120)
            seeds = torch.autograd.grad(st["g_vec"], leaves, grad_outputs=list(grads), retain_graph=True, allow_unused=True)
        roots, root_grads = [], []
        for t, g in zip([st["fake"]] + [st["kp_joined"][k] for k in names], seeds):
            if g is not None and t.requires_grad:
                roots.append(t)
                root_grads.append(g)
        if tp["loss_weights"]["reconstruction_deformed"] != 0:       # the only term of L_G that does not pass the cut: vector 0
            roots.append(st["g_vec"][0])
            root_grads.append(grads[0])
        torch.autograd.backward(roots, root_grads, inputs=self.g_params + self.k_params,
                                retain_graph=not tp["detach_kp_discriminator"])
        self.owners["g"].materialize_grads()
        self.owners["k"].materialize_grads()

    def _phase_c(self, st, grads):
        """dL_D / d(discriminator parameters) (and, unless detach_kp_discriminator, the key-point detector's share) through the
        discriminator graph phase B retained"""
        tp = self.tp
        if tp["detach_kp_discriminator"]:
            with mops.no_leaf_input_grads():
                torch.autograd.backward(st["d_vec"], list(grads), inputs=self.d_params)
        else:
            kl = [st["kp_leaf"][k] for k in st["names"]]
            for t in kl:
                t.grad = None
            torch.autograd.backward(st["d_vec"], list(grads), inputs=self.d_params + kl)
            back = [(st["kp_joined"][k], st["kp_leaf"][k].grad) for k in st["names"] if st["kp_leaf"][k].grad is not None]
            if back:
                # (the key-point detector has not stepped yet, train.py:133-135: a second contribution to its gradients -- the
                # sinks' slow path adds it, FlatGrads.add_to_sink)
                torch.autograd.backward([t for t, _ in back], [g for _, g in back], inputs=self.k_params)
                self.owners["k"].materialize_grads()
        self.owners["d"].materialize_grads()

    # ---- capture -----------------------------------------------------------------------------------------------------------------
    def _clear_grads(self):
        for ps in (self.g_params, self.k_params, self.d_params):
            for p in ps:
                p.grad = None
        for o in self.owners.values():
            o.begin_pass()

    def _capture(self, x):
        b = int(x["source"].shape[0])
        dev = self.device
        prog = _Program()
        prog.keep = []
        prog.x = {k: x[k].to(dev).clone() for k in ("source", "video")}
        mods = (self.kp, self.gen, self.disc)
        snap = [(t, t.detach().clone()) for m in mods for t in m.buffers()]       # BatchNorm running statistics
        n_g = None
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for it in range(2):          # sizes scratch buffers, creates the packed weights and every descriptor table
                st = self._phase_a(prog.x)
                n_g, n_d = len(st["g_vec"]), len(st["d_vec"])
                ones_g = [torch.full((b,), 1.0 / b, device=dev) for _ in range(n_g)]
                ones_d = [torch.full((b,), 1.0 / b, device=dev) for _ in range(n_d)]
                self._phase_b(st, ones_g)
                self._phase_c(st, ones_d)
                del st
                self._clear_grads()
            prog.g_grads = [torch.full((b,), 1.0 / b, device=dev) for _ in range(n_g)]
            prog.d_grads = [torch.full((b,), 1.0 / b, device=dev) for _ in range(n_d)]
            mops.repack_registered()     # every packed layout fresh: no pack launch is recorded into phase A
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        gc_was_on = gc.isenabled()
        gc.disable()
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        graphs = []
        current = [None]

        def captured(fn):
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool)
            current[0] = g
            fn()
            g.capture_end()
            current[0] = None
            graphs.append(g)
            return g

        state = {}
        try:
            with torch.cuda.stream(cap):
                prog.gA = captured(lambda: state.update(self._phase_a(prog.x)))
                prog.gB = captured(lambda: self._phase_b(state, prog.g_grads))
                prog.grads_g = [(p, self.owners[n].sink(p)) for n, ps in (("g", self.g_params), ("k", self.k_params))
                                for p in ps if p.grad is not None]
                for p, _ in prog.grads_g:
                    p.grad = None
                prog.gC = captured(lambda: self._phase_c(state, prog.d_grads))
                own = (("d", self.d_params),) if self.tp["detach_kp_discriminator"] else (("d", self.d_params), ("k", self.k_params))
                prog.grads_d = [(p, self.owners[n].sink(p)) for n, ps in own for p in ps if p.grad is not None]
        except BaseException:
            if current[0] is not None:
                try:
                    current[0].capture_end()
                except Exception:
                    _BROKEN.append(current[0])        # (its destructor would abort the process: keep it)
            raise
        finally:
            if gc_was_on:
                gc.enable()
        torch.cuda.current_stream().wait_stream(cap)
        prog.g_vec = [v.detach() for v in state["g_vec"]]
        prog.d_vec = [v.detach() for v in state["d_vec"]]
        prog.generated = _detached(state["generated"])
        prog.kp_joined = _detached(state["kp_joined"])
        state.clear()                    # the autograd graph of the capture run: its memory stays the graphs' (private pool)
        self._clear_grads()
        with torch.no_grad():
            for t, c in snap:
                t.copy_(c)
        prog.reg_version = mops.pack_registry_version()
        self.stats["captures"] += 1
        return prog

    # ---- the two calls of the loop -------------------------------------------------------------------------------------------
    def _vectors(self, which):
        cur = self.cur
        if isinstance(cur, _Program):
            return cur.g_vec if which == "g" else cur.d_vec
        return cur["g_vec"] if which == "g" else cur["d_vec"]

    def generator_call(self, x):
        """`generator_full_par(x)` (train.py:110): tuple(loss vectors) + (generated, kp_joined), or NotImplemented"""
        if not self._ready(x):
            self.stats["fallbacks"] += 1
            self.cur = self._last = None
            return NotImplemented
        self.epoch += 1
        dev = self.device
        if self.use_graph:
            key = tuple(x["source"].shape)
            prog = self.programs.get(key)
            if prog is not None and prog.reg_version != mops.pack_registry_version():
                prog = None              # packed-weight buffers were re-created (parameters moved): their addresses are in the graphs
            if prog is None:
                if len(self.programs) >= 4:
                    self.programs.clear()
                prog = self.programs[key] = self._capture(x)
            for k in ("source", "video"):
                prog.x[k].copy_(x[k], non_blocking=True)
            mops.repack_registered(only_if_stale=True)       # a stock optimiser stepped: one pack launch; MnkAdam / AdoptedAdam: none
            for o in self.owners.values():
                o.begin_pass()
            prog.gA.replay()
            mops.bump_inference_epoch()
            self.cur = prog
            generated, kp_joined = dict(prog.generated), dict(prog.kp_joined)
            self.stats["graph_calls"] += 1
        else:
            xd = {k: x[k].to(dev, non_blocking=True) for k in ("source", "video")}
            mops.repack_registered(only_if_stale=True)
            self.cur = st = self._phase_a(xd)
            generated, kp_joined = _detached(st["generated"]), _detached(st["kp_joined"])
            self.stats["phase_calls"] += 1
        self._done = {"g": False, "d": False}
        outs = _PhaseFn.apply(self._anchor, self, "g", self.epoch)
        self._last = {"epoch": self.epoch, "video": x["video"], "prediction": generated["video_prediction"],
                      "kp_joined": kp_joined, "d_versions": [p._version for p in self.d_params]}
        return tuple(outs) + (generated, kp_joined)

    def discriminator_call(self, x, kp_joined, generated):
        """`discriminator_full_par(x, kp_joined, generated)` (train.py:127): the discriminator-loss vectors of THIS iteration's
        forward, or NotImplemented when the arguments are not this iteration's (the wrapped module then runs as it is)"""
        last = self._last
        ok = (last is not None and last["epoch"] == self.epoch and self.cur is not None and torch.is_grad_enabled()
              and self.disc.training and isinstance(x, dict) and x.get("video") is last["video"]
              and isinstance(generated, dict) and generated.get("video_prediction") is last["prediction"]
              and isinstance(kp_joined, dict) and all(kp_joined.get(k) is v for k, v in last["kp_joined"].items())
              and not self._done["d"]
              and all(p._version == v for p, v in zip(self.d_params, last["d_versions"]))
              and all(p.grad is None for p in self.d_params))
        if ok and not self.tp["detach_kp_discriminator"]:
            ok = all(p.grad is None for p in self.k_params)
        if not ok:
            self.stats["d_fallbacks"] += 1
            return NotImplemented
        return list(_PhaseFn.apply(self._anchor, self, "d", self.epoch))

    def _backward(self, which, epoch, grads):
        if epoch != self.epoch or self.cur is None:
            raise RuntimeError("backward through the outputs of an earlier generator_full_par(x) call: the drop-in runner keeps one "
                               "iteration (MNK_DROPIN_GRAPH=0 runs the wrapped modules as they are)")
        if self._done[which]:
            raise RuntimeError("a second backward pass through the same full-model call is not served by the drop-in runner "
                               "(MNK_DROPIN_GRAPH=0 runs the wrapped modules as they are)")
        if which == "d" and not self._done["g"]:
            # the discriminator pass back-propagated first: phase C needs the graph phase B retains -- run B with zero gradients? No:
            raise RuntimeError("discriminator-pass backward before the generator-pass backward of the same iteration is not served "
                               "by the drop-in runner (MNK_DROPIN_GRAPH=0)")
        self._done[which] = True
        cur = self.cur
        if isinstance(cur, _Program):
            static = cur.g_grads if which == "g" else cur.d_grads
            for s, g in zip(static, grads):
                if g is None:
                    s.zero_()
                else:
                    s.copy_(g.reshape(s.shape))
            (cur.gB if which == "g" else cur.gC).replay()
            for p, sink in (cur.grads_g if which == "g" else cur.grads_d):
                if p.grad is None:
                    p.grad = sink
                elif p.grad.data_ptr() != sink.data_ptr():
                    p.grad.add_(sink)
        else:
            vecs = cur["g_vec"] if which == "g" else cur["d_vec"]
            grads = [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, vecs)]
            if which == "g":
                self._phase_b(cur, grads)
            else:
                self._phase_c(cur, grads)
        if which == "d" and not isinstance(cur, _Program):
            self.cur = None              # the iteration's autograd graph may go


_BROKEN = []
_RUNNERS = {}


def _triple(module):
    kp, gen, disc = (getattr(module, n, None) for n in ("kp_extractor", "generator", "discriminator"))
    tp = getattr(module, "train_params", None)
    if not all(isinstance(m, torch.nn.Module) for m in (kp, gen, disc)) or not isinstance(tp, dict) or "loss_weights" not in tp:
        return None
    return kp, gen, disc, tp


def runner_for(module):
    """the TrainPairRunner of a wrapped full model (train.py:24-75), or None"""
    if knobs.get("MNK_DROPIN_GRAPH") == "0":
        return None
    t = _triple(module)
    if t is None:
        return None
    kp, gen, disc, tp = t
    key = (id(kp), id(gen), id(disc))
    r = _RUNNERS.get(key)
    if r is not None and r.kp is kp and r.gen is gen and r.disc is disc:
        r.tp = tp
        return r
    try:
        dev = next(gen.parameters()).device
        from . import _lib
        if (dev.type == "cuda") != bool(_lib.lib().is_device_build):
            return None
        r = TrainPairRunner(kp, gen, disc, tp)
    except (StopIteration, ValueError):
        return None
    _RUNNERS[key] = r
    weakref.finalize(gen, _RUNNERS.pop, key, None)      # (the runner holds the networks weakly: this fires)
    return r


# ---- the reference's per-frame evaluation loops (reconstruction.py:45-62, transfer.py:65-79, demo.py) ------------------------------
class EvalRunner:
    """`DataParallelWithCallback(generator)` / `(kp_detector)` in evaluation mode under no_grad (reconstruction.py:45-49): the
    wrapped network's forward captured ONCE per input signature as a hipGraph with frozen weights -- the packed GEMM layouts and
    the evaluation-mode norm coefficients are made before the capture and only referenced by it (a captured forward of
    mnk.engine.Reconstructor re-packs inside the graph: right for a batch of 512, most of the launches at batch 1) -- and replayed
    per call: inputs are copied into static buffers, outputs are returned as fresh tensors (the loop keeps `kp_source` across
    calls).  Re-captured when a parameter, a buffer or the optimiser epoch changed (load_state_dict between two videos)."""

    MAX_PROGRAMS = 8

    def __init__(self, module):
        self._module = weakref.ref(module)
        self.programs = {}
        self.stats = {"replays": 0, "captures": 0}

    @staticmethod
    def _flatten(obj, out, path=()):
        if torch.is_tensor(obj):
            out.append((path, obj))
            return ("T", tuple(obj.shape), obj.dtype)
        if isinstance(obj, dict):
            return ("D",) + tuple((k, EvalRunner._flatten(obj[k], out, path + (k,))) for k in sorted(obj, key=str))
        if isinstance(obj, (list, tuple)):
            return ("L", type(obj).__name__) + tuple(EvalRunner._flatten(v, out, path + (i,)) for i, v in enumerate(obj))
        return ("V", repr(obj))

    @staticmethod
    def _rebuild(obj, leaves, path=()):
        if torch.is_tensor(obj):
            return leaves[path]
        if isinstance(obj, dict):
            return {k: EvalRunner._rebuild(v, leaves, path + (k,)) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(EvalRunner._rebuild(v, leaves, path + (i,)) for i, v in enumerate(obj))
        return obj

    def _fingerprint(self, module):
        return (mops._PACK_EPOCH[0], mops._BN_EVAL_EPOCH[0], tuple(p._version for p in module.parameters()),
                tuple(b._version for b in module.buffers()))

    def __call__(self, inputs, kwargs, device):
        module = self._module()
        flat = []
        key = self._flatten((inputs, kwargs), flat)
        if not flat or any(t.dtype != torch.float32 for _, t in flat):
            return NotImplemented
        fp = self._fingerprint(module)
        prog = self.programs.get(key)
        if prog is None or prog["fp"] != fp:
            if len(self.programs) >= self.MAX_PROGRAMS:
                self.programs.clear()
            prog = self.programs[key] = self._capture(module, inputs, kwargs, flat, device, fp)
        dsts, srcs = [], []
        for i, ((path, t), s) in enumerate(zip(flat, prog["static"])):
            if t.device.type == "cpu" and not t.is_pinned():
                # a host frame (reconstruction.py:52-58 slices them from the loader's batch): a copy from pageable memory waits for
                # everything queued on the stream -- the previous frame's replay -- before the host may go on.  Through a ring of
                # page-locked staging buffers the copy is asynchronous and the host statements of frame i + 1 overlap frame i.
                buf, ev = self._stage(prog, i, t)
                s.copy_(buf, non_blocking=True)
                ev.record()
            elif t.device == s.device and t.is_contiguous():
                dsts.append(s)
                srcs.append(t)
            else:
                s.copy_(t, non_blocking=True)
        if dsts:                                  # the key points of a generator call (four tensors + the source frame): one launch
            torch._foreach_copy_(dsts, srcs)
        prog["graph"].replay()
        self.stats["replays"] += 1
        return _clone_all(prog["out"])

    STAGES = 4

    @staticmethod
    def _stage(prog, i, t):
        ring = prog.setdefault("ring", {}).get(i)
        if ring is None:
            ring = prog["ring"][i] = {"buf": [torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(EvalRunner.STAGES)],
                                      "ev": [None] * EvalRunner.STAGES, "next": 0}
        k = ring["next"]
        ring["next"] = (k + 1) % EvalRunner.STAGES
        if ring["ev"][k] is not None:
            ring["ev"][k].synchronize()          # the copy that last read this slot (STAGES calls ago) is done
        else:
            ring["ev"][k] = torch.cuda.Event()
        ring["buf"][k].copy_(t)
        return ring["buf"][k], ring["ev"][k]

    def _capture(self, module, inputs, kwargs, flat, device, fp):
        static = [t.to(device).clone() for _, t in flat]
        leaves = {path: s for (path, _), s in zip(flat, static)}
        s_in, s_kw = self._rebuild((inputs, kwargs), leaves)
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):         # the caches are keyed by stream: fill them on the stream the capture runs on
            for _ in range(2):
                module(*s_in, **s_kw)
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        mops.FROZEN_CAPTURE[0] = True
        try:
            with torch.cuda.graph(graph, stream=cap):
                out = module(*s_in, **s_kw)
        finally:
            mops.FROZEN_CAPTURE[0] = False
        self.stats["captures"] += 1
        return {"graph": graph, "static": static, "out": out, "fp": fp}


def _clone_all(obj):
    """fresh tensors with the values of a program's static outputs, copied in ONE launch (a dict of two to four small tensors)"""
    flat = []
    EvalRunner._flatten(obj, flat)
    if len(flat) < 2 or any(not t.is_contiguous() for _, t in flat):
        return _walk_clone(obj)
    fresh = [torch.empty_like(t) for _, t in flat]
    torch._foreach_copy_(fresh, [t for _, t in flat])
    return EvalRunner._rebuild(obj, {path: f for (path, _), f in zip(flat, fresh)})


def _walk_clone(obj):
    if torch.is_tensor(obj):
        return obj.clone()
    if isinstance(obj, dict):
        return {k: _walk_clone(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_walk_clone(v) for v in obj)
    return obj


def eval_runner_for(wrapper):
    """the EvalRunner of an evaluation-mode wrapper around a KPDetector / MotionTransferGenerator on the device, or None"""
    if not knobs.on("MNK_EVAL_GRAPH") or torch.is_grad_enabled() or mdist.initialized():
        return None
    module = wrapper.module
    r = wrapper.__dict__.get("_mnk_eval_runner")
    if r is not None:
        return r if r._module() is module else None
    from modules.generator import MotionTransferGenerator
    from modules.keypoint_detector import KPDetector
    if not isinstance(module, (MotionTransferGenerator, KPDetector)):
        return None
    try:
        if next(module.parameters()).device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            return None
    except StopIteration:
        return None
    r = EvalRunner(module)
    object.__setattr__(wrapper, "_mnk_eval_runner", r)
    return r


def eval_runner_for_wrapper(wrapper):
    """the EvalRunner a wrapper has made (tests, bench.py), or None"""
    return wrapper.__dict__.get("_mnk_eval_runner")
