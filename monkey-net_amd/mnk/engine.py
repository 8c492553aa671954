"""Host-side mirror of the reference's training iteration (train.py:24-75,110-136) for one rank: the two "full
model" wrappers and the alternating generator / discriminator update with three Adam optimisers.  The reference's
own train.py can equally be used on top of the drop-in `modules` / `sync_batchnorm` packages; this file exists so
that bench.py, the smoke test and the parity tests have a self-contained step that does not import the reference."""
import torch

from modules.losses import generator_loss, discriminator_loss
from . import dist as mdist
from . import knobs
from . import ops as mops


def split_kp(kp_joined, detach=False):
    """Frame 0 of the joined key-points is the source, the rest is the driving video (train.py:14-21)."""
    out = {'kp_driving': {}, 'kp_source': {}}
    for k, v in kp_joined.items():
        if detach:
            v = v.detach()
        # one split node instead of two slices: its backward is one concatenation (two slice nodes: two zero fills, two
        # copies and the sum of the two padded gradients)
        out['kp_source'][k], out['kp_driving'][k] = v.split([1, v.shape[1] - 1], dim=1)
    return out


def joined_kp(kp_detector, x):
    """The key points of [source frame | driving frames] per video (train.py:27 calls the detector on
    torch.cat([source, video], dim=2)).  The detector works per frame, so for one driving frame per video -- every training
    iteration -- it is called on the frames stacked along the BATCH axis instead, and the (2B,1,...) result is viewed as
    (B,2,...): source and driving key points are then two contiguous halves of one buffer, and the split views that every
    consumer asks `.contiguous()` of (two embeddings in the generator, one in the discriminator, the motion field) cost no
    copy launches.  Same frames, same batch statistics (the sums run over the frames in a different order)."""
    src, vid = x['source'], x['video']
    if vid.shape[2] != 1 or src.shape[2] != 1:
        return kp_detector(torch.cat([src, vid], dim=2))
    b = src.shape[0]
    kp = kp_detector(mops.StackedBatch(src, vid) if src.is_contiguous() and vid.is_contiguous()
                     else torch.cat([src, vid], dim=0))
    # the detector saw 2B one-frame samples [all sources | all drivings]: keypoint_indices() of this call must hand its
    # heat-map arg-max out in the (B, 2, K) layout of the key points returned here
    if getattr(kp_detector, "_last_heat", None) is not None:
        heat, k, _, _ = kp_detector._last_heat
        kp_detector._last_heat = (heat, k, b, 2, True)
    return {k: v.view(2, b, *v.shape[2:]).transpose(0, 1) for k, v in kp.items()}


def discriminate_pair(discriminator, fake, real, kp_dict):
    """(D(fake, kp), D(real, kp)) -- train.py:43-45,66-68 call the discriminator twice with the same key-points.
    Every layer of it works per sample (convolutions, InstanceNorm, LeakyReLU, pooling; no batch statistics), so the
    two calls are ONE pass over the batch [fake; real]: half the launches on layers this small (64x64 and below at
    batch 32 are launch / latency bound).  (knobs.FORMS["DISC_BATCHED"] = False: the two separate calls, for the test.)"""
    if not knobs.form("DISC_BATCHED"):
        return discriminator(fake, **kp_dict), discriminator(real, **kp_dict)
    b = fake.shape[0]
    kp2 = {name: {k: torch.cat([v, v], dim=0) for k, v in kp.items()} for name, kp in kp_dict.items()}
    maps = discriminator(torch.cat([fake, real], dim=0), **kp2)
    return [m[:b] for m in maps], [m[b:] for m in maps]


def fused_pair_losses(discriminator, fake, real, kp_dict, video_deformed, loss_weights):
    """generator_loss(...) and discriminator_loss(...) of modules/losses.py for one batched discriminator pass, with the
    feature-matching terms of the down-block outputs reduced on the device from the NHWC activations (ops.PairL1Fn)
    instead of from NCDHW copies of every feature map.  Needs a discriminator with forward_acts (the gfx950-kernel
    one).  Returns (generator loss vectors in generator_loss's order, discriminator loss vectors)."""
    b = fake.shape[0]
    emb = getattr(discriminator, "kp_embedding", None)
    if emb is not None and getattr(emb, "use_deformed_source_image", False):      # (an embedding that reads the frames)
        kp_dict = {name: {k: torch.cat([v, v], dim=0) for k, v in kp.items()} for name, kp in kp_dict.items()}
    w = loss_weights
    rec = w['reconstruction'] if w['reconstruction'] != 0 else []
    taps = {}

    def tap(i, act, c):          # feature-matching term of block output i (map i + 1): taken where the map is made
        if i + 1 < len(rec) and rec[i + 1] != 0:
            act, taps[i + 1] = mops.PairL1TapFn.apply(act, c, b, float(rec[i + 1]))
        return act

    acts, score = discriminator.forward_acts(mops.StackedBatch(fake, real), **kp_dict, tap=tap)
    g_values = []
    if w['reconstruction_deformed'] != 0:
        g_values.append(mops.L1MeanFn.apply(real, video_deformed, w['reconstruction_deformed']))
    if w['reconstruction'] != 0:
        if rec[0] != 0:                               # map 0 is the frame itself
            g_values.append(mops.L1MeanFn.apply(fake, real, rec[0]))
        g_values.extend(taps[i] for i in sorted(taps))
    gen_gan, disc_gan = mops.GanTermsFn.apply(score, b, w['generator_gan'], w['discriminator_gan'])
    g_values.append(gen_gan)
    return g_values, [disc_gan]


class GeneratorFullModel(torch.nn.Module):
    """train.py:24-53."""

    def __init__(self, kp_extractor, generator, discriminator, train_params):
        super(GeneratorFullModel, self).__init__()
        self.kp_extractor = kp_extractor
        self.generator = generator
        self.discriminator = discriminator
        self.train_params = train_params

    def forward(self, x):
        kp_joined = joined_kp(self.kp_extractor, x)
        generated = self.generator(x['source'], **split_kp(kp_joined, self.train_params['detach_kp_generator']))
        kp_dict = split_kp(kp_joined, False)
        maps_generated, maps_real = discriminate_pair(self.discriminator, generated['video_prediction'], x['video'],
                                                      kp_dict)
        generated.update(kp_dict)
        losses = generator_loss(discriminator_maps_generated=maps_generated, discriminator_maps_real=maps_real,
                                video_deformed=generated['video_deformed'],
                                loss_weights=self.train_params['loss_weights'])
        return tuple(losses) + (generated, kp_joined)


class DiscriminatorFullModel(torch.nn.Module):
    """train.py:56-75."""

    def __init__(self, kp_extractor, generator, discriminator, train_params):
        super(DiscriminatorFullModel, self).__init__()
        self.kp_extractor = kp_extractor
        self.generator = generator
        self.discriminator = discriminator
        self.train_params = train_params

    def forward(self, x, kp_joined, generated):
        kp_dict = split_kp(kp_joined, self.train_params['detach_kp_discriminator'])
        maps_generated, maps_real = discriminate_pair(self.discriminator, generated['video_prediction'].detach(),
                                                      x['video'], kp_dict)
        return discriminator_loss(discriminator_maps_generated=maps_generated, discriminator_maps_real=maps_real,
                                  loss_weights=self.train_params['loss_weights'])


_BROKEN_CAPTURES = []


class TrainStep:
    """One iteration of train.py:110-136 on this rank's shard, with gradients averaged over ranks (RCCL) before
    every optimiser step."""

    def __init__(self, generator, discriminator, kp_detector, train_params, fused_adam=None, use_graph=False):
        """fused_adam: None / True -- mnk.optim.MnkAdam (one hand-written launch per optimiser step that also emits the
        packed conv weights of the next iteration; weight-gradient split reductions of all layers in one launch; the flat
        gradient buffer is what the ranks all-reduce); False -- stock torch.optim.Adam (non-fused) + per-layer
        reductions + bucketed GradAverager, the structure of round 1, kept as the comparison path of the tests."""
        self.generator, self.discriminator, self.kp_detector = generator, discriminator, kp_detector
        self.tp = train_params
        lr = train_params['lr']
        self.use_graph = bool(use_graph)
        self.mnk_adam = True if fused_adam is None else bool(fused_adam)
        self._graph = None                    # the captured iteration: a list of hipGraphs and host calls between them
        self._segment = None                  # (graph being captured, pool) while _capture runs
        self._ones = {}
        self._replaying = False               # inside step()'s replay of a captured iteration (host calls look at it)
        self._static_x = None
        self._static_out = None
        self._weights_touched = True          # packed weights must be (re)made before the next iteration
        self._shard_checked = None            # batch size the ranks were last found to agree on (mnk.dist.check_equal_shards)
        self._iterations = 0
        if self.mnk_adam:
            from . import optim as moptim
            self.opt_g = moptim.MnkAdam(generator.parameters(), lr=lr, betas=(0.5, 0.999))
            self.opt_d = moptim.MnkAdam(discriminator.parameters(), lr=lr, betas=(0.5, 0.999))
            self.opt_k = moptim.MnkAdam(kp_detector.parameters(), lr=lr, betas=(0.5, 0.999))
            for m in (generator, discriminator, kp_detector):     # a checkpoint load invalidates the packed copies
                m.register_load_state_dict_post_hook(lambda *a, **k: self.weights_changed())
        else:
            kw = {'capturable': True} if self.use_graph else {}
            self.opt_g = torch.optim.Adam(generator.parameters(), lr=lr, betas=(0.5, 0.999), **kw)
            self.opt_d = torch.optim.Adam(discriminator.parameters(), lr=lr, betas=(0.5, 0.999), **kw)
            self.opt_k = torch.optim.Adam(kp_detector.parameters(), lr=lr, betas=(0.5, 0.999), **kw)
        self.gfull = GeneratorFullModel(kp_detector, generator, discriminator, train_params)
        self.dfull = DiscriminatorFullModel(kp_detector, generator, discriminator, train_params)
        self.avg_gk = _Averager(self, (self.opt_g, self.opt_k),
                                list(generator.parameters()) + list(kp_detector.parameters()))
        self.avg_d = _Averager(self, (self.opt_d,), list(discriminator.parameters()))
        self.avg_k = _Averager(self, (self.opt_k,), list(kp_detector.parameters()))

    def _one(self, like):
        t = self._ones.get(like.device)
        if t is None:
            t = self._ones[like.device] = torch.ones((), dtype=torch.float32, device=like.device)
        return t

    def weights_changed(self):
        """Parameters were written from outside (load_state_dict does this by itself): the packed GEMM layouts are
        re-made before the next iteration."""
        self._weights_touched = True
        mops.invalidate_packed_weights()

    def _begin_iteration(self):
        # every conv parameter seen so far packed in one launch -- unless the optimiser kernel of the previous iteration
        # already wrote the layouts (mnk.optim.MnkAdam)
        mops.repack_registered(only_if_stale=self.mnk_adam)
        mops.clear_dz_stats()

    def step(self, x):
        """Eager iteration, or -- with use_graph -- a replay of the whole iteration captured once as a hipGraph
        (several hundred small launches per iteration make the eager path launch/host bound).  The first call captures:
        it runs warm-up iterations to size buffers and create optimiser state, then puts parameters, buffers and
        optimiser state back, so that every call -- the first one included -- applies exactly one update.  With
        use_graph the returned losses and `generated` are static buffers that the next call overwrites: clone what
        must survive."""
        self._iterations += 1
        if mdist._P2P["handle"] is not None and self._iterations % 32 == 0:
            mdist.check_p2p()                 # a peer that an exchange gave up on: raise, do not train on poisoned statistics
        if not self.use_graph:
            return self._eager_step(x)
        if self._graph is None:
            self._capture(x)
        if self.mnk_adam:
            if self._weights_touched:         # e.g. a checkpoint was loaded between two replays
                mops.repack_registered()
            for opt in (self.opt_g, self.opt_d, self.opt_k):
                opt.sync_scalars()            # a scheduler's new learning rate reaches the device scalars
        self._weights_touched = False
        for k in self._static_x:
            self._static_x[k].copy_(x[k], non_blocking=True)
        self._replaying = True
        try:
            for piece in self._graph:         # one hipGraph, or linear hipGraphs with the gradient exchange's host calls between
                piece.replay() if isinstance(piece, torch.cuda.CUDAGraph) else piece()
        finally:
            self._replaying = False
        mops.invalidate_packed_weights()      # the captured optimiser steps changed the parameters
        return self._static_out

    def _capture(self, x, warmup=3):
        # With several ranks the captured iteration contains the RCCL all-reduces (SyncBN sums, gradient buckets):
        # every rank replays the same sequence -- collectives inside hipGraphs, as the hipGraph-captured serving stacks
        # on this hardware use them.  Exercised on the MI355X with a forced single-rank process group
        # (MNK_DIST_FORCE=1: 16.0 ms per iteration against 17.7 ms eager); MNK_DIST_GRAPH=0 opts out.
        assert not mdist.active() or knobs.on("MNK_DIST_GRAPH"), \
            "graph capture with torch.distributed active was disabled (MNK_DIST_GRAPH=0)"
        self._static_x = {k: v.clone() for k, v in x.items()}
        snap = self._snapshot()
        import gc
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for it in range(warmup + 1):     # sizes the scratch buffers, creates Adam state and the descriptor tables
                if it == warmup:
                    # the captured iteration of one process: the optimiser kernels read the gradients of the deep levels from
                    # the tap-major partials themselves (MnkAdam.tap_direct; p.grad is not observable inside a replay)
                    direct = self.mnk_adam and not mdist.grads_active()
                    for o in (self.opt_g, self.opt_d, self.opt_k):
                        if hasattr(o, "tap_direct"):
                            o.tap_direct = bool(direct)
                    # what torch.cuda.graph() does before a capture -- but in front of the LAST warm-up iteration: objects
                    # of earlier runs that only the cycle collector frees (models, optimisers) take their packed-weight
                    # registrations with them, and the tables the capture must find unchanged are keyed by that registry
                    torch.cuda.synchronize()
                    gc.collect()
                    torch.cuda.empty_cache()
                self._eager_step(self._static_x, set_to_none=True)
        torch.cuda.current_stream().wait_stream(side)
        # The iteration is captured as LINEAR hipGraphs.  A captured graph with a second branch that holds a kernel replays
        # 0.85-1.0 ms slower on this runtime, whatever the branch does (profiles/r03_knob_ab_log.txt), so the overlapped
        # gradient exchange of several ranks is not a branch of one graph: _cut() ends the graph in front of it, the
        # exchange is started / awaited by ordinary stream calls at replay time, and a new graph continues behind it.
        torch.cuda.synchronize()
        gc_was_on = gc.isenabled()
        gc.disable()                          # (no collection -- no finaliser -- in the middle of the capture either)
        program = []
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(cap):
                self._segment = [None, pool, program]
                self._segment_begin()
                self._static_out = self._eager_step(self._static_x, set_to_none=True)
                if self._segment[0] is not None:
                    self._segment_end()
        except BaseException:
            if self._segment[0] is not None:          # leave the stream out of capture mode, whatever went wrong
                try:
                    self._segment[0].capture_end()
                except Exception:
                    _BROKEN_CAPTURES.append(self._segment[0])     # (its destructor would abort the process: keep it)
            raise
        finally:
            self._segment = None
            if gc_was_on:
                gc.enable()
            for o in (self.opt_g, self.opt_d, self.opt_k):      # eager iterations keep every p.grad valid
                if hasattr(o, "tap_direct"):
                    o.tap_direct = False
        torch.cuda.current_stream().wait_stream(cap)
        self._graph = program
        self._restore(snap)

    def _segment_begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self._segment[1])
        self._segment[0] = g

    def _segment_end(self):
        g, self._segment[0] = self._segment[0], None
        g.capture_end()
        self._segment[2].append(g)

    def _cut(self, host_call, last=False):
        """`host_call()` now -- and, when the iteration is being captured, between two hipGraphs at every replay (last: nothing
        that launches follows in this iteration, no further graph is begun)."""
        if self._segment is None:
            host_call()
            return
        self._segment_end()
        self._segment[2].append(host_call)
        host_call()
        if not last:
            self._segment_begin()

    def _snapshot(self):
        """Everything the warm-up iterations of the capture change: parameters, buffers (BatchNorm running statistics),
        optimiser state.  Restored IN PLACE (the captured graph holds these addresses)."""
        mods = (self.generator, self.discriminator, self.kp_detector)
        tensors = [t for m in mods for t in list(m.parameters()) + list(m.buffers())]
        snap = {"tensors": [(t, t.detach().clone()) for t in tensors]}
        opts = (self.opt_g, self.opt_d, self.opt_k)
        if self.mnk_adam:
            snap["opt"] = [(o.flat_m.clone(), o.flat_v.clone(), o.hyper.clone()) for o in opts]
        else:
            snap["opt"] = [{p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                            for p, st in o.state.items()} for o in opts]
        return snap

    def _restore(self, snap):
        with torch.no_grad():
            for t, c in snap["tensors"]:
                t.copy_(c)
            for o, saved in zip((self.opt_g, self.opt_d, self.opt_k), snap["opt"]):
                if self.mnk_adam:
                    o.flat_m.copy_(saved[0])
                    o.flat_v.copy_(saved[1])
                    # only what the warm-up's steps advanced: bias corrections [4:6] and the step counter [7].  The learning
                    # rate [0] and the gradient scale [6] = 1 / world size were put there by sync_scalars() during the
                    # warm-up and their host mirrors say so: a captured step never re-syncs them, so restoring the
                    # snapshot's 1.0 would make every replay SUM the ranks' gradients (exp_avg ws x, exp_avg_sq ws^2 x)
                    o.hyper[4:6].copy_(saved[2][4:6])
                    o.hyper[7:8].copy_(saved[2][7:8])
                    continue
                for p, st in o.state.items():           # state created by the warm-up: back to its initial zeros
                    for k, v in st.items():
                        if torch.is_tensor(v):
                            v.copy_(saved[p][k]) if p in saved else v.zero_()
        self.weights_changed()                          # the packed copies belong to the warm-up's parameters

    def _eager_step(self, x, set_to_none=True):
        b = int(x['source'].shape[0])
        # count = local * world (SyncBN) and the 1 / world gradient scale need equal shards.  The check is a collective, so the
        # decision to enter it must not depend on this rank's own history: EVERY eager iteration of a process group checks (a
        # rank whose batch did not change would otherwise walk into the SyncBN collectives while another rank sits in this
        # one -- a hang instead of the ValueError).  One 8-byte MAX-reduce per eager iteration; a captured
        # iteration is checked when it is captured (its batch size is frozen with the graph).
        if mdist.initialized() and not (x['source'].is_cuda and torch.cuda.is_current_stream_capturing()):
            mdist.check_equal_shards(b)
            self._shard_checked = b
        self._weights_touched = False
        if knobs.form("DISC_SHARED"):
            return self._eager_step_shared(x)
        return self._eager_step_two_pass(x)

    def _eager_step_shared(self, x):
        """train.py:110-136 with ONE discriminator forward per iteration.  The reference runs the discriminator on
        (generated, real) in the generator pass and again, on the same frames with the same (not yet updated)
        discriminator weights, in the discriminator pass (train.py:43-45,66-68): the second forward recomputes the
        first one's values.  Here the graph is cut at the discriminator's inputs (generated frames and key-points become
        leaves), so that one forward serves both losses:
          1. dL_G / d(generated, key-points) by a backward through the discriminator only (no weight gradients);
          2. one backward through generator + key-point detector seeded with those gradients (and L_G's direct terms);
          3. after the generator step, dL_D / d(discriminator weights) through the same retained graph (the generated
             frames are a leaf there -- the reference's `.detach()`); key-point gradients of L_D continue into the
             detector unless train_params['detach_kp_discriminator'].
        Same losses and gradients as the two-pass form (tests/test_step.py), one discriminator forward less."""
        tp = self.tp
        self._begin_iteration()
        g_params = list(self.generator.parameters())
        k_params = list(self.kp_detector.parameters())
        d_params = list(self.discriminator.parameters())
        kp_joined = joined_kp(self.kp_detector, x)
        generated = self.generator(x['source'], **split_kp(kp_joined, tp['detach_kp_generator']))
        fake = generated['video_prediction']
        fake_leaf = fake.detach().requires_grad_(True)
        kp_names = list(kp_joined.keys())
        kp_leaf = {k: kp_joined[k].detach().requires_grad_(True) for k in kp_names}
        if knobs.form("FUSED_FM_LOSS") and hasattr(self.discriminator, "forward_acts"):
            # feature-matching terms straight from the NHWC activations (profiles/README.md: 14.59 -> 14.26 ms per step)
            g_vec, d_vec = fused_pair_losses(self.discriminator, fake_leaf, x['video'], split_kp(kp_leaf, False),
                                             generated['video_deformed'], tp['loss_weights'])
            generated.update(split_kp(kp_joined, False))
            # batch means of all terms and their sum in one launch (the reference: [val.mean() for val in losses], train.py:114)
            g_means, g_total_fused = mops.LossMeansFn.apply(*g_vec)
            d_means, d_total_fused = mops.LossMeansFn.apply(*d_vec)
            loss_values, d_values = list(g_means.unbind(0)), list(d_means.unbind(0))
        else:
            g_total_fused = d_total_fused = None
            maps_generated, maps_real = discriminate_pair(self.discriminator, fake_leaf, x['video'],
                                                          split_kp(kp_leaf, False))
            generated.update(split_kp(kp_joined, False))
            loss_values = [v.mean() for v in generator_loss(
                discriminator_maps_generated=maps_generated, discriminator_maps_real=maps_real,
                video_deformed=generated['video_deformed'], loss_weights=tp['loss_weights'])]
            d_values = [v.mean() for v in discriminator_loss(
                discriminator_maps_generated=maps_generated, discriminator_maps_real=maps_real,
                loss_weights=tp['loss_weights'])]
        g_total = g_total_fused if g_total_fused is not None else sum(loss_values)
        # 1. through the discriminator only
        leaves = [fake_leaf] + [kp_leaf[k] for k in kp_names]
        one = self._one(g_total)                  # the root gradient, made once (autograd's default: a fill launch per call)
        with mops.no_param_grads():
            seeds = torch.autograd.grad(g_total, leaves, grad_outputs=one, retain_graph=True, allow_unused=True)
        # 2. generator + key-point detector, one traversal
        roots, root_grads = [], []
        for t, g in zip([fake] + [kp_joined[k] for k in kp_names], seeds):
            if g is not None and t.requires_grad:
                roots.append(t)
                root_grads.append(g)
        if tp['loss_weights']['reconstruction_deformed'] != 0:      # the only term of L_G that does not pass the cut
            roots.append(g_total)
            root_grads.append(one)
        self.avg_gk.arm()
        torch.autograd.backward(roots, root_grads, inputs=g_params + k_params,
                                retain_graph=not tp['detach_kp_discriminator'])
        self.avg_gk.average()
        # Several ranks, MnkAdam: the exchange of the generator's (and key-point detector's) flat gradient buffers starts here
        # on the communication stream, and the optimiser steps that consume them move BEHIND the discriminator-loss backward
        # (step 3 runs on the retained discriminator graph with the generated frames as leaves: it reads neither the
        # generator's weights nor its update) -- 265-379 MB per iteration leave the critical path.  Same arithmetic, same
        # order of the three updates' inputs; MNK_GRAD_OVERLAP=0 keeps the in-order exchange.
        # (one rank of a forced process group has nothing to hide: in-order unless MNK_GRAD_OVERLAP=force, the GPU test's setting)
        ov = knobs.get("MNK_GRAD_OVERLAP")
        overlap = self.mnk_adam and mdist.grads_active() and (ov == "force" or (ov != "0" and mdist.world_size() > 1))
        step_k_now = tp['detach_kp_discriminator']

        def step_generator_side():
            self.opt_g.step()
            self.opt_g.zero_grad()
            if step_k_now:
                self.opt_k.step()
                self.opt_k.zero_grad()

        if overlap:
            ex = [self.opt_g] + ([self.opt_k] if step_k_now else [])
            for o in ex:
                o.materialize_grads()                               # (captured) every gradient in its flat buffer
            self._cut(lambda: [o.exchange_begin() for o in ex])    # (host call) the sums start on the communication stream
        else:
            step_generator_side()
        self.opt_d.zero_grad()
        # 3. the discriminator loss through the retained discriminator graph
        self.avg_d.arm()
        d_total = d_total_fused if d_total_fused is not None else sum(d_values)
        if tp['detach_kp_discriminator']:
            with mops.no_leaf_input_grads():     # nothing below the discriminator's first convolution is asked for
                torch.autograd.backward(d_total, one, inputs=d_params)
        else:
            self.avg_k.arm()
            kl = [kp_leaf[k] for k in kp_names]
            for t in kl:
                t.grad = None
            torch.autograd.backward(d_total, one, inputs=d_params + kl)
            back = [(kp_joined[k], kp_leaf[k].grad) for k in kp_names if kp_leaf[k].grad is not None]
            if back:
                torch.autograd.backward([t for t, _ in back], [g for _, g in back], inputs=k_params)
        self.avg_d.average()
        self.opt_d.step()
        self.opt_d.zero_grad()
        if not tp['detach_kp_discriminator']:
            self.avg_k.average()
            self.opt_k.step()
            self.opt_k.zero_grad()
        if overlap:
            # (host call, the LAST piece of a captured iteration: two hipGraphs, not three -- every further graph launch costs
            # 0.35 ms) the kernels' stream waits for the sums, then the generator-side updates run as plain launches: at
            # replay time from the optimisers' cached descriptor tables (tick + one Adam launch each)
            def finish():
                for o in ex:
                    o.exchange_end()
                if self._replaying:
                    for o in ex:
                        o.replay_step()
                else:
                    step_generator_side()
            self._cut(finish, last=True)
        return [v.detach() for v in loss_values], [v.detach() for v in d_values], generated

    def _eager_step_two_pass(self, x, set_to_none=True):
        """The reference's structure: generator pass and discriminator pass each run the discriminator
        (knobs.FORMS["DISC_SHARED"] = False: the comparison form of tests/test_step.py)."""
        tp = self.tp
        # The generator pass back-propagates THROUGH the discriminator but the reference throws the discriminator's own
        # weight gradients of this pass away (optimizer_discriminator.zero_grad(), train.py:120): do not compute them.
        d_params = list(self.discriminator.parameters())
        for p in d_params:
            p.requires_grad_(False)
        self._begin_iteration()
        try:
            out = self.gfull(x)
            loss_values = [v.mean() for v in out[:-2]]
            generated, kp_joined = out[-2], out[-1]
            self.avg_gk.arm()
            sum(loss_values).backward(retain_graph=not tp['detach_kp_discriminator'])
        finally:
            for p in d_params:
                p.requires_grad_(True)
        self.avg_gk.average()
        self.opt_g.step()
        self.opt_g.zero_grad()
        self.opt_d.zero_grad()
        if tp['detach_kp_discriminator']:
            self.opt_k.step()
            self.opt_k.zero_grad()
        d_values = [v.mean() for v in self.dfull(x, kp_joined, generated)]
        self.avg_d.arm()
        if not tp['detach_kp_discriminator']:
            self.avg_k.arm()
        sum(d_values).backward()
        self.avg_d.average()
        self.opt_d.step()
        self.opt_d.zero_grad()
        if not tp['detach_kp_discriminator']:
            self.avg_k.average()
            self.opt_k.step()
            self.opt_k.zero_grad()
        return [v.detach() for v in loss_values], [v.detach() for v in d_values], generated


class _Averager:
    """What happens between a backward pass and the optimiser steps that consume it.  torch optimisers: the bucketed,
    overlapped mnk.dist.GradAverager.  MnkAdam: every gradient lands in the optimiser's flat buffer (one reduction launch
    for all weight-gradient partials + one multi-tensor gather) -- the exchange itself is part of MnkAdam.step."""

    def __init__(self, step, opts, params):
        self.opts = opts
        self.legacy = None if step.mnk_adam else mdist.GradAverager(params)
        if self.legacy is not None:
            for o in opts:       # mnk.dist's generic pre-step hook (installed by DataParallelWithCallback) must not average these
                o._mnk_owns_exchange = True      # gradients a second time: GradAverager.average() already has

    def arm(self):
        if self.legacy is not None:
            self.legacy.arm()

    def average(self):
        if self.legacy is not None:
            return self.legacy.average()
        for opt in self.opts:
            opt.materialize_grads()
        return 0


class Reconstructor:
    """Batched eval-mode frame generation -- what reconstruction.py:12-25,57-61 / transfer.py:65-79 do frame by frame:
    key-points of the source and of every driving frame, then one generator call per (source, driving) pair.  In eval
    mode BatchNorm uses running statistics, so frames are independent and (video, frame) folds into the batch.  With
    use_graph the whole forward (kp detector x2 + generator, ~250 launches + the weight packs) is captured once as a
    hipGraph and replayed per batch (BASELINE config 5: "hipGraph-captured generator forward").  The returned tensors of
    the graph form are static buffers that the next call overwrites: clone what must survive."""

    def __init__(self, kp_detector, generator, use_graph=False):
        self.kp_detector, self.generator = kp_detector.eval(), generator.eval()
        self.use_graph = bool(use_graph)
        self._graph = None
        self._static_in = None
        self._static_out = None

    @torch.no_grad()
    def _forward(self, source, driving):
        kp_source = self.kp_detector(source)
        kp_driving = self.kp_detector(driving)
        out = self.generator(source, kp_driving=kp_driving, kp_source=kp_source)
        return {"video_prediction": out["video_prediction"], "video_deformed": out["video_deformed"],
                "kp_driving_mean": kp_driving["mean"], "kp_source_mean": kp_source["mean"]}

    def __call__(self, source, driving):
        if not self.use_graph:
            return self._forward(source, driving)
        if self._graph is None or self._static_in[0].shape != source.shape:
            # the captured forward contains its own weight-pack launches (ops._packed_fwd_weight under capture), so a
            # replay always runs on the live parameters: optimiser steps, load_state_dict and in-place writes need no
            # re-capture
            self._static_in = (source.clone(), driving.clone())
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._forward(*self._static_in)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static_out = self._forward(*self._static_in)
        self._static_in[0].copy_(source, non_blocking=True)
        self._static_in[1].copy_(driving, non_blocking=True)
        self._graph.replay()
        return self._static_out


def normalize_kp(kp_video, kp_appearance, movement_mult=False, move_location=False, adapt_variance=False, clip_mean=False):
    """transfer.py:31-62 on the device, same arguments and return value (a new key-point dict): the convex-hull areas
    (transfer.py:34-36; the reference copies the key-points to the host for scipy), the re-centring on the source's
    key-points and the covariance transfer with its symmetric positive-definite repair are two kernel launches, no
    host synchronisation."""
    mean_v = kp_video['mean'].contiguous().float()
    mops._check_device(mean_v)
    b, d, k, _ = mean_v.shape
    mean_a = kp_appearance['mean'].contiguous().float()
    has_var = 'var' in kp_video
    var_v = kp_video['var'].contiguous().float() if has_var else None
    var_a = kp_appearance['var'].contiguous().float() if has_var and 'var' in kp_appearance else None
    adapt = bool(adapt_variance and has_var)
    if clip_mean and not move_location:
        raise NameError("clip_mean without move_location is an error in the reference too (transfer.py:49)")
    if not (move_location or adapt):
        return {key: v for key, v in kp_video.items()}
    area_a = area_v = None
    if movement_mult:
        area_a = torch.empty(1, dtype=torch.float32, device=mean_v.device)
        area_v = torch.empty(1, dtype=torch.float32, device=mean_v.device)
        mops._call("mnk_kp_hull_area", mean_v, mops._p(mean_a), k, mops._p(area_a))      # kp_appearance['mean'][0, 0]
        mops._call("mnk_kp_hull_area", mean_v, mops._p(mean_v), k, mops._p(area_v))      # kp_video['mean'][0, 0]
    mean_out = torch.empty_like(mean_v)
    var_out = torch.empty_like(var_v) if has_var else None
    mops._call("mnk_kp_normalize", mean_v, mops._p(mean_v), mops._p(var_v), mops._p(mean_a), mops._p(var_a), b, d, k,
               mops._p(area_a), mops._p(area_v), int(bool(move_location)), int(bool(clip_mean)), int(adapt), mops._p(mean_out),
               mops._p(var_out))
    out = {key: v for key, v in kp_video.items()}
    out['mean'] = mean_out
    if has_var:
        out['var'] = var_out
    return out


class Transfer:
    """transfer.py:65-79 transfer_one without its per-frame Python loops: ONE key-point detector call over all driving
    frames (the detector folds the time axis into the batch), normalize_kp on the device, and ONE generator call with the
    (video, frame) pairs folded into the batch.  Eval mode (running BatchNorm statistics), so frames are independent.
    Returns the dict transfer_one returns."""

    def __init__(self, kp_detector, generator, normalization_params):
        self.kp_detector, self.generator = kp_detector.eval(), generator.eval()
        self.params = dict(normalization_params)

    @torch.no_grad()
    def __call__(self, source_image, driving_video):
        b, c, d, h, w = driving_video.shape
        kp_driving = self.kp_detector(driving_video)                     # (B, d, K, .)
        kp_source = self.kp_detector(source_image)                       # (B, 1, K, .)
        kp_norm = normalize_kp(kp_driving, kp_source, **self.params)
        fold = lambda t: t.reshape((b * d, 1) + t.shape[2:])             # frame f of video v -> batch entry v * d + f
        rep = lambda t: t.repeat_interleave(d, dim=0)
        out = self.generator(rep(source_image), kp_driving={k: fold(v) for k, v in kp_norm.items()},
                             kp_source={k: rep(v) for k, v in kp_source.items()})
        unfold = lambda t: t.reshape(b, d, c, h, w).permute(0, 2, 1, 3, 4)       # (B*d, C, 1, H, W) -> (B, C, d, H, W)
        return {"video_prediction": unfold(out["video_prediction"]), "video_deformed": unfold(out["video_deformed"]),
                "kp_driving": kp_driving, "kp_source": kp_source, "kp_norm": kp_norm}
