"""Environment switches of the Python host side, in one place.  Every switch keeps a measured default; the other
setting exists so that an A/B can be repeated (tools/gpu_knob_ab.sh) or the reference's own structure restored.
Values are read from the environment at call time (tests flip them with monkeypatch.setenv).

The measured defaults of the kernel launch plans are NOT environment switches: they are named tuning values inside the
library (`tuning_knob` in monkey-net_amd/csrc/*.hip), set by tuning scripts through the C-ABI (mnk_set_tuning) or, for an
A/B visit, all at once through the single variable MNK_TUNING="name=value,...".  This file is the complete list of the
package's environment switches (+ MNK_BUILD_TAG / MNK_EXTRA_FLAGS of csrc/build.sh)."""
import os

KNOBS = {
    # name: (default, meaning)
    "MNK_LIBRARY": ("", "path of libmonkeynet_hip.so to load instead of the in-tree build (variant builds, A/B)"),
    "MNK_WGRAD_BG": ("10", "eager iterations: giga-MACs of recorded weight-gradient GEMMs after which they are launched on a second "
                           "stream during backward (0: all of them at the end, as a captured iteration always does)"),
    "MNK_DIST_GRAPH": ("1", "with a process group: capture the iteration incl. its RCCL collectives as a hipGraph"),
    "MNK_DIST_FORCE": ("", "1: run the collective code paths even with a single rank (tests, single-GPU RCCL exercise)"),
    "MNK_RCCL_DIRECT": ("1", "nccl backend: SyncBN sums and flat gradient buffers are all-reduced by the library's own RCCL "
                             "communicator on the kernels' stream (0: through torch.distributed)"),
    "MNK_SYNCBN_P2P": ("1", "one node, nccl backend: the SyncBN sums are exchanged by the library's own peer-to-peer kernel over "
                            "IPC-mapped buffers (csrc/p2p.hip) instead of one RCCL all-reduce per norm layer and direction"),
    "MNK_P2P_TIMEOUT_MS": ("120000", "peer-to-peer SyncBN exchange: milliseconds a rank waits for a peer's word before it gives the "
                                     "peer up (the sums then become NaN and mnk.dist.p2p_error() names the rank; TrainStep raises)"),
    "MNK_DP_SCATTER": ("broadcast", "DataParallelWithCallback under a process group: rank 0's batch is broadcast and every rank "
                                     "takes its slice (DataParallel's scatter); slice: trust identical batches; off: no scatter"),
    "MNK_GRAPH_DEADLINE_S": ("", "bench.py under a process group: seconds the hipGraph capture may take before the eager time stands"),
    "MNK_TUNING": ("", "name=value,... for the library's tuning values (read by the library itself; A/B visits)"),
    "MNK_CLIP_VARIANCE_MODE": ("stable", "sigma_min of clip_variance: stable = |det| / sigma_max (finite on nearly singular "
                                         "covariances); reference = the reference's own fp32 sqrt((s1 - s2) / 2), NaNs included"),
    "MNK_DROPIN_GRAPH": ("1", "DataParallelWithCallback around the reference's GeneratorFullModel / DiscriminatorFullModel "
                              "(train.py:104-105): serve `out = generator_full_par(x)`, `loss.backward()` and the discriminator pass "
                              "from three captured hipGraphs behind whole-model autograd Functions (mnk.dropin); phases: the same "
                              "three phases as eager launches; 0: call the wrapped module as it is"),
    "MNK_EVAL_GRAPH": ("1", "DataParallelWithCallback around a KPDetector / MotionTransferGenerator in evaluation mode under no_grad "
                            "(reconstruction.py:45-62's per-frame loop): the forward is captured once per input signature as a "
                            "hipGraph with frozen weights and replayed (mnk.dropin.EvalRunner); 0: eager launches"),
    "MNK_ADOPT_ADAM": ("1", "a stock torch.optim.Adam over a network whose gradients the drop-in runner keeps in one flat buffer is "
                            "stepped by mnk_adam_multi on the optimiser's own state tensors (mnk.optim.AdoptedAdam); 0: the stock step"),
    "MNK_GRAD_OVERLAP": ("1", "MnkAdam: the generator-side gradient exchange runs next to the discriminator backward when there is "
                              "more than one rank (force: also on one rank); GradAverager: a bucket's all-reduce starts when its "
                              "last gradient is written; 0: in-order exchanges"),
}

# Comparison forms -- NOT environment switches (round 4: 25 switches -> 11).  Every entry is a structural choice whose winner
# was measured on the MI355X (profiles/r0*_knob_ab_log.txt) and is what the product runs; the other form stays reachable for
# the TESTS that pin "fused form == the reference's structure" (tests flip an entry with monkeypatch.setitem(knobs.FORMS, ...)).
# Forms that no test needs were deleted outright: MNK_HAND_ADAM (TrainStep(fused_adam=...)), MNK_WGRAD_GROUPED,
# MNK_ADAM_TAP_DIRECT, MNK_PACK_MULTI, MNK_BN_SMALL, MNK_UP_SUBPIXEL, MNK_BN_ZERO_BIAS_GRAD as switches (their winning forms
# are unconditional now).
FORMS = {
    "DISC_BATCHED": True,       # D(generated) and D(real) of a pass as one call on the batch [generated; real]
    "DISC_SHARED": True,        # one discriminator forward per training iteration (False: the reference's two passes)
    "FUSED_FM_LOSS": True,      # feature-matching L1 terms reduced on the device from the NHWC activations
    "WARP_LEVELS": True,        # all warps of a generator pass in one launch each way (False: one launch per decoder level)
    "SKIP_GRAD_FUSED": True,    # hourglass levels: the skip gradient rides in the next data-gradient GEMM's epilogue
    "RES_SKIP_FUSED": True,     # residual blocks: the skip gradient is added inside the first norm layer's dy pass
    "DGRAD_BN_STATS": True,     # the data-gradient GEMM behind a (non-pooling) norm layer leaves that layer's backward statistics
    "EVAL_SPLIT_FUSED": True,   # no_grad + eval: a norm layer behind a split-K convolution sums the partials itself (one launch)
}


def form(name):
    return FORMS[name]


def get(name):
    default, _ = KNOBS[name]
    return os.environ.get(name, default)


def on(name):
    """True unless the switch is set to "0" (switches with default "1"), or only when set to "1" (default "0" / "")."""
    default, _ = KNOBS[name]
    v = os.environ.get(name, default)
    return v != "0" if default == "1" else v == "1"
