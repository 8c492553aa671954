"""Environment switches of the Python host side, in one place.  Every switch keeps a measured default; the other
setting exists so that an A/B can be repeated (tools/gpu_knob_ab.sh) or the reference's own structure restored.
Values are read from the environment at call time (tests flip them with monkeypatch.setenv).

The measured defaults of the kernel launch plans are NOT environment switches: they are named tuning values inside the
library (`tuning_knob` in monkey-net_amd/csrc/*.hip), set by tuning scripts through the C-ABI (mnk_set_tuning) or, for an
A/B visit, all at once through the single variable MNK_TUNING="name=value,...".  This file is the complete list of the
package's environment switches (+ MNK_TUNING, and MNK_BUILD_TAG / MNK_EXTRA_FLAGS of csrc/build.sh)."""
import os

KNOBS = {
    # name: (default, meaning)
    "MNK_LIBRARY": ("", "path of libmonkeynet_hip.so to load instead of the in-tree build (variant builds, A/B)"),
    "MNK_DISC_BATCHED": ("1", "D(generated) and D(real) of a pass as one call on the batch [generated; real]"),
    "MNK_DISC_SHARED": ("1", "one discriminator forward per training iteration (0: the reference's two passes)"),
    "MNK_FUSED_FM_LOSS": ("1", "feature-matching L1 terms reduced on the device from the NHWC activations (14.59 -> 14.26 ms/step)"),
    "MNK_HAND_ADAM": ("1", "TrainStep default optimiser = mnk.optim.MnkAdam (one launch, emits the packed weights; deferred "
                           "weight-gradient reductions); 0: torch.optim.Adam(fused=True) + per-layer reductions (round 1)"),
    "MNK_WGRAD_GROUPED": ("1", "MnkAdam pipeline: the tap-major weight-gradient GEMMs of all layers in one launch per tile "
                               "shape at the end of backward (0: one launch per layer during backward)"),
    "MNK_WGRAD_BG": ("10", "eager iterations: giga-MACs of recorded weight-gradient GEMMs after which they are launched on a second "
                           "stream during backward (0: all of them at the end, as a captured iteration always does)"),
    "MNK_REPLAY_STREAMS": ("0", "captured iteration of one process: 0 = hipGraphLaunch; n >= 1 = the library's stream executor "
                                "(csrc/replay.hip) on at most n streams; n >= 2 also keeps the background weight-gradient "
                                "launches of the backward pass as a branch of the captured graph"),
    "MNK_ADAM_TAP_DIRECT": ("1", "captured iteration of one process: the optimiser kernel reads the gradients of the few-split "
                                 "tap-major layers from their partials (no reduction pass for them; p.grad of those parameters "
                                 "is not written)"),
    "MNK_WARP_LEVELS": ("1", "all warps (and key-point embedding copies) of a generator pass in one launch each way (0: one "
                             "launch per decoder level)"),
    "MNK_UP_SUBPIXEL": ("1", "UpBlock3D convolutions in their sub-pixel forms (four 2x2 phase convolutions forward, one 4x4 "
                             "stride-2 convolution for the data gradient; 0: 3x3 over the up-sampled view + sum-pool)"),
    "MNK_BN_ZERO_BIAS_GRAD": ("1", "the bias of a convolution in front of a training-mode BatchNorm gets no gradient (it is "
                                   "analytically zero; 0: compute the rounding noise the reference computes)"),
    "MNK_BN_SMALL": ("1", "small layers (<= 512 pixel rows): split-K sum + BatchNorm statistics + finalisation + apply in one "
                          "launch, and the backward in one launch (0: the general multi-launch forms; measured 12.39 vs 12.42 ms)"),
    "MNK_PACK_MULTI": ("1", "re-pack every conv weight of the model in one launch per iteration (0: one launch per layer)"),
    "MNK_DIST_GRAPH": ("1", "with a process group: capture the iteration incl. its RCCL collectives as a hipGraph"),
    "MNK_DIST_FORCE": ("", "1: run the collective code paths even with a single rank (tests, single-GPU RCCL exercise)"),
    "MNK_RCCL_DIRECT": ("1", "nccl backend: SyncBN sums and flat gradient buffers are all-reduced by the library's own RCCL "
                             "communicator on the kernels' stream (0: through torch.distributed)"),
    "MNK_DP_SCATTER": ("broadcast", "DataParallelWithCallback under a process group: rank 0's batch is broadcast and every rank "
                                     "takes its slice (DataParallel's scatter); slice: trust identical batches; off: no scatter"),
    "MNK_GRAPH_DEADLINE_S": ("", "bench.py under a process group: seconds the hipGraph capture may take before the eager time stands"),
    "MNK_TUNING": ("", "name=value,... for the library's tuning values (read by the library itself; A/B visits)"),
    "MNK_CLIP_VARIANCE_MODE": ("stable", "sigma_min of clip_variance: stable = |det| / sigma_max (finite on nearly singular "
                                         "covariances); reference = the reference's own fp32 sqrt((s1 - s2) / 2), NaNs included"),
    "MNK_GRAD_OVERLAP": ("1", "MnkAdam: the generator-side gradient exchange runs next to the discriminator backward when there is "
                              "more than one rank (force: also on one rank); GradAverager: a bucket's all-reduce starts when its "
                              "last gradient is written; 0: in-order exchanges"),
    "MNK_SKIP_GRAD_FUSED": ("1", "hourglass levels: the next down block's data-gradient GEMM adds the gradient of the level's other "
                                 "consumer (decoder skip / warp) in its epilogue (0: autograd accumulates the two gradients)"),
    "MNK_RES_SKIP_FUSED": ("1", "residual blocks: the first norm layer's backward adds the skip gradient in its dy pass (0: autograd "
                                "accumulates the two gradients of the block's input in a pass of its own)"),
}


def get(name):
    default, _ = KNOBS[name]
    return os.environ.get(name, default)


def on(name):
    """True unless the switch is set to "0" (switches with default "1"), or only when set to "1" (default "0" / "")."""
    default, _ = KNOBS[name]
    v = os.environ.get(name, default)
    return v != "0" if default == "1" else v == "1"
