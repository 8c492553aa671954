"""Patch discriminator of the training step (modules/discriminator.py), first "next" row of SURVEY.md section 8f, on the
same gfx950 kernels as the hot path; `modules.discriminator.Discriminator` resolves to this class (the only backend: the
stock-op twin of round 1 and its switch were deleted in round 3).

Measured on the MI355X in round 1 (profiles/README.md): its layers are tiny (61x61x13 -> ... -> 2x2x256 at batch 32), so
with four separate passes per iteration it only matched a stock-op network (MIOpen Winograd / implicit-GEMM kernels):
16.48 vs 16.47 ms per moving-gif iteration.  With the generated and the real frames of a pass batched into one call
(mnk.engine.discriminate_pair -- every layer here is per sample) it wins: 15.41 vs 15.62 ms.

Kernels: the (1,4,4) convolutions without padding run on the implicit-GEMM conv kernels
(K x K form), InstanceNorm + LeakyReLU(0.2) + avg-pool is one fused pass (per-frame statistics), the score head is
the linear 1x1 kernel.  Constructor, state_dict keys (5-D conv weights, InstanceNorm3d affine parameters) and the
forward signature / returned list of feature maps are the reference's."""
from torch import nn

from modules.movement_embedding import MovementEmbeddingModule
from mnk import ops


class DownBlock3D(nn.Module):
    """conv(1,k,k) without padding -> InstanceNorm (optional) -> LeakyReLU(0.2) -> avg-pool (1,2,2)
    (modules/discriminator.py:7-33)."""

    def __init__(self, in_features, out_features, norm=False, kernel_size=4):
        super(DownBlock3D, self).__init__()
        self.conv = nn.Conv3d(in_channels=in_features, out_channels=out_features,
                              kernel_size=(1, kernel_size, kernel_size))
        self.norm = nn.InstanceNorm3d(out_features, affine=True) if norm else None
        self.in_features, self.out_features, self.kernel_size = in_features, out_features, kernel_size

    def forward_act(self, x, c, leaf_input=False):
        k = self.kernel_size
        out = ops.ConvKxKFn.apply(x, self.conv.weight, self.conv.bias, c, k, k, 0, leaf_input)
        if self.norm is not None:
            out = ops.InstNormActFn.apply(out, self.norm.weight, self.norm.bias, self.out_features, 0.2, True,
                                          self.norm.eps)
        else:
            out = ops.InstNormActFn.apply(out, None, None, self.out_features, 0.2, True, 0.0)
        return out, self.out_features

    def forward(self, x):
        out, c = self.forward_act(ops.to_act(x), self.in_features)
        return ops.from_act(out, c, x.shape[0])


class Discriminator(nn.Module):
    """Pix2Pix-like discriminator on [frame | key-point heat-maps]; returns every intermediate feature map
    (modules/discriminator.py:36-79)."""

    def __init__(self, num_channels=3, num_kp=10, kp_variance=0.01, scale_factor=1,
                 block_expansion=64, num_blocks=4, max_features=512, kp_embedding_params=None):
        super(Discriminator, self).__init__()
        if kp_embedding_params is not None:
            self.kp_embedding = MovementEmbeddingModule(num_kp=num_kp, kp_variance=kp_variance,
                                                        num_channels=num_channels, **kp_embedding_params)
            embedding_channels = self.kp_embedding.out_channels
        else:
            self.kp_embedding = None
            embedding_channels = 0
        widths = [num_channels + embedding_channels] + [min(max_features, block_expansion * (2 ** (i + 1)))
                                                        for i in range(num_blocks)]
        self.down_blocks = nn.ModuleList([DownBlock3D(widths[i], widths[i + 1], norm=(i != 0), kernel_size=4)
                                          for i in range(num_blocks)])
        self.conv = nn.Conv3d(self.down_blocks[-1].conv.out_channels, out_channels=1, kernel_size=1)
        self.scale_factor = scale_factor
        self.num_channels = num_channels

    def forward_acts(self, x, kp_driving, kp_source, tap=None):
        """The same pass with the block outputs left in the kernels' NHWC form: ([(act, channels), ...] per down block,
        score (B,1,1,h,w)) -- for callers that reduce the feature maps on the device (mnk.engine, PairL1Fn).  x may hold
        twice as many samples as the key points ([generated | real] of the same videos): both halves get the same embedding.
        tap(i, act, channels) -> act: called on every block output; the next block continues on what it returns."""
        b, _, d = x.shape[:3]
        if d != 1:
            raise NotImplementedError("one frame per sample (train.py:43-44,69-70)")
        step = ops.step_from_scale(self.scale_factor)
        out, c = ops.to_act(x, step), self.num_channels
        if self.kp_embedding:
            bk = kp_driving['mean'].shape[0]
            if bk != b and (2 * bk != b or self.kp_embedding.use_deformed_source_image):
                raise ValueError("key points of %d samples for %d frames" % (bk, b))
            emb, ce = self.kp_embedding.forward_act(x[:bk], kp_driving, kp_source, pre_step=step)
            out, c = (ops.Concat2Fn if bk == b else ops.Concat2PairFn).apply(out, c, emb, ce), c + ce
        acts = []
        for i, down_block in enumerate(self.down_blocks):
            out, c = down_block.forward_act(out, c, leaf_input=(i == 0))
            if tap is not None:
                out = tap(i, out, c)
            acts.append((out, c))
        return acts, ops.Conv1x1SigmoidFn.apply(out, self.conv.weight, self.conv.bias, c, b, 0)

    def forward(self, x, kp_driving, kp_source):
        b, _, d = x.shape[:3]
        if d != 1:
            raise NotImplementedError("InstanceNorm3d statistics span the time axis; every caller of the reference "
                                      "passes one frame (train.py:43-44,69-70)")
        out_maps = [x]
        step = ops.step_from_scale(self.scale_factor)
        out, c = ops.to_act(x, step), self.num_channels
        if self.kp_embedding:
            emb, ce = self.kp_embedding.forward_act(x, kp_driving, kp_source, pre_step=step)
            out, c = ops.Concat2Fn.apply(out, c, emb, ce), c + ce
        for i, down_block in enumerate(self.down_blocks):
            out, c = down_block.forward_act(out, c, leaf_input=(i == 0))
            out_maps.append(ops.from_act(out, c, b))
        out_maps.append(ops.Conv1x1SigmoidFn.apply(out, self.conv.weight, self.conv.bias, c, b, 0))
        return out_maps
