"""Patch discriminator of the training step (modules/discriminator.py; SURVEY.md section 8f-1, the first "next" row).

`Discriminator` / `DownBlock3D` are the gfx950-kernel classes of `mnk/discriminator_hip.py` (4x4 no-pad implicit-GEMM
convolutions, fused InstanceNorm + LeakyReLU + avg-pool, 1x1 score head): with the two discriminator calls of a pass
batched into one (mnk.engine.discriminate_pair) they are the faster choice on the MI355X (15.41 vs 15.62 ms per
moving-gif iteration, profiles/README.md).  `StockDiscriminator` below is the same network on stock PyTorch-ROCm ops
(MIOpen); MNK_NATIVE_DISC=0 selects it.  Both keep the reference's constructor, state_dict keys (5-D conv weights)
and forward signature, so checkpoints and train.py interoperate; the (1,4,4) convolutions are evaluated as 2-D
convolutions on the folded frames and the key-point heat-maps come from the HIP embedding kernel either way."""
import torch
from torch import nn
import torch.nn.functional as F

from modules.movement_embedding import MovementEmbeddingModule
from mnk import knobs


class StockDownBlock3D(nn.Module):
    """conv(1,k,k) without padding -> InstanceNorm (optional) -> LeakyReLU(0.2) -> avg-pool (1,2,2)
    (modules/discriminator.py:7-33)."""

    def __init__(self, in_features, out_features, norm=False, kernel_size=4):
        super(StockDownBlock3D, self).__init__()
        self.conv = nn.Conv3d(in_channels=in_features, out_channels=out_features,
                              kernel_size=(1, kernel_size, kernel_size))
        self.norm = nn.InstanceNorm3d(out_features, affine=True) if norm else None

    def forward(self, x):
        b, c, d, h, w = x.shape
        y = F.conv2d(x.transpose(1, 2).reshape(b * d, c, h, w), self.conv.weight[:, :, 0], self.conv.bias)
        if self.norm is not None:
            y = F.instance_norm(y, weight=self.norm.weight, bias=self.norm.bias, eps=self.norm.eps)
        y = F.avg_pool2d(F.leaky_relu(y, 0.2), 2)
        return y.reshape(b, d, y.shape[1], y.shape[2], y.shape[3]).transpose(1, 2)


class StockDiscriminator(nn.Module):
    """Pix2Pix-like discriminator on [frame | key-point heat-maps]; returns every intermediate feature map
    (modules/discriminator.py:36-79)."""

    def __init__(self, num_channels=3, num_kp=10, kp_variance=0.01, scale_factor=1,
                 block_expansion=64, num_blocks=4, max_features=512, kp_embedding_params=None):
        super(StockDiscriminator, self).__init__()
        if kp_embedding_params is not None:
            self.kp_embedding = MovementEmbeddingModule(num_kp=num_kp, kp_variance=kp_variance,
                                                        num_channels=num_channels, **kp_embedding_params)
            embedding_channels = self.kp_embedding.out_channels
        else:
            self.kp_embedding = None
            embedding_channels = 0
        widths = [num_channels + embedding_channels] + [min(max_features, block_expansion * (2 ** (i + 1)))
                                                        for i in range(num_blocks)]
        self.down_blocks = nn.ModuleList([StockDownBlock3D(widths[i], widths[i + 1], norm=(i != 0), kernel_size=4)
                                          for i in range(num_blocks)])
        self.conv = nn.Conv3d(self.down_blocks[-1].conv.out_channels, out_channels=1, kernel_size=1)
        self.scale_factor = scale_factor

    def forward(self, x, kp_driving, kp_source):
        out_maps = [x]
        if self.scale_factor != 1:
            x = F.interpolate(x, scale_factor=(1, self.scale_factor, self.scale_factor))
        out = x
        if self.kp_embedding:
            out = torch.cat([x, self.kp_embedding(x, kp_driving, kp_source)], dim=1)
        for down_block in self.down_blocks:
            out = down_block(out)
            out_maps.append(out)
        b, c, d, h, w = out.shape
        score = F.conv2d(out.transpose(1, 2).reshape(b * d, c, h, w), self.conv.weight[:, :, 0], self.conv.bias)
        out_maps.append(score.reshape(b, d, 1, h, w).transpose(1, 2))
        return out_maps


from mnk.discriminator_hip import Discriminator as HipDiscriminator, DownBlock3D as HipDownBlock3D  # noqa: E402

if knobs.on("MNK_NATIVE_DISC"):
    Discriminator, DownBlock3D = HipDiscriminator, HipDownBlock3D
else:
    Discriminator, DownBlock3D = StockDiscriminator, StockDownBlock3D
