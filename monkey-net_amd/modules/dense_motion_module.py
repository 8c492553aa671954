"""DenseMotionModule / IdentityDeformation (modules/dense_motion_module.py): key-point displacements + source
appearance -> dense sampling field.  Embedding, grouped 1x1 blocks, hourglass and the mask-softmax head all run on
the gfx950 kernels; the field is kept as (N,h,w,2) internally and only expanded to the reference's (B,d,h,w,3)
(z == 0) at the public boundary."""
import torch
from torch import nn

from modules.util import Hourglass, SameBlock3D, make_coordinate_grid
from modules.movement_embedding import MovementEmbeddingModule
from mnk import ops


class DenseMotionModule(nn.Module):
    def __init__(self, block_expansion, num_blocks, max_features, mask_embedding_params, num_kp,
                 num_channels, kp_variance, use_correction, use_mask, bg_init=2, num_group_blocks=0, scale_factor=1):
        super(DenseMotionModule, self).__init__()
        self.mask_embedding = MovementEmbeddingModule(num_kp=num_kp, kp_variance=kp_variance, num_channels=num_channels,
                                                      add_bg_feature_map=True, **mask_embedding_params)
        # kept for attribute parity; the (dx,dy) maps it would render are constants folded into the head kernel
        self.difference_embedding = MovementEmbeddingModule(num_kp=num_kp, kp_variance=kp_variance,
                                                            num_channels=num_channels,
                                                            add_bg_feature_map=True, use_difference=True,
                                                            use_heatmap=False, use_deformed_source_image=False)
        emb_c = self.mask_embedding.out_channels
        self.group_blocks = nn.ModuleList([SameBlock3D(emb_c, emb_c, groups=num_kp + 1, kernel_size=(1, 1, 1),
                                                       padding=(0, 0, 0)) for _ in range(num_group_blocks)])
        self.hourglass = Hourglass(block_expansion=block_expansion, in_features=emb_c,
                                   out_features=(num_kp + 1) * use_mask + 2 * use_correction,
                                   max_features=max_features, num_blocks=num_blocks)
        # identity field at initialisation: zero weights, background logit = bg_init (dense_motion_module.py:33-35)
        head = self.hourglass.decoder.conv
        with torch.no_grad():
            head.weight.zero_()
            head.bias.copy_(torch.tensor(([bg_init] + [0] * num_kp) * use_mask + [0, 0] * use_correction,
                                         dtype=torch.float))
        self.num_kp = num_kp
        self.use_correction = use_correction
        self.use_mask = use_mask
        self.scale_factor = scale_factor

    def field_act(self, source_image, kp_driving, kp_source):
        """-> sampling field (B*d, h, w, 2) in [-1,1] coordinates (x, y)."""
        step = ops.step_from_scale(self.scale_factor)
        b = source_image.shape[0]
        # MovementEmbeddingModule applies its own scale_factor (1 here) on top of this module's down-scaling
        pred, c = self.mask_embedding.forward_act(source_image, kp_driving, kp_source, pre_step=step)
        for block in self.group_blocks:
            pred, c = block.forward_act(pred, c)     # the reference's extra leaky_relu(0.2) after ReLU is the identity
        pred, c = self.hourglass.forward_act(pred, c)
        delta = None
        if self.use_mask and kp_driving['mean'].shape[1] == 1 and kp_source['mean'].shape == kp_driving['mean'].shape:
            # one driving frame per video: the key-point difference is formed inside the kernels
            return ops.MotionFieldKPFn.apply(pred, kp_source['mean'], kp_driving['mean'], self.num_kp, bool(self.use_correction))
        if self.use_mask:
            diff = kp_source['mean'] - kp_driving['mean']                       # (B,d,K,2)
            d = diff.shape[1]
            delta = torch.cat([torch.zeros_like(diff[:, :, :1]), diff], dim=2).reshape(b * d, self.num_kp + 1, 2)
        return ops.MotionFieldFn.apply(pred, delta, self.num_kp, bool(self.use_mask), bool(self.use_correction))

    def forward(self, source_image, kp_driving, kp_source):
        b = source_image.shape[0]
        field = self.field_act(source_image, kp_driving, kp_source)
        n, h, w, _ = field.shape
        field = torch.cat([field, torch.zeros_like(field[..., :1])], dim=-1)
        return field.view(b, n // b, h, w, 3)


class IdentityDeformation(nn.Module):
    """Identity sampling field (modules/dense_motion_module.py:79-87)."""

    def field_act(self, source_image, kp_driving, kp_source):
        b, _, _, h, w = source_image.shape
        d = kp_driving['mean'].shape[1]
        grid = make_coordinate_grid((h, w), type=source_image.type()).to(source_image.device)
        return grid.view(1, h, w, 2).repeat(b * d, 1, 1, 1).contiguous()

    def forward(self, appearance_frame, kp_video, kp_appearance):
        b = appearance_frame.shape[0]
        field = self.field_act(appearance_frame, kp_video, kp_appearance)
        n, h, w, _ = field.shape
        field = torch.cat([field, torch.zeros_like(field[..., :1])], dim=-1)
        return field.view(b, n // b, h, w, 3)
