"""MovementEmbeddingModule (modules/movement_embedding.py:8-92): parameter-free per-key-point feature maps
[heat-map | (dx,dy) | source translated by kp_source - kp_driving] x (background +) K slots, rendered by ONE kernel
straight into the NHWC buffer the next convolution reads (no repeat / grid_sample / cat intermediates)."""
from torch import nn

from modules.keypoint_detector import _split_variance
from mnk import ops


class MovementEmbeddingModule(nn.Module):
    def __init__(self, num_kp, kp_variance, num_channels, use_deformed_source_image=False, use_difference=False,
                 use_heatmap=True, add_bg_feature_map=False, heatmap_type='gaussian', norm_const='sum', scale_factor=1):
        super(MovementEmbeddingModule, self).__init__()
        assert heatmap_type in ['gaussian', 'difference']
        assert ((int(use_heatmap) + int(use_deformed_source_image) + int(use_difference)) >= 1)
        self.out_channels = (1 * use_heatmap + 2 * use_difference + num_channels * use_deformed_source_image) * (
            num_kp + add_bg_feature_map)
        self.num_kp = num_kp
        self.num_channels = num_channels
        self.kp_variance = kp_variance
        self.heatmap_type = heatmap_type
        self.use_difference = use_difference
        self.use_deformed_source_image = use_deformed_source_image
        self.use_heatmap = use_heatmap
        self.add_bg_feature_map = add_bg_feature_map
        self.norm_const = norm_const
        self.scale_factor = scale_factor

    def forward_act(self, source_image, kp_driving, kp_source, src_act=None, pre_step=1):
        """-> (act (B*d, h, w, ld), channels).  `src_act`: the source image already as an act at this module's
        resolution (saves a conversion); only read when use_deformed_source_image.  `pre_step`: an additional
        nearest down-scaling applied by the caller's own scale_factor (DenseMotionModule)."""
        b = source_image.shape[0]
        step = ops.step_from_scale(self.scale_factor) * pre_step
        h, w = source_image.shape[3] // step, source_image.shape[4] // step
        d, k = kp_driving['mean'].shape[1:3]
        img = None
        if self.use_deformed_source_image:
            img = src_act if src_act is not None else ops.to_act(source_image.detach(), step)
        var_d, const_var = _split_variance(kp_driving, self.kp_variance) if self.use_heatmap else (None, 1.0)
        var_s, _ = _split_variance(kp_source, self.kp_variance) if self.use_heatmap else (None, 1.0)
        cfg = (b, d, h, w, k, self.num_channels, bool(self.add_bg_feature_map), bool(self.use_heatmap),
               bool(self.use_difference), bool(self.use_deformed_source_image), self.heatmap_type == 'difference',
               self.norm_const, const_var)
        out = ops.MovementEmbeddingFn.apply(img, kp_driving['mean'], var_d, kp_source['mean'], var_s, cfg)
        return out, self.out_channels

    def forward(self, source_image, kp_driving, kp_source):
        out, c = self.forward_act(source_image, kp_driving, kp_source)
        return ops.from_act(out, c, source_image.shape[0])
