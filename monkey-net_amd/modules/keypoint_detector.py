"""KPDetector and the heat-map <-> key-point transforms (modules/keypoint_detector.py) on the gfx950 kernels."""
import torch
from torch import nn

from modules.util import Hourglass, smallest_singular
from mnk import knobs, ops


def _split_variance(kp, kp_variance):
    """-> (var tensor (...,2,2) or None, constant variance or 0.0) in the form the embedding kernel takes."""
    if kp_variance == 'matrix':
        return kp['var'], 0.0
    if kp_variance == 'single':
        v = kp['var']                                   # (...,1,1): isotropic -> diag(v, v)
        eye = torch.eye(2, dtype=v.dtype, device=v.device)
        return v * eye, 0.0
    return None, float(kp_variance)


def kp2gaussian(kp, spatial_size, kp_variance='matrix'):
    """exp(-0.5 (g-mu)^T Sigma^-1 (g-mu)) on the [-1,1] grid (modules/keypoint_detector.py:7-40).
    kp['mean'] (..., K, 2) -> (..., K, h, w).  Public helper: the generator uses the fused embedding kernel."""
    mean = kp['mean']
    lead = mean.shape[:-1]
    h, w = spatial_size
    var, const_var = _split_variance(kp, kp_variance)
    m = mean.reshape(-1, 1, 1, 2)
    v = var.reshape(-1, 1, 1, 2, 2) if var is not None else None
    n = m.shape[0]
    cfg = (n, 1, h, w, 1, 0, False, True, False, False, False, 1.0, const_var)
    out = ops.MovementEmbeddingFn.apply(None, m, v, m, v, cfg)          # (n,h,w,4), channel 0 = heat-map
    return out[..., 0].reshape(lead + (h, w))


def gaussian2kp(heatmap, kp_variance='matrix', clip_variance=None, clip_variance_mode=None):
    """Mean / covariance of a normalised heat-map (B,K,D,H,W) (modules/keypoint_detector.py:43-78).
    Public helper; KPDetector.forward fuses the soft-max into the same kernel instead of calling this.
    clip_variance_mode (not in the reference): "stable" (default) | "reference", see _finish_kp."""
    b, k, d, h, w = heatmap.shape
    act = ops.to_act(torch.log(heatmap))                               # softmax(log p) == p for a normalised p
    mean, var = ops.SoftmaxKPFn.apply(act, k, 1.0)
    return _finish_kp(mean.view(b, d, k, 2), var.view(b, d, k, 2, 2), kp_variance, clip_variance, clip_variance_mode)


def _finish_kp(mean, var, kp_variance, clip_variance, clip_variance_mode=None):
    """clip_variance_mode: how sigma_min of the covariance is evaluated (keypoint_detector.py:62-65, util.py:244-255):
    "stable" (default, MNK_CLIP_VARIANCE_MODE) -- |det| / sigma_max, the reference's value in exact arithmetic and the fp64
    reference's value in fp32; "reference" -- the reference's own fp32 closed form sqrt((s1 - s2) / 2), which cancels to 0 / NaN
    once sigma_min / sigma_max <~ 2e-4 (line-shaped heat-maps reach that): for bit-level parity work against the reference."""
    kp = {'mean': mean}
    if kp_variance == 'matrix':
        if clip_variance:
            mode = clip_variance_mode or knobs.get("MNK_CLIP_VARIANCE_MODE")
            var = ops.ClipVarianceFn.apply(var, clip_variance, mode)    # var * max(clip, sigma_min) / sigma_min (:62-65)
        kp['var'] = var
    elif kp_variance == 'single':
        kp['var'] = ((var[..., 0, 0] + var[..., 1, 1]) / 2).unsqueeze(-1).unsqueeze(-1)
    return kp


class KPDetector(nn.Module):
    """Hourglass -> K heat-maps -> spatial soft-max (temperature) -> soft-argmax mean + covariance
    (modules/keypoint_detector.py:81-109).  x (B,C,D,H,W) -> {'mean': (B,D,K,2), 'var': (B,D,K,2,2)}."""

    def __init__(self, block_expansion, num_kp, num_channels, max_features, num_blocks, temperature,
                 kp_variance, scale_factor=1, clip_variance=None):
        super(KPDetector, self).__init__()
        self.predictor = Hourglass(block_expansion, in_features=num_channels, out_features=num_kp,
                                   max_features=max_features, num_blocks=num_blocks)
        self.temperature = temperature
        self.kp_variance = kp_variance
        self.scale_factor = scale_factor
        self.clip_variance = clip_variance
        self.num_kp = num_kp
        self.num_channels = num_channels
        self.clip_variance_mode = None      # None: MNK_CLIP_VARIANCE_MODE ("stable"); "reference": the reference's fp32 sigma_min
        self._last_heat = None

    def forward(self, x):
        b, _, d = x.shape[:3]
        act = ops.to_act(x, ops.step_from_scale(self.scale_factor))
        heat, k = self.predictor.forward_act(act, self.num_channels)
        mean, var = ops.SoftmaxKPFn.apply(heat, k, self.temperature)
        self._last_heat = (heat.detach(), k, b, d)          # for keypoint_indices(): no copy, no extra launch
        return _finish_kp(mean.view(b, d, k, 2), var.view(b, d, k, 2, 2), self.kp_variance, self.clip_variance,
                          self.clip_variance_mode)

    def keypoint_indices(self, kp, frame_size=None):
        """The integer key-point positions of the LAST forward call (not part of the reference's API; the north star's
        "bit-exact keypoint indices"): {'pixel': (B,D,K,2) int32 = floor(size * (mean + 1) / 2), the pixel the reference's
        Visualizer draws the key point at (logger.py:99-100), for frames of `frame_size` = (W, H) (default: the heat-map's
        own size); 'argmax': (B,D,K) int32, h * W + w of the largest heat-map value (keypoint_detector.py:103-104)}."""
        heat, k, b, d = self._last_heat[:4]
        n, h, w, _ = heat.shape
        size = (w, h) if frame_size is None else frame_size
        am = ops.heatmap_argmax(heat, k)
        if len(self._last_heat) > 4 and self._last_heat[4]:
            # mnk.engine.joined_kp ran the detector on [sources | drivings] stacked along the batch axis and returned the key
            # points as (B, 2, K, .): the same re-layout for the arg-max of that call
            am = am.view(d, b, k).transpose(0, 1)
        else:
            am = am.view(b, d, k)
        return {'pixel': ops.kp_pixel_index(kp['mean'], size), 'argmax': am}
