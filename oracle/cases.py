"""Shared definitions of the parity cases (configs, seeded inputs, weight perturbation).
TEST INFRASTRUCTURE ONLY -- used by oracle/make_golden.py (with the real reference), by tests/ and by
bench.py's cpu_baseline leg.  Nothing here touches /root/reference.
"""
import copy

import torch

# A deliberately small configuration that still exercises every feature of the hot path:
# group blocks, use_difference, scale_factor 0.5 sub-networks, clip_variance, kp embedding.
TINY = {
    "model_params": {
        "common_params": {"num_kp": 4, "kp_variance": "matrix", "num_channels": 3},
        "kp_detector_params": {"temperature": 0.1, "block_expansion": 8, "max_features": 32, "num_blocks": 3,
                               "scale_factor": 0.5, "clip_variance": 0.001},
        "generator_params": {
            "block_expansion": 8, "max_features": 32, "num_blocks": 3, "num_refinement_blocks": 2,
            "dense_motion_params": {
                "block_expansion": 8, "max_features": 32, "num_blocks": 3, "use_mask": True,
                "use_correction": True, "scale_factor": 0.5,
                "mask_embedding_params": {"use_heatmap": True, "use_deformed_source_image": True,
                                          "use_difference": True, "heatmap_type": "difference",
                                          "norm_const": 100},
                "num_group_blocks": 2},
            "kp_embedding_params": {"scale_factor": 0.5, "use_heatmap": True, "norm_const": 100,
                                    "heatmap_type": "difference"}},
        "discriminator_params": {"kp_embedding_params": {"norm_const": 100}, "block_expansion": 8,
                                 "max_features": 32, "num_blocks": 3},
    },
    "train_params": {"detach_kp_generator": False, "detach_kp_discriminator": True, "lr": 2.0e-4,
                     "loss_weights": {"reconstruction": [10, 10, 10, 1], "reconstruction_deformed": 1,
                                      "generator_gan": 1, "discriminator_gan": 1}},
}

# Same topology as TINY but full resolution sub-networks, 'gaussian' heatmaps and no group blocks
TINY2 = copy.deepcopy(TINY)
TINY2["model_params"]["kp_detector_params"].update(scale_factor=1, clip_variance=None)
TINY2["model_params"]["kp_detector_params"].pop("clip_variance")
_dm = TINY2["model_params"]["generator_params"]["dense_motion_params"]
_dm.update(scale_factor=1, num_group_blocks=0)
_dm["mask_embedding_params"] = {"use_heatmap": True, "use_deformed_source_image": True,
                                "heatmap_type": "gaussian", "norm_const": 10}
TINY2["model_params"]["generator_params"]["kp_embedding_params"] = {"use_heatmap": True, "norm_const": 10,
                                                                    "heatmap_type": "gaussian"}


def synthetic_pair(batch, height, width, seed=1234, channels=3):
    """BASELINE.md section 2 protocol: source, video ~ U[0,1) float32 (B,3,1,H,W) from a seeded CPU
    generator."""
    g = torch.Generator().manual_seed(seed)
    source = torch.rand(batch, channels, 1, height, width, generator=g)
    video = torch.rand(batch, channels, 1, height, width, generator=g)
    return source, video


def smooth_pair(batch, height, width, seed=1234, channels=3):
    """Low-frequency images (sum of a few sinusoids), closer to real frames than white noise: keypoint
    heatmaps become peaked and the warps are exercised on smooth content."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.linspace(0, 1, height).view(1, 1, 1, height, 1)
    xs = torch.linspace(0, 1, width).view(1, 1, 1, 1, width)
    out = []
    for _ in range(2):
        img = torch.zeros(batch, channels, 1, height, width)
        for _ in range(4):
            fx = torch.rand(batch, channels, 1, 1, 1, generator=g) * 6
            fy = torch.rand(batch, channels, 1, 1, 1, generator=g) * 6
            ph = torch.rand(batch, channels, 1, 1, 1, generator=g) * 6.28
            img = img + torch.sin(fx * xs * 6.28 + fy * ys * 6.28 + ph)
        out.append((img / 8 + 0.5).clamp(0, 1))
    return out[0], out[1]


def perturb_state_dict(sd, seed, scale=0.02):
    """Deterministic in-place perturbation so that zero-initialised heads (dense_motion_module.py:33-35),
    unit BN weights and fresh running statistics are all exercised.  Keys are visited in sorted order."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd.keys()):
        v = sd[k]
        if not torch.is_tensor(v) or not v.is_floating_point():
            continue
        noise = torch.randn(v.shape, generator=g)
        with torch.no_grad():
            if k.endswith("running_var"):
                v.mul_(1 + 0.5 * noise.abs())
            elif k.endswith("running_mean"):
                v.add_(0.1 * noise)
            elif ".norm" in k:
                v.add_(0.1 * noise)
            else:
                v.add_(scale * noise)
    return sd


def random_kp(batch, d, num_kp, seed, spread=0.6, with_var=True):
    g = torch.Generator().manual_seed(seed)
    mean = (torch.rand(batch, d, num_kp, 2, generator=g) * 2 - 1) * spread
    kp = {"mean": mean}
    if with_var:
        a = torch.randn(batch, d, num_kp, 2, 2, generator=g) * 0.05
        var = torch.matmul(a, a.transpose(-1, -2)) + 0.01 * torch.eye(2)
        kp["var"] = var
    return kp


def is_noise_bias(key):
    """Conv biases that sit directly in front of a BatchNorm (DownBlock3D/UpBlock3D/SameBlock3D conv,
    ResBlock3D conv1), and the keypoint detector's last conv bias (a per-channel shift in front of a
    spatial softmax): their gradient is analytically zero, so the reference's value is rounding noise
    and is excluded from gradient comparisons."""
    if not key.endswith(".bias"):
        return False
    if key == "predictor.decoder.conv.bias":
        return True
    return any(s in key for s in ("down_blocks.", "up_blocks.", "group_blocks.")) and key.endswith(".conv.bias") \
        or key.endswith(".conv1.bias")
