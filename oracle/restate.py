"""CPU restatement of the reference's frame-generation hot path.  TEST INFRASTRUCTURE ONLY.

Plain PyTorch (CPU, fp32 or fp64) functional code over a ``state_dict`` that uses the reference's own
parameter names, so the same weights can be fed to the reference, to this file and to the HIP path.
Every function cites the reference lines it restates (paths relative to the reference root).  The time
axis D is folded into the batch (``temporal`` is never set, every Conv3d has a
(1,3,3) kernel), i.e. all convolutions are 2-D.

Pinned (see tests/test_oracle_golden.py and oracle/make_golden.py): every function here is checked
against the *real* reference imported from /root/reference by ``oracle/ref_shim.py``; the outputs of
that comparison are committed under ``tests/golden/`` so the check also runs where the reference tree
is absent.  The reference itself ships no tests or golden vectors for this path.

May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # sync_batchnorm/batchnorm.py:39
BN_MOMENTUM = 0.1    # sync_batchnorm/batchnorm.py:39


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def fold(x5):
    """(B,C,D,H,W) -> (B*D,C,H,W), frame index = b*D + d."""
    b, c, d, h, w = x5.shape
    return x5.permute(0, 2, 1, 3, 4).reshape(b * d, c, h, w)


def unfold(x4, b):
    """(B*D,C,H,W) -> (B,C,D,H,W)."""
    n, c, h, w = x4.shape
    return x4.reshape(b, n // b, c, h, w).permute(0, 2, 1, 3, 4)


def make_coordinate_grid(h, w, dtype=torch.float32):
    """modules/util.py:26-42.  (h,w,2); [...,0] = x = 2*(j/(w-1))-1, [...,1] = y."""
    x = torch.arange(w).to(dtype)
    y = torch.arange(h).to(dtype)
    x = 2 * (x / (w - 1)) - 1
    y = 2 * (y / (h - 1)) - 1
    return torch.stack([x.view(1, w).expand(h, w), y.view(h, 1).expand(h, w)], dim=2)


def smallest_singular(m):
    """modules/util.py:244-255 (closed form for 2x2, op order kept)."""
    a, b, c, d = m[..., 0, 0], m[..., 0, 1], m[..., 1, 0], m[..., 1, 1]
    s1 = a ** 2 + b ** 2 + c ** 2 + d ** 2
    s2 = (a ** 2 + b ** 2 - c ** 2 - d ** 2) ** 2
    s2 = torch.sqrt(s2 + 4 * (a * c + b * d) ** 2)
    return torch.sqrt((s1 - s2) / 2).unsqueeze(-1)


def matrix_inverse(m):
    """modules/util.py:206-224, eps == 0 branch: solve(m, I) (reference: torch.gesv LU)."""
    eye = torch.eye(m.shape[-1], dtype=m.dtype).expand_as(m)
    return torch.linalg.solve(m, eye)


def nearest_resize(x4, size):
    """F.interpolate(mode='nearest') to `size` on the two trailing dims (generator.py:55,72)."""
    return F.interpolate(x4, size=size, mode="nearest")


def nearest_scale(x4, scale):
    """F.interpolate(scale_factor=(1,s,s)) nearest (keypoint_detector.py:99, dense_motion_module.py:44,
    movement_embedding.py:44).  For s = 0.5/0.25 this is x[..., ::k, ::k]."""
    if scale == 1:
        return x4
    return F.interpolate(x4, scale_factor=(scale, scale), mode="nearest")


# ----------------------------------------------------------------------------------------------
# blocks (modules/util.py)
# ----------------------------------------------------------------------------------------------
class Ctx:
    """Carries the state dict + mode; records running-stat updates instead of mutating the dict."""

    def __init__(self, sd, training=True, sync_clamp=False):
        self.sd = sd
        self.training = training
        self.new_stats = {}
        # sync_batchnorm/batchnorm.py:125 (parallel path) uses clamp(var, eps); the CPU / single
        # device path (batchnorm.py:51-53) uses var + eps.  The build pins var + eps.
        self.sync_clamp = sync_clamp

    def p(self, name):
        return self.sd[name]


def conv(ctx, x, prefix, padding=1, groups=1):
    """nn.Conv3d with kernel (1,k,k) == conv2d on folded frames (modules/util.py:139-140)."""
    w = ctx.p(prefix + ".weight")
    b = ctx.sd.get(prefix + ".bias")
    return F.conv2d(x, w[:, :, 0], b, padding=padding, groups=groups)


def batch_norm(ctx, x, prefix):
    """sync_batchnorm/batchnorm.py:48-53 (non-parallel branch -> F.batch_norm) and :113-125 for the
    running-stat update rule (momentum 0.1, unbiased variance)."""
    w, b = ctx.p(prefix + ".weight"), ctx.p(prefix + ".bias")
    rm, rv = ctx.p(prefix + ".running_mean"), ctx.p(prefix + ".running_var")
    if not ctx.training:
        scale = w / torch.sqrt(rv + BN_EPS)
        return x * scale.view(1, -1, 1, 1) + (b - rm * scale).view(1, -1, 1, 1)
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    ctx.new_stats[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
    ctx.new_stats[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * n / (n - 1)
    inv_std = var.clamp(min=BN_EPS) ** -0.5 if ctx.sync_clamp else (var + BN_EPS) ** -0.5
    return (x - mean.view(1, -1, 1, 1)) * (inv_std * w).view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def down_block(ctx, x, prefix):
    """modules/util.py:103-108: conv -> BN -> ReLU -> avgpool(1,2,2)."""
    return F.avg_pool2d(F.relu(batch_norm(ctx, conv(ctx, x, prefix + ".conv"), prefix + ".norm")), 2)


def up_block(ctx, x, prefix):
    """modules/util.py:83-88: nearest x2 -> conv -> BN -> ReLU."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.relu(batch_norm(ctx, conv(ctx, x, prefix + ".conv"), prefix + ".norm"))


def same_block(ctx, x, prefix, groups):
    """modules/util.py:122-126 with kernel (1,1,1), padding 0 (dense_motion_module.py:26-27)."""
    return F.relu(batch_norm(ctx, conv(ctx, x, prefix + ".conv", padding=0, groups=groups), prefix + ".norm"))


def res_block(ctx, x, prefix):
    """modules/util.py:59-68: BN -> ReLU -> conv -> BN -> ReLU -> conv, += x."""
    out = conv(ctx, F.relu(batch_norm(ctx, x, prefix + ".norm1")), prefix + ".conv1")
    out = conv(ctx, F.relu(batch_norm(ctx, out, prefix + ".norm2")), prefix + ".conv2")
    return out + x


def encoder(ctx, x, prefix, num_blocks):
    """modules/util.py:147-152: returns [x, d1, ..., dn]."""
    outs = [x]
    for i in range(num_blocks):
        outs.append(down_block(ctx, outs[-1], "%s.down_blocks.%d" % (prefix, i)))
    return outs


def decoder(ctx, skips, prefix, num_blocks, use_last_conv=True):
    """modules/util.py:181-189: pop, up-block, cat([out, skip]); optional last conv."""
    skips = list(skips)
    out = skips.pop()
    for i in range(num_blocks):
        out = up_block(ctx, out, "%s.up_blocks.%d" % (prefix, i))
        out = torch.cat([out, skips.pop()], dim=1)
    if use_last_conv:
        return conv(ctx, out, prefix + ".conv")
    return out


def hourglass(ctx, x, prefix, num_blocks):
    """modules/util.py:202-203."""
    return decoder(ctx, encoder(ctx, x, prefix + ".encoder", num_blocks), prefix + ".decoder", num_blocks)


# ----------------------------------------------------------------------------------------------
# keypoints <-> gaussians (modules/keypoint_detector.py)
# ----------------------------------------------------------------------------------------------
def gaussian2kp(heatmap, kp_variance="matrix", clip_variance=None):
    """modules/keypoint_detector.py:43-78.  heatmap (B,K,D,H,W) already soft-maxed."""
    b, k, d, h, w = heatmap.shape
    hm = heatmap.unsqueeze(-1) + 1e-7
    grid = make_coordinate_grid(h, w, heatmap.dtype).view(1, 1, 1, h, w, 2)
    mean = (hm * grid).sum(dim=(3, 4))                       # (B,K,D,2)
    kp = {"mean": mean.permute(0, 2, 1, 3)}
    if kp_variance == "matrix":
        ms = grid - mean.view(b, k, d, 1, 1, 2)
        var = ms.unsqueeze(-1) * ms.unsqueeze(-2)            # (B,K,D,H,W,2,2)
        var = (var * hm.unsqueeze(-1)).sum(dim=(3, 4)).permute(0, 2, 1, 3, 4)
        if clip_variance:
            sg = smallest_singular(var).unsqueeze(-1)
            var = torch.max(torch.tensor(clip_variance, dtype=var.dtype), sg) * var / sg
        kp["var"] = var
    elif kp_variance == "single":
        ms = grid - mean.view(b, k, d, 1, 1, 2)
        var = ((ms ** 2) * hm).sum(dim=(3, 4)).mean(dim=-1, keepdim=True).unsqueeze(-1)
        kp["var"] = var.permute(0, 2, 1, 3, 4)
    return kp


def kp2gaussian(kp, spatial_size, kp_variance="matrix"):
    """modules/keypoint_detector.py:7-40.  kp['mean'] (...,2) -> (..., h, w)."""
    mean = kp["mean"]
    h, w = spatial_size
    lead = mean.shape[:-1]
    grid = make_coordinate_grid(h, w, mean.dtype).view((1,) * len(lead) + (h, w, 2))
    ms = grid - mean.reshape(lead + (1, 1, 2))
    if kp_variance == "matrix":
        inv = matrix_inverse(kp["var"]).reshape(lead + (1, 1, 2, 2))
        under = torch.matmul(torch.matmul(ms.unsqueeze(-2), inv), ms.unsqueeze(-1)).squeeze(-1).squeeze(-1)
        return torch.exp(-0.5 * under)
    if kp_variance == "single":
        return torch.exp(-0.5 * (ms ** 2).sum(-1) / kp["var"])
    return torch.exp(-0.5 * (ms ** 2).sum(-1) / kp_variance)


def kp_detector_forward(sd, params, x, training=True, ctx=None):
    """modules/keypoint_detector.py:97-109.  x (B,C,D,H,W) -> {'mean','var'}."""
    ctx = ctx or Ctx(sd, training)
    b = x.shape[0]
    x4 = nearest_scale(fold(x), params.get("scale_factor", 1))
    heat = hourglass(ctx, x4, "predictor", params["num_blocks"])        # (B*D,K,h,w)
    n, k, h, w = heat.shape
    heat = F.softmax(heat.reshape(n, k, h * w) / params["temperature"], dim=2).reshape(n, k, h, w)
    return gaussian2kp(unfold(heat, b), params["kp_variance"], params.get("clip_variance"))


# ----------------------------------------------------------------------------------------------
# movement embedding / dense motion / generator
# ----------------------------------------------------------------------------------------------
def grid_sample_ac(inp, grid):
    """F.grid_sample bilinear, zeros padding, align_corners=True (torch 0.4.1 default)."""
    return F.grid_sample(inp, grid, mode="bilinear", padding_mode="zeros", align_corners=True)


def movement_embedding(p, source_image, kp_driving, kp_source):
    """modules/movement_embedding.py:42-92.  `p` = dict(num_kp, kp_variance, num_channels,
    use_deformed_source_image, use_difference, use_heatmap, add_bg_feature_map, heatmap_type,
    norm_const, scale_factor).  Returns (B, C_emb, d, h, w)."""
    use_heatmap = p.get("use_heatmap", True)
    use_diff = p.get("use_difference", False)
    use_def = p.get("use_deformed_source_image", False)
    add_bg = p.get("add_bg_feature_map", False)
    norm_const = p.get("norm_const", "sum")
    kpv = p["kp_variance"]
    b = source_image.shape[0]
    src4 = nearest_scale(fold(source_image), p.get("scale_factor", 1))
    h, w = src4.shape[2:]
    _, d, num_kp, _ = kp_driving["mean"].shape

    def normalize(hm):                                      # movement_embedding.py:33-40
        if norm_const == "sum":
            return hm / hm.sum(dim=(3, 4), keepdim=True)
        return hm / norm_const

    inputs = []
    if use_heatmap:
        hm = normalize(kp2gaussian(kp_driving, (h, w), kpv))             # (B,d,K,h,w)
        if p.get("heatmap_type", "gaussian") == "difference":
            hm = hm - normalize(kp2gaussian(kp_source, (h, w), kpv))
        if add_bg:
            hm = torch.cat([torch.zeros(b, d, 1, h, w, dtype=hm.dtype), hm], dim=2)
        inputs.append(hm.unsqueeze(3))
    slots = num_kp + int(add_bg)
    if use_diff or use_def:
        diff = kp_source["mean"] - kp_driving["mean"]                    # (B,d,K,2)
        if add_bg:
            diff = torch.cat([torch.zeros(b, d, 1, 2, dtype=diff.dtype), diff], dim=2)
        diff_maps = diff.view(b, d, slots, 2, 1, 1).expand(b, d, slots, 2, h, w)
    if use_diff:
        inputs.append(diff_maps)
    if use_def:
        c = src4.shape[1]
        rep = src4.view(b, 1, 1, c, h, w).expand(b, d, slots, c, h, w).reshape(b * d * slots, c, h, w)
        grid = make_coordinate_grid(h, w, src4.dtype).view(1, h, w, 2) + \
            diff_maps.reshape(b * d * slots, 2, h, w).permute(0, 2, 3, 1)
        inputs.append(grid_sample_ac(rep, grid).view(b, d, slots, c, h, w))
    enc = torch.cat(inputs, dim=3)                                       # (B,d,slots,per,h,w)
    return enc.reshape(b, d, -1, h, w).permute(0, 2, 1, 3, 4)


def dense_motion_forward(ctx, prefix, p, common, source_image, kp_driving, kp_source):
    """modules/dense_motion_module.py:42-76.  Returns the sampling field (B,d,h,w,3)."""
    num_kp = common["num_kp"]
    emb_p = dict(p["mask_embedding_params"], num_kp=num_kp, kp_variance=common["kp_variance"],
                 num_channels=common["num_channels"], add_bg_feature_map=True, scale_factor=1)
    sf = p.get("scale_factor", 1)
    if sf != 1:
        source_image = unfold(nearest_scale(fold(source_image), sf), source_image.shape[0])
    b = source_image.shape[0]
    pred5 = movement_embedding(emb_p, source_image, kp_driving, kp_source)
    d = pred5.shape[2]
    pred = fold(pred5)
    for i in range(p.get("num_group_blocks", 0)):
        pred = same_block(ctx, pred, "%s.group_blocks.%d" % (prefix, i), groups=num_kp + 1)
        pred = F.leaky_relu(pred, 0.2)
    pred = hourglass(ctx, pred, prefix + ".hourglass", p["num_blocks"])  # (B*d, K+1+2, h, w)
    n, _, h, w = pred.shape
    rel = 0
    if p["use_mask"]:
        mask = F.softmax(pred[:, :num_kp + 1], dim=1)                    # (n,K+1,h,w)
        diff = kp_source["mean"] - kp_driving["mean"]                    # (B,d,K,2)
        diff = torch.cat([torch.zeros(b, d, 1, 2, dtype=diff.dtype), diff], dim=2).reshape(n, num_kp + 1, 2)
        rel = (diff.view(n, num_kp + 1, 2, 1, 1) * mask.unsqueeze(2)).sum(dim=1)   # (n,2,h,w)
    if p["use_correction"]:
        rel = rel + pred[:, -2:]
    field = rel.permute(0, 2, 3, 1) + make_coordinate_grid(h, w, pred.dtype).view(1, h, w, 2)
    field = torch.cat([field, torch.zeros(n, h, w, 1, dtype=field.dtype)], dim=-1)
    return field.view(b, d, h, w, 3)


def resize_field(field, size, mode):
    """generator.py:53-56: field (B,d,ho,wo,3) -> (B,d,h,w,3).  d stays, so 'trilinear' is bilinear
    (align_corners=False) in h,w and 'nearest' is an index pick."""
    b, d, ho, wo, _ = field.shape
    f = field.permute(0, 4, 1, 2, 3)
    f = F.interpolate(f, size=(d,) + tuple(size), mode=mode)
    return f.permute(0, 2, 3, 4, 1)


def deform_input(inp, field, mode="nearest"):
    """generator.py:51-58.  inp (B,C,1,h,w), field (B,d,ho,wo,3) -> (B,C,d,h,w).  The 5-D grid_sample
    with input depth 1 and z == 0 is a 2-D bilinear sample."""
    b, c, _, h, w = inp.shape
    d = field.shape[1]
    f = resize_field(field, (h, w), mode)[..., :2]                       # (B,d,h,w,2)
    rep = inp[:, :, 0].unsqueeze(1).expand(b, d, c, h, w).reshape(b * d, c, h, w)
    out = grid_sample_ac(rep, f.reshape(b * d, h, w, 2))
    return out.view(b, d, c, h, w).permute(0, 2, 1, 3, 4)


def generator_forward(sd, gp, common, source_image, kp_driving, kp_source, training=True, ctx=None):
    """modules/generator.py:60-82.  gp = generator_params, common = common_params."""
    ctx = ctx or Ctx(sd, training)
    b = source_image.shape[0]
    mode = gp.get("interpolation_mode", "nearest")
    nb = gp["num_blocks"]
    skips = [unfold(s, b) for s in encoder(ctx, fold(source_image), "appearance_encoder", nb)]
    if gp.get("dense_motion_params") is not None:
        field = dense_motion_forward(ctx, "dense_motion_module", gp["dense_motion_params"], common,
                                     source_image, kp_driving, kp_source)
    else:                                                                # dense_motion_module.py:79-87
        h, w = source_image.shape[3:]
        d = kp_driving["mean"].shape[1]
        g = make_coordinate_grid(h, w, source_image.dtype).view(1, 1, h, w, 2).expand(b, d, h, w, 2)
        field = torch.cat([g, torch.zeros(b, d, h, w, 1, dtype=g.dtype)], dim=-1)
    deformed = [deform_input(s, field, mode) for s in skips]
    if gp.get("kp_embedding_params") is not None:
        d = kp_driving["mean"].shape[1]
        emb_p = dict(gp["kp_embedding_params"], num_kp=common["num_kp"], kp_variance=common["kp_variance"],
                     num_channels=common["num_channels"])
        emb = movement_embedding(emb_p, source_image, kp_driving, kp_source)
        kp_skips = [F.interpolate(emb, size=(d,) + tuple(s.shape[3:]), mode=mode) for s in skips]
        deformed = [torch.cat([a, k], dim=1) for a, k in zip(deformed, kp_skips)]
    video_deformed = deform_input(source_image, field, mode)
    out = decoder(ctx, [fold(s) for s in deformed], "video_decoder", nb, use_last_conv=False)
    for i in range(gp["num_refinement_blocks"]):
        out = res_block(ctx, out, "refinement_module.r%d" % i)
    out = conv(ctx, out, "refinement_module.conv-last", padding=0)
    return {"video_prediction": unfold(torch.sigmoid(out), b), "video_deformed": video_deformed,
            "_field": field}


# ----------------------------------------------------------------------------------------------
# discriminator + losses + one training iteration (callers of the hot path; "next" row f-1 of
# SURVEY.md section 8) -- restated so that step-level losses can be compared.
# ----------------------------------------------------------------------------------------------
def discriminator_forward(sd, dp, common, x, kp_driving, kp_source):
    """modules/discriminator.py:63-79 (+ DownBlock3D :26-33).  Returns the list of feature maps (5-D)."""
    b = x.shape[0]
    out_maps = [x]
    sf = dp.get("scale_factor", 1)
    x4 = nearest_scale(fold(x), sf)
    if dp.get("kp_embedding_params") is not None:
        emb_p = dict(dp["kp_embedding_params"], num_kp=common["num_kp"], kp_variance=common["kp_variance"],
                     num_channels=common["num_channels"])
        heat = movement_embedding(emb_p, unfold(x4, b), kp_driving, kp_source)
        x4 = torch.cat([x4, fold(heat)], dim=1)
    out = x4
    for i in range(dp.get("num_blocks", 4)):
        pre = "down_blocks.%d" % i
        out = F.conv2d(out, sd[pre + ".conv.weight"][:, :, 0], sd[pre + ".conv.bias"])
        if i != 0:
            out = F.instance_norm(out, weight=sd[pre + ".norm.weight"], bias=sd[pre + ".norm.bias"], eps=1e-5)
        out = F.avg_pool2d(F.leaky_relu(out, 0.2), 2)
        out_maps.append(unfold(out, b))
    out = F.conv2d(out, sd["conv.weight"][:, :, 0], sd["conv.bias"])
    out_maps.append(unfold(out, b))
    return out_maps


def mean_batch(v):
    return v.reshape(v.shape[0], -1).mean(-1)                            # modules/losses.py:4-5


def generator_losses(maps_gen, maps_real, video_deformed, lw):
    """modules/losses.py:46-60 -> list of per-sample loss vectors."""
    vals = []
    if lw["reconstruction_deformed"] != 0:
        vals.append(lw["reconstruction_deformed"] * mean_batch(torch.abs(maps_real[0] - video_deformed)))
    if lw["reconstruction"] != 0:
        for i, (a, bm) in enumerate(zip(maps_real[:-1], maps_gen[:-1])):
            if lw["reconstruction"][i] == 0:
                continue
            vals.append(lw["reconstruction"][i] * mean_batch(torch.abs(bm - a)))
    vals.append(lw["generator_gan"] * mean_batch((1 - maps_gen[-1]) ** 2))
    return vals


def discriminator_losses(maps_gen, maps_real, lw):
    """modules/losses.py:19-23,63-67."""
    return [lw["discriminator_gan"] * mean_batch((1 - maps_real[-1]) ** 2 + maps_gen[-1] ** 2)]


def split_kp(kp, detach=False):
    """train.py:14-21."""
    f = (lambda t: t.detach()) if detach else (lambda t: t)
    return ({k: f(v[:, 1:]) for k, v in kp.items()}, {k: f(v[:, :1]) for k, v in kp.items()})


def generator_full_forward(sds, cfg, source, video, training=True):
    """train.py:36-53 (GeneratorFullModel.forward).  sds = dict(generator=, discriminator=, kp_detector=)
    state dicts.  Returns (loss list, generated dict, kp_joined, ctx_gen, ctx_kp)."""
    mp, tp = cfg["model_params"], cfg["train_params"]
    common = mp["common_params"]
    ctx_kp = Ctx(sds["kp_detector"], training)
    ctx_g = Ctx(sds["generator"], training)
    kp_joined = kp_detector_forward(sds["kp_detector"], dict(mp["kp_detector_params"], **common),
                                    torch.cat([source, video], dim=2), ctx=ctx_kp)
    drv, src = split_kp(kp_joined, tp["detach_kp_generator"])
    gen = generator_forward(sds["generator"], mp["generator_params"], common, source, drv, src, ctx=ctx_g)
    drv2, src2 = split_kp(kp_joined, False)
    maps_gen = discriminator_forward(sds["discriminator"], mp["discriminator_params"], common,
                                     gen["video_prediction"], drv2, src2)
    maps_real = discriminator_forward(sds["discriminator"], mp["discriminator_params"], common, video, drv2, src2)
    losses = generator_losses(maps_gen, maps_real, gen["video_deformed"], tp["loss_weights"])
    return losses, gen, kp_joined, ctx_g, ctx_kp


def discriminator_full_forward(sds, cfg, video, kp_joined, generated):
    """train.py:68-75 (DiscriminatorFullModel.forward)."""
    mp, tp = cfg["model_params"], cfg["train_params"]
    common = mp["common_params"]
    drv, src = split_kp(kp_joined, tp["detach_kp_discriminator"])
    maps_gen = discriminator_forward(sds["discriminator"], mp["discriminator_params"], common,
                                     generated["video_prediction"].detach(), drv, src)
    maps_real = discriminator_forward(sds["discriminator"], mp["discriminator_params"], common, video, drv, src)
    return discriminator_losses(maps_gen, maps_real, tp["loss_weights"])


# ----------------------------------------------------------------------------------------------
# algorithmic work (denominator of the roofline)
# ----------------------------------------------------------------------------------------------
def conv_flops_hot_path(cfg, height, width, frames_kp=2):
    """Forward conv FLOPs (2*MAC) of KPDetector (on `frames_kp` frames) + generator (1 frame) for ONE
    training pair, derived from the channel ladders of modules/util.py:142-143,169-171.
    Returns dict(kp=, gen=, total=, layers=[(name, cin, cout, h, w, k, flops)])."""
    mp = cfg["model_params"]
    common = mp["common_params"]
    layers = []

    def add(name, cin, cout, h, w, k=3, frames=1, groups=1):
        layers.append((name, cin, cout, h, w, k, 2 * (cin // groups) * cout * k * k * h * w * frames))

    def hourglass_layers(name, be, cin, cout, nb, mx, h, w, frames, last=True, extra=0, enc=True, cin_dec=None):
        chans = [cin]
        hh, ww = h, w
        if enc:
            for i in range(nb):
                co = min(mx, be * 2 ** (i + 1))
                add("%s.enc%d" % (name, i), chans[-1], co, hh, ww, frames=frames)
                chans.append(co)
                hh, ww = hh // 2, ww // 2
        else:
            for i in range(nb):
                chans.append(min(mx, be * 2 ** (i + 1)))
                hh, ww = hh // 2, ww // 2
        for j, i in enumerate(range(nb)[::-1]):
            ci = (1 if i == nb - 1 else 2) * min(mx, be * 2 ** (i + 1)) + extra
            co = min(mx, be * 2 ** i)
            hh, ww = hh * 2, ww * 2
            add("%s.dec%d" % (name, j), ci, co, hh, ww, frames=frames)
        if last:
            add(name + ".last", be + cin + extra, cout, h, w, frames=frames)

    kpp = mp["kp_detector_params"]
    s = kpp.get("scale_factor", 1)
    hourglass_layers("kp", kpp["block_expansion"], common["num_channels"], common["num_kp"], kpp["num_blocks"],
                     kpp["max_features"], int(height * s), int(width * s), frames_kp)
    n_kp = len(layers)
    gp = mp["generator_params"]
    be, mx, nb = gp["block_expansion"], gp["max_features"], gp["num_blocks"]
    cin = common["num_channels"]
    hh, ww, c = height, width, cin
    for i in range(nb):
        co = min(mx, be * 2 ** (i + 1))
        add("gen.app%d" % i, c, co, hh, ww)
        c, hh, ww = co, hh // 2, ww // 2
    dm = gp.get("dense_motion_params")
    K = common["num_kp"]
    if dm is not None:
        s = dm.get("scale_factor", 1)
        me = dm["mask_embedding_params"]
        per = int(me.get("use_heatmap", True)) + 2 * int(me.get("use_difference", False)) + \
            cin * int(me.get("use_deformed_source_image", False))
        cemb = per * (K + 1)
        for i in range(dm.get("num_group_blocks", 0)):
            add("gen.dm.group%d" % i, cemb, cemb, int(height * s), int(width * s), k=1, groups=K + 1)
        hourglass_layers("gen.dm", dm["block_expansion"], cemb, (K + 1) * dm["use_mask"] + 2 * dm["use_correction"],
                         dm["num_blocks"], dm["max_features"], int(height * s), int(width * s), 1)
    kpe = gp.get("kp_embedding_params")
    extra = 0
    if kpe is not None:
        extra = (int(kpe.get("use_heatmap", True)) + 2 * int(kpe.get("use_difference", False)) +
                 cin * int(kpe.get("use_deformed_source_image", False))) * (K + int(kpe.get("add_bg_feature_map", False)))
    hourglass_layers("gen.dec", be, cin, cin, nb, mx, height, width, 1, last=False, extra=extra, enc=False)
    cref = be + cin + extra
    for i in range(gp["num_refinement_blocks"]):
        add("gen.ref%d.conv1" % i, cref, cref, height, width)
        add("gen.ref%d.conv2" % i, cref, cref, height, width)
    add("gen.conv-last", cref, cin, height, width, k=1)
    kp = sum(l[-1] for l in layers[:n_kp])
    gen = sum(l[-1] for l in layers[n_kp:])
    return {"kp": kp, "gen": gen, "total": kp + gen, "layers": layers}


def to_dtype(obj, dtype):
    if isinstance(obj, dict):
        return {k: to_dtype(v, dtype) for k, v in obj.items()}
    if torch.is_tensor(obj) and obj.is_floating_point():
        return obj.to(dtype)
    return obj


def _selfcheck():  # pragma: no cover
    g = make_coordinate_grid(4, 5)
    assert g.shape == (4, 5, 2) and g[0, 0, 0] == -1 and g[3, 4, 1] == 1
    assert math.isclose(float(g[0, 1, 0]), -0.5)


if __name__ == "__main__":  # pragma: no cover
    _selfcheck()
    print("ok")
