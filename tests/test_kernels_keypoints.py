"""softmax->key-point and key-point->embedding kernels against the oracle restatement (fp64)."""
import pytest
import torch
import torch.nn.functional as F

from _util import to_nhwc, from_nhwc, ceil4, relerr, maxerr
from oracle import restate, cases


# 8x12 / 16x16 / 9x7: the one-pass kernel with 4 pixels per lane; 32x32: 16; 64x64 and 40x50: 64; 72x64: the multi-pass kernels
@pytest.mark.parametrize("shape", [(2, 4, 8, 12), (3, 10, 16, 16), (1, 16, 9, 7), (2, 10, 32, 32), (1, 3, 64, 64), (2, 5, 40, 50),
                                   (1, 2, 72, 64)])
@pytest.mark.parametrize("temperature", [0.1, 1.0])
def test_softmax_kp(be, shape, temperature):
    n, k, h, w = shape
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(n, k, h, w, generator=g) * (2.0 if temperature < 1 else 4.0)
    ld = ceil4(k)
    # reference in fp64 through the restated gaussian2kp (modules/keypoint_detector.py:43-78,103-107)
    ld64 = logits.double().requires_grad_(True)
    heat = F.softmax(ld64.view(n, k, -1) / temperature, dim=2).view(n, k, 1, h, w)
    kp = restate.gaussian2kp(heat, "matrix", None)           # (n,1,k,2), (n,1,k,2,2)
    dmean = torch.randn(n, k, 2, generator=g).double()
    dvar = torch.randn(n, k, 2, 2, generator=g).double() * 3
    ((kp["mean"][:, 0] * dmean).sum() + (kp["var"][:, 0] * dvar).sum()).backward()

    X = be.t(to_nhwc(logits, pad_value=float("nan")))
    mean, var, stat = be.empty(n, k, 2), be.empty(n, k, 4), be.empty(n, k, 2)
    be.call("mnk_softmax_kp_fwd", X, ld, n, h, w, k, temperature, mean, var, stat)
    be.sync()
    assert maxerr(mean.cpu(), kp["mean"][:, 0]) < 2e-6
    assert maxerr(var.cpu().view(n, k, 2, 2), kp["var"][:, 0]) < 2e-6
    D = be.empty(n, h, w, ld)
    be.call("mnk_softmax_kp_bwd", X, ld, n, h, w, k, temperature, mean, stat, be.t(dmean.float()),
            be.t(dvar.float().reshape(n, k, 4)), D, ld)
    be.sync()
    assert relerr(from_nhwc(D.cpu(), k), ld64.grad) < 2e-5
    assert torch.all(D.cpu()[..., k:] == 0)


VARIANTS = {
    "mask": dict(use_heatmap=True, use_deformed_source_image=True, heatmap_type="difference", norm_const=100,
                 add_bg_feature_map=True),
    "mask_diff": dict(use_heatmap=True, use_deformed_source_image=True, use_difference=True,
                      heatmap_type="difference", norm_const=100, add_bg_feature_map=True),
    "kpemb": dict(use_heatmap=True, norm_const=100, heatmap_type="difference"),
    "sum": dict(use_heatmap=True, use_deformed_source_image=True, heatmap_type="gaussian", norm_const="sum",
                add_bg_feature_map=True),
    "sum_diff": dict(use_heatmap=True, heatmap_type="difference", norm_const="sum"),
    "gauss10": dict(use_heatmap=True, norm_const=10, heatmap_type="gaussian"),
    "diffonly": dict(use_heatmap=False, use_difference=True, add_bg_feature_map=True),
}


def run_embedding(be, p, src, kpd, kps, dout=None):
    """Drive mnk_movement_embedding_{fwd,bwd}; src (B,C,1,h,w) already at the embedding resolution."""
    b, c, _, h, w = src.shape
    _, d, K, _ = kpd["mean"].shape
    add_bg = int(p.get("add_bg_feature_map", False))
    uh, ud, us = int(p.get("use_heatmap", True)), int(p.get("use_difference", False)), \
        int(p.get("use_deformed_source_image", False))
    per = uh + 2 * ud + c * us
    slots = K + add_bg
    ld_out = ceil4(slots * per)
    img = be.t(to_nhwc(src[:, :, 0]))
    md, vd = be.t(kpd["mean"].reshape(b * d, K, 2)), be.t(kpd["var"].reshape(b * d, K, 4))
    ms, vs = be.t(kps["mean"].reshape(b, K, 2)), be.t(kps["var"].reshape(b, K, 4))
    norm = p.get("norm_const", "sum")
    nd = ns = None
    norm_c = float(norm) if norm != "sum" else 0.0
    if norm == "sum" and uh:
        nd, ns = be.empty(b * d * K), be.empty(b * K)
        be.call("mnk_gaussian_sums", md, vd, 0.0, b * d * K, h, w, nd)
        be.call("mnk_gaussian_sums", ms, vs, 0.0, b * K, h, w, ns)
    out = be.empty(b * d, h, w, ld_out)
    hd = int(p.get("heatmap_type", "gaussian") == "difference")
    args = (img, img.shape[-1], c, md, vd, ms, vs, 0.0, b, d, h, w, K, add_bg, uh, ud, us, hd, norm_c, nd, ns)
    be.call("mnk_movement_embedding_fwd", *args, out, ld_out)
    grads = None
    if dout is not None:
        DO = be.t(dout)
        g = [be.empty(b * d, K, 2), be.empty(b * d, K, 4), be.empty(b * d, K, 2), be.empty(b * d, K, 4)]
        be.call("mnk_movement_embedding_bwd", *args, DO, ld_out, *g)
        grads = [t.cpu() for t in g]
    be.sync()
    return out.cpu(), slots * per, grads


@pytest.mark.parametrize("name", list(VARIANTS))
@pytest.mark.parametrize("d", [1, 2])
def test_movement_embedding(be, name, d):
    p = dict(VARIANTS[name], num_kp=4, kp_variance="matrix", num_channels=3)
    g = torch.Generator().manual_seed(7)
    b, h, w, K = 2, 10, 12, 4
    src = torch.rand(b, 3, 1, h, w, generator=g)
    kpd, kps = cases.random_kp(b, d, K, seed=6), cases.random_kp(b, 1, K, seed=8)
    # fp64 reference with autograd
    kd64 = {k: v.double().requires_grad_(True) for k, v in kpd.items()}
    ks64 = {k: v.double().requires_grad_(True) for k, v in kps.items()}
    ref = restate.movement_embedding(p, src.double(), kd64, ks64)           # (B,C,d,h,w)
    cemb = ref.shape[1]
    dref = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * dref).sum().backward()
    ld_out = ceil4(cemb)
    dout = torch.zeros(b * d, h, w, ld_out)
    dout[..., :cemb] = dref.permute(0, 2, 3, 4, 1).reshape(b * d, h, w, cemb).float()
    out, c_used, grads = run_embedding(be, p, src, kpd, kps, dout)
    assert c_used == cemb
    ref_nhwc = ref.permute(0, 2, 3, 4, 1).reshape(b * d, h, w, cemb)
    assert maxerr(out[..., :cemb], ref_nhwc) < 3e-6
    assert torch.all(out[..., cemb:] == 0)
    dmd, dvd, dms, dvs = grads
    tol = 5e-5

    def gr(t):
        return t.grad if t.grad is not None else torch.zeros_like(t)

    if p.get("use_heatmap", True) or p.get("use_difference") or p.get("use_deformed_source_image"):
        assert relerr(dmd.view(b, d, K, 2), gr(kd64["mean"])) < tol
        assert maxerr(dms.view(b, d, K, 2).sum(1, keepdim=True), gr(ks64["mean"])) < tol * (1 + float(gr(ks64["mean"]).abs().max()))
    if p.get("use_heatmap", True):
        assert relerr(dvd.view(b, d, K, 2, 2), kd64["var"].grad) < tol
        if p.get("heatmap_type") == "difference":
            assert relerr(dvs.view(b, d, K, 2, 2).sum(1, keepdim=True), ks64["var"].grad) < tol


def test_embedding_matches_reference_golden(be):
    """The committed outputs of the REAL reference module (tests/golden/functions.pt)."""
    import os
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "functions.pt"), weights_only=False)
    src, kpd, kps = gold["emb_src"], gold["emb_kpd"], gold["emb_kps"]
    for tag, kw in gold["emb_variants"].items():
        sf = kw.get("scale_factor", 1)
        s = src[..., ::int(1 / sf), ::int(1 / sf)] if sf != 1 else src
        p = dict(kw, num_kp=4, kp_variance="matrix", num_channels=3)
        out, cemb, _ = run_embedding(be, p, s, kpd, kps)
        ref = gold["emb_" + tag]                                            # (B,C,d,h,w)
        b, c, d, h, w = ref.shape
        assert c == cemb
        assert maxerr(out[..., :c], ref.permute(0, 2, 3, 4, 1).reshape(b * d, h, w, c)) < 3e-6, tag


def test_clip_variance(be):
    """mnk_kp_clip_variance_fwd/bwd == var * max(clip, sigma_min) / sigma_min with the reference's closed-form sigma_min
    (keypoint_detector.py:62-65, modules/util.py:244-255), forward and gradient, clipped and unclipped matrices."""
    g = torch.Generator().manual_seed(4)
    m = 37
    a = torch.randn(m, 2, 2, generator=g) * 0.1
    var = a @ a.transpose(1, 2) + 0.002 * torch.eye(2)          # SPD, sigma_min around the clip value
    clip = 0.004
    vd = var.double().requires_grad_(True)
    sg = restate.smallest_singular(vd).unsqueeze(-1)
    ref = torch.max(torch.full((), clip, dtype=torch.float64), sg) * vd / sg
    dout = torch.randn(m, 2, 2, generator=g, dtype=torch.float64)
    ref.backward(dout)
    clipped = (sg.detach().flatten() < clip)
    assert 3 < int(clipped.sum()) < m - 3                        # both branches are exercised
    # the closed form cancels (s1 - s2) in fp32: the yard-stick is the same formula evaluated by torch in fp32
    v32 = var.clone().requires_grad_(True)
    sg32 = restate.smallest_singular(v32).unsqueeze(-1)
    ref32 = torch.max(torch.full((), clip), sg32) * v32 / sg32
    ref32.backward(dout.float())
    spread_f, spread_b = relerr(ref32.detach(), ref.detach()), relerr(v32.grad, vd.grad)
    V, O, DV = be.t(var), be.empty(m, 2, 2), be.empty(m, 2, 2)
    be.call("mnk_kp_clip_variance_fwd", V, clip, m, O, 0)
    be.call("mnk_kp_clip_variance_bwd", V, clip, m, be.t(dout.float()), DV, 0)
    be.sync()
    assert relerr(O.cpu(), ref.detach()) < 4 * spread_f + 1e-6
    assert relerr(DV.cpu(), vd.grad) < 4 * spread_b + 1e-5


def test_clip_variance_reference_mode(be):
    """reference_mode = 1 (ops.ClipVarianceFn mode="reference", MNK_CLIP_VARIANCE_MODE=reference): sigma_min by the reference's own
    fp32 closed form sqrt((s1 - s2) / 2) in its operation order (modules/util.py:244-255) -- equal to torch's fp32 evaluation
    of the restated formula to rounding (forward and autograd's backward), and unusable on a nearly singular covariance
    exactly where the reference's is."""
    from mnk import ops
    g = torch.Generator().manual_seed(4)
    m = 37
    a = torch.randn(m, 2, 2, generator=g) * 0.1
    var = a @ a.transpose(1, 2) + 0.002 * torch.eye(2)
    clip = 0.004
    v32 = var.clone().requires_grad_(True)
    sg32 = restate.smallest_singular(v32).unsqueeze(-1)
    ref32 = torch.max(torch.full((), clip), sg32) * v32 / sg32
    dout = torch.randn(m, 2, 2, generator=g)
    ref32.backward(dout)
    V = be.t(var).requires_grad_(True)
    out = ops.ClipVarianceFn.apply(V, clip, "reference")
    out.backward(be.t(dout))
    be.sync()
    # same formula, same precision: only the association of a few products differs (fma contraction)
    assert relerr(out.detach().cpu(), ref32.detach()) < 2e-5
    assert relerr(V.grad.cpu(), v32.grad) < 2e-4
    stable = ops.ClipVarianceFn.apply(be.t(var), clip, "stable")
    assert relerr(stable.cpu(), out.detach().cpu()) < 1e-3            # two evaluations of one number on benign matrices
    # a line-shaped covariance: sigma_min / sigma_max = 1e-6 -- the reference's form has no digits left
    bad = torch.tensor([[[0.5, 0.5], [0.5, 0.5 + 1e-6]]])
    ref_sg = restate.smallest_singular(bad)
    got = ops.ClipVarianceFn.apply(be.t(bad), 0.001, "reference").cpu()
    ok = ops.ClipVarianceFn.apply(be.t(bad), 0.001, "stable").cpu()
    assert torch.isfinite(ok).all()
    if not bool(torch.isfinite(ref_sg).all()) or float(ref_sg) <= 0:
        assert not bool(torch.isfinite(got).all())                    # the reference's failure is reproduced, not repaired
    with pytest.raises(ValueError):
        ops.ClipVarianceFn.apply(be.t(var), clip, "exact")


def test_clip_variance_of_nearly_singular_covariances(be):
    """Heat-maps stretched along a line give covariances with sigma_min ~ 1e-6 at entries ~ 0.4 (measured in training: det
    2e-6, off-diagonal 0.45).  The reference's closed form takes sigma_min^2 from s1 - s2, which is far below the rounding of
    s1 there: evaluated in fp32 it is rounding noise (zero or negative for a good part of such matrices -> inf / NaN after the
    clip).  The kernels take sigma_min = |det| / sigma_max: finite, and equal to the fp64 evaluation of the reference's formula."""
    g = torch.Generator().manual_seed(12)
    m = 64
    ang = torch.rand(m, generator=g, dtype=torch.float64) * 3.14159
    r = torch.stack([torch.stack([ang.cos(), -ang.sin()], -1), torch.stack([ang.sin(), ang.cos()], -1)], -2)
    big = 0.2 + 0.7 * torch.rand(m, generator=g, dtype=torch.float64)
    small = 10.0 ** (-7.0 + 3.0 * torch.rand(m, generator=g, dtype=torch.float64))        # 1e-7 .. 1e-4
    var = (r @ torch.diag_embed(torch.stack([big, small], -1)) @ r.transpose(1, 2)).float()     # what the kernel is given
    clip = 0.001
    vd = var.double().requires_grad_(True)
    sg = restate.smallest_singular(vd).unsqueeze(-1)
    ref = torch.max(torch.full((), clip, dtype=torch.float64), sg) * vd / sg
    dout = torch.randn(m, 2, 2, generator=g, dtype=torch.float64)
    ref.backward(dout)
    # the fp32 evaluation of the reference's formula on the same matrices: how often is it unusable?
    sg32 = restate.smallest_singular(var)
    broken = int((~torch.isfinite(sg32) | (sg32 <= 0)).sum())
    V, O, DV = be.t(var), be.empty(m, 2, 2), be.empty(m, 2, 2)
    be.call("mnk_kp_clip_variance_fwd", V, clip, m, O, 0)
    be.call("mnk_kp_clip_variance_bwd", V, clip, m, be.t(dout.float()), DV, 0)
    be.sync()
    assert torch.isfinite(O.cpu()).all() and torch.isfinite(DV.cpu()).all()
    # fp32 inputs carry ~6e-8 absolute rounding: sigma_min = det / sigma_max is known to ~1e-7 / sigma_min relative
    per = ((O.cpu().double() - ref.detach()).abs().amax(dim=(1, 2)) / ref.detach().abs().amax(dim=(1, 2)))
    assert float(per.max()) < 1e-3, float(per.max())
    gper = ((DV.cpu().double() - vd.grad).abs().amax(dim=(1, 2)) / vd.grad.abs().amax(dim=(1, 2)))
    assert float(gper.max()) < 5e-3, float(gper.max())
    print("fp32 evaluation of the reference's closed form: %d of %d matrices give a zero / non-finite sigma_min" % (broken, m))
