"""BatchNorm statistics / apply / backward kernels and the layout kernels against torch CPU ops."""
import pytest
import torch
import torch.nn.functional as F

from _util import to_nhwc, from_nhwc, ceil4, relerr, maxerr


@pytest.mark.parametrize("shape", [(2, 5, 6, 4), (3, 45, 4, 6), (2, 300, 2, 2), (1, 64, 16, 16)])
@pytest.mark.parametrize("pool", [0, 1])
def test_bn_train_forward_backward(be, shape, pool):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    # reference (fp64)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    z = F.relu(F.batch_norm(xd, rm, rv, gd, bd, True, 0.1, 1e-5))
    if pool:
        z = F.avg_pool2d(z, 2)
    dz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(dz)

    ld = ceil4(c)
    X = be.t(to_nhwc(x))
    rows = n * h * w
    nws = be.query("mnk_bn_workspace_floats", rows, ld)
    ws = be.empty(nws)
    sums = be.empty(2 * c)
    be.call("mnk_bn_stats", X, ld, rows, c, sums, ws, nws)
    mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
    RM, RV, G, Bt = be.t(rm0.clone()), be.t(rv0.clone()), be.t(gamma), be.t(beta)
    be.call("mnk_bn_finalize", sums, float(rows), G, RM, RV, 0.1, 1e-5, c, 1, mean, invstd, scale)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    ldz = ld + 8
    Z = be.zeros(n, ho, wo, ldz)
    be.call("mnk_bn_act_fwd", X, ld, mean, scale, Bt, Z, ldz, 4, n, h, w, c, 1, pool)
    be.sync()
    assert maxerr(Z.cpu()[..., 4:4 + c].permute(0, 3, 1, 2), z) < 2e-5
    assert torch.all(Z.cpu()[..., :4] == 0) and torch.all(Z.cpu()[..., 4 + c:] == 0)
    assert maxerr(RM.cpu(), rm) < 1e-5 and maxerr(RV.cpu(), rv) < 1e-4
    # backward
    DZ = be.zeros(n, ho, wo, ldz)
    DZ[..., 4:4 + c] = be.t(dz.float().permute(0, 2, 3, 1))
    bs = be.empty(2 * c)
    be.call("mnk_bn_act_bwd_stats", X, ld, DZ, ldz, 4, mean, invstd, scale, Bt, n, h, w, c, 1, pool, bs, ws, nws)
    DY = be.empty(n, h, w, ld)
    be.call("mnk_bn_act_bwd_apply", X, ld, DZ, ldz, 4, mean, invstd, scale, Bt, bs, float(rows), 1, DY, ld, n, h, w, c,
            1, pool)
    be.sync()
    assert relerr(bs.cpu()[:c], bd.grad) < 1e-4
    assert relerr(bs.cpu()[c:], gd.grad) < 1e-4
    assert relerr(from_nhwc(DY.cpu(), c), xd.grad) < 1e-4
    assert torch.all(DY.cpu()[..., c:] == 0)
    # fused forms: second stage + finalisation in one launch; dy pass that also returns the column sums of dy
    m2, i2, s2, sums2 = be.empty(c), be.empty(c), be.empty(c), be.empty(2 * c)
    RM2, RV2 = be.t(rm0.clone()), be.t(rv0.clone())
    be.call("mnk_bn_stats_finalize", X, ld, rows, c, None, 0, float(rows), G, RM2, RV2, 0.1, 1e-5, 1, sums2, m2, i2, s2,
            ws, nws)
    DY2, dys = be.empty(n, h, w, ld), be.empty(c)
    be.call("mnk_bn_act_bwd_apply_colsum", X, ld, DZ, ldz, 4, mean, invstd, scale, Bt, bs, float(rows), 1, DY2, ld, n, h,
            w, c, 1, pool, dys, ws, nws)
    be.sync()
    for a, b in ((m2, mean), (i2, invstd), (s2, scale), (sums2, sums), (RM2, RM), (RV2, RV), (DY2, DY)):
        assert torch.equal(a.cpu(), b.cpu())
    ref_sum = DY.cpu().double().reshape(-1, ld).sum(0)[:c]
    assert maxerr(dys.cpu(), ref_sum) <= 1e-5 * float(DY.cpu().abs().double().reshape(-1, ld).sum(0).max()) + 1e-6


@pytest.mark.parametrize("row_blocks", [1, 63, 257, 700, 2048])
def test_second_stage_over_many_row_block_partials(be, row_blocks):
    """The second stage alone (mnk_bn_stats_finish, and fused with the finalisation: mnk_bn_stats_finalize on partials a conv
    epilogue left) over as many row blocks as the 64 x 64-tile launches of the 64^2 layers leave: fp64-exact column sums, and
    the two forms agree to the bit."""
    c, ld = 13, 16
    g = torch.Generator().manual_seed(8)
    part = torch.randn(row_blocks, 2, ld, generator=g) * 3
    part[:, 1] = part[:, 1].abs() * 40 + 50            # sums of squares: large against the sums, so that variances are positive
    ref = part.double().sum(0)[:, :c]
    count = 64.0 * row_blocks
    P = be.t(part)
    sums = be.empty(2 * c)
    be.call("mnk_bn_stats_finish", P, row_blocks, ld, c, sums)
    gamma = torch.rand(c, generator=g) + 0.5
    mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
    RM, RV = be.zeros(c), be.zeros(c) + 1
    be.call("mnk_bn_finalize", sums, count, be.t(gamma), RM, RV, 0.1, 1e-5, c, 1, mean, invstd, scale)
    m2, i2, s2, sums2 = be.empty(c), be.empty(c), be.empty(c), be.empty(2 * c)
    RM2, RV2 = be.zeros(c), be.zeros(c) + 1
    be.call("mnk_bn_stats_finalize", None, ld, 0, c, P, row_blocks, count, be.t(gamma), RM2, RV2, 0.1, 1e-5, 1, sums2, m2, i2,
            s2, None, 0)
    be.sync()
    assert torch.equal(sums.cpu(), ref.float().reshape(-1))          # fp64 accumulation, one rounding
    for a, b in ((m2, mean), (i2, invstd), (s2, scale), (sums2, sums), (RM2, RM), (RV2, RV)):
        assert torch.equal(a.cpu(), b.cpu())
    m = ref[0] / count
    assert maxerr(mean.cpu(), m) < 1e-6 * (1 + float(m.abs().max()))
    assert relerr(invstd.cpu(), 1 / torch.sqrt(ref[1] / count - m * m + 1e-5)) < 1e-5


def test_bn_eval(be):
    n, c, h, w = 2, 13, 4, 4
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, c, h, w, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.2
    ref = F.relu(F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-5))
    ld = ceil4(c)
    X = be.t(to_nhwc(x))
    mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
    be.call("mnk_bn_eval_coeffs", be.t(gamma), be.t(rm), be.t(rv), 1e-5, c, mean, invstd, scale)
    Z = be.zeros(n, h, w, ld)
    be.call("mnk_bn_act_fwd", X, ld, mean, scale, be.t(beta), Z, ld, 0, n, h, w, c, 1, 0)
    # eval-mode backward is a per-channel scale
    dz = torch.randn(n, c, h, w, generator=g)
    DZ = be.t(to_nhwc(dz))
    DY = be.empty(n, h, w, ld)
    be.call("mnk_bn_act_bwd_apply", X, ld, DZ, ld, 0, mean, invstd, scale, be.t(beta), None, 1.0, 0, DY, ld, n, h, w, c,
            1, 0)
    be.sync()
    assert maxerr(from_nhwc(Z.cpu(), c), ref) < 1e-5
    xr = x.clone().requires_grad_(True)
    F.relu(F.batch_norm(xr, rm, rv, gamma, beta, False, 0.1, 1e-5)).backward(dz)
    assert maxerr(from_nhwc(DY.cpu(), c), xr.grad) < 1e-5


def test_layout_roundtrip_and_scale(be):
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 2, 8, 12, generator=g)
    for step in (1, 2, 4):
        ho, wo = 8 // step, 12 // step
        out = be.empty(4, ho, wo, 4)
        be.call("mnk_ncdhw_to_nhwc", be.t(x), out, 2, 3, 2, 8, 12, step, 4)
        be.sync()
        ref = x[:, :, :, ::step, ::step].permute(0, 2, 3, 4, 1).reshape(4, ho, wo, 3)
        assert torch.equal(out.cpu()[..., :3], ref) and torch.all(out.cpu()[..., 3] == 0)
        if step > 1:
            sf = 1.0 / step
            ref2 = F.interpolate(x, scale_factor=(1, sf, sf))
            assert torch.equal(ref2, x[:, :, :, ::step, ::step])
    nh = be.empty(4, 8, 12, 4)
    be.call("mnk_ncdhw_to_nhwc", be.t(x), nh, 2, 3, 2, 8, 12, 1, 4)
    back = be.empty(2, 3, 2, 8, 12)
    be.call("mnk_nhwc_to_ncdhw", nh, 4, back, 2, 3, 2, 8, 12)
    be.sync()
    assert torch.equal(back.cpu(), x)


def test_copy_channels_and_resize(be):
    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, 7, 4, 4, generator=g)
    A = be.t(to_nhwc(a))
    D = be.zeros(2, 4, 4, 16)
    be.call("mnk_copy_channels", A, 8, 2, D, 16, 5, 4, 2 * 4 * 4, 0)
    be.call("mnk_copy_channels", A, 8, 2, D, 16, 5, 4, 2 * 4 * 4, 1)
    be.sync()
    assert torch.allclose(D.cpu()[..., 5:9], 2 * to_nhwc(a)[..., 2:6])
    for (hs, ws, hd, wd) in ((8, 8, 4, 4), (4, 4, 8, 8), (8, 8, 1, 1), (6, 6, 6, 6), (8, 8, 2, 2)):
        s = torch.randn(2, 3, hs, ws, generator=g)
        S = be.t(to_nhwc(s))
        O = be.zeros(2, hd, wd, 8)
        be.call("mnk_resize_nearest", S, 4, hs, ws, O, 8, 2, hd, wd, 2, 3)
        ref = F.interpolate(s, size=(hd, wd), mode="nearest")
        dd = torch.randn(2, 3, hd, wd, generator=g)
        DD = be.zeros(2, hd, wd, 8)
        DD[..., 2:5] = be.t(dd.permute(0, 2, 3, 1))
        DS = be.empty(2, hs, ws, 4)
        be.call("mnk_resize_nearest_bwd", DD, 8, 2, hd, wd, DS, 4, hs, ws, 2, 3)
        be.sync()
        assert torch.equal(O.cpu()[..., 2:5].permute(0, 3, 1, 2), ref)
        sr = s.clone().requires_grad_(True)
        F.interpolate(sr, size=(hd, wd), mode="nearest").backward(dd)
        assert maxerr(DS.cpu()[..., :3].permute(0, 3, 1, 2), sr.grad) < 1e-6
        be.call("mnk_resize_nearest_bwd_accumulate", DD, 8, 2, hd, wd, DS, 4, hs, ws, 2, 3)     # added to what is there
        be.sync()
        assert maxerr(DS.cpu()[..., :3].permute(0, 3, 1, 2), 2 * sr.grad) < 2e-6


def test_resize_bilinear(be):
    """interpolation_mode='trilinear' resize of the key-point embedding (generator.py:72) == F.interpolate bilinear,
    align_corners=False, forward and adjoint."""
    g = torch.Generator().manual_seed(5)
    for (hs, ws, hd, wd) in ((16, 16, 4, 4), (8, 8, 32, 32), (16, 16, 1, 1), (6, 6, 6, 6)):
        s = torch.randn(2, 3, hs, ws, generator=g)
        S = be.t(to_nhwc(s))
        O = be.zeros(2, hd, wd, 8)
        be.call("mnk_resize_bilinear", S, 4, hs, ws, O, 8, 2, hd, wd, 2, 3)
        sr = s.double().requires_grad_(True)
        ref = F.interpolate(sr, size=(hd, wd), mode="bilinear", align_corners=False)
        dd = torch.randn(2, 3, hd, wd, generator=g)
        ref.backward(dd.double())
        DD = be.zeros(2, hd, wd, 8)
        DD[..., 2:5] = be.t(dd.permute(0, 2, 3, 1))
        DS = be.zeros(2, hs, ws, 4)
        be.call("mnk_resize_bilinear_bwd", DD, 8, 2, hd, wd, DS, 4, hs, ws, 2, 3)
        be.sync()
        assert maxerr(O.cpu()[..., 2:5].permute(0, 3, 1, 2), ref) < 2e-6
        assert maxerr(DS.cpu()[..., :3].permute(0, 3, 1, 2), sr.grad) < 1e-5


@pytest.mark.parametrize("shape", [(3, 12, 13, 13), (2, 70, 6, 5)])
@pytest.mark.parametrize("pool", [0, 1])
def test_instance_norm_leaky_pool(be, shape, pool):
    """Per-frame statistics + LeakyReLU(0.2) + avg-pool with odd sizes (the discriminator's DownBlock3D,
    modules/discriminator.py:26-33) through the mnk_norm_* entry points."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(8)
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    z = F.leaky_relu(F.instance_norm(xd, weight=gd, bias=bd, eps=1e-5), 0.2)
    if pool:
        z = F.avg_pool2d(z, 2)
    dz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(dz)
    ld = ceil4(c)
    X = be.t(to_nhwc(x))
    nws = be.query("mnk_norm_workspace_floats", h * w, n, ld)
    ws, sums = be.empty(nws), be.empty(2 * n * c)
    be.call("mnk_norm_stats", X, ld, h * w, n, c, sums, ws, nws)
    mean, invstd, scale = be.empty(n * c), be.empty(n * c), be.empty(n * c)
    G, Bt = be.t(gamma), be.t(beta)
    be.call("mnk_norm_finalize", sums, float(h * w), G, None, None, 0.0, 1e-5, c, n, 0, mean, invstd, scale)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    Z = be.empty(n, ho, wo, ld)
    be.call("mnk_norm_act_fwd", X, ld, mean, scale, Bt, 1, Z, ld, 0, n, h, w, c, 0.2, pool)
    DZ = be.t(to_nhwc(dz.float()))
    bs = be.empty(2 * n * c)
    be.call("mnk_norm_act_bwd_stats", X, ld, DZ, ld, 0, mean, invstd, scale, Bt, 1, n, h, w, c, 0.2, pool, bs, ws, nws)
    DY = be.empty(n, h, w, ld)
    be.call("mnk_norm_act_bwd_apply", X, ld, DZ, ld, 0, mean, invstd, scale, Bt, 1, bs, float(h * w), 1, DY, ld, n, h, w, c,
            0.2, pool)
    be.sync()
    assert maxerr(from_nhwc(Z.cpu(), c), z) < 2e-5
    assert relerr(from_nhwc(DY.cpu(), c), xd.grad) < 1e-4
    # affine gradients = per-frame sums added over the frames
    assert relerr(bs.cpu()[:n * c].view(n, c).sum(0), bd.grad) < 1e-4
    assert relerr(bs.cpu()[n * c:].view(n, c).sum(0), gd.grad) < 1e-4


@pytest.mark.parametrize("shape", [(2, 5, 6, 4), (3, 45, 4, 6), (2, 300, 2, 2), (1, 64, 16, 16), (4, 520, 8, 8), (2, 1024, 2, 2)])
@pytest.mark.parametrize("pool", [0, 1])
def test_bn_small_layer_one_launch_forms(be, shape, pool):
    """mnk_bn_small_fwd / _bwd (statistics + finalisation + apply, and the whole backward, in one launch each) against
    F.batch_norm(training) + relu + avg_pool2d in fp64."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    z = F.relu(F.batch_norm(xd, rm, rv, gd, bd, True, 0.1, 1e-5))
    if pool:
        z = F.avg_pool2d(z, 2)
    dz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(dz)
    assert n * h * w <= be.query("mnk_bn_small_rows")
    ld = ceil4(c)
    X = be.t(to_nhwc(x))
    mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
    RM, RV, G, Bt = be.t(rm0.clone()), be.t(rv0.clone()), be.t(gamma), be.t(beta)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    Z = be.empty(n, ho, wo, ld)
    be.call("mnk_bn_small_fwd", None, 0, ld, 1, None, X, ld, n, h, w, c, G, Bt, RM, RV, 0.1, 1e-5, mean, invstd, scale, Z, ld, 1,
            pool)
    be.sync()
    assert maxerr(from_nhwc(Z.cpu(), c), z) < 2e-5
    assert torch.all(Z.cpu()[..., c:] == 0)
    assert maxerr(RM.cpu(), rm) < 1e-5 and maxerr(RV.cpu(), rv) < 1e-4
    DZ = be.t(to_nhwc(dz.float()))
    bs, DY = be.empty(2 * c), be.empty(n, h, w, ld)
    be.call("mnk_bn_small_bwd", X, ld, DZ, ld, mean, invstd, scale, Bt, float(n * h * w), n, h, w, c, 1, pool, bs, DY, ld)
    be.sync()
    assert relerr(bs.cpu()[:c], bd.grad) < 1e-4 and relerr(bs.cpu()[c:], gd.grad) < 1e-4
    assert relerr(from_nhwc(DY.cpu(), c), xd.grad) < 1e-4
    assert torch.all(DY.cpu()[..., c:] == 0)


@pytest.mark.parametrize("splits", [1, 3, 4, 7, 8, 13, 21])
def test_bn_small_adds_any_number_of_split_partials_in_order(be, splits):
    """mnk_bn_small_fwd on synthetic [split][M][ldw] partials: y == bias + the partials added one after another in fp32 (the
    kernel fetches eight / four of them ahead of the adds; the order of the adds is the plain loop's), then BatchNorm + ReLU."""
    n, h, w, c = 2, 4, 6, 22
    ld, rows = ceil4(c), 2 * 4 * 6
    g = torch.Generator().manual_seed(splits)
    part = torch.randn(splits, rows, ld, generator=g)
    b = torch.randn(c, generator=g)
    y = torch.zeros(rows, ld)
    y[:, :c] = b
    for sp in range(splits):
        y = y + part[sp]                                  # fp32, in order
    y[:, c:] = 0
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    yd = y[:, :c].double().reshape(n, h, w, c).permute(0, 3, 1, 2)
    zref = F.relu(F.batch_norm(yd, torch.zeros(c).double(), torch.ones(c).double(), gamma.double(), beta.double(), True, 0.1, 1e-5))
    Y, Z = be.empty(n, h, w, ld), be.empty(n, h, w, ld)
    mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
    RM, RV = be.zeros(c), be.zeros(c) + 1
    be.call("mnk_bn_small_fwd", be.t(part), splits, ld, 1, be.t(b), Y, ld, n, h, w, c, be.t(gamma), be.t(beta), RM, RV,
            0.1, 1e-5, mean, invstd, scale, Z, ld, 1, 0)
    be.sync()
    assert torch.equal(Y.cpu().reshape(rows, ld), y)
    assert maxerr(from_nhwc(Z.cpu(), c), zref) < 2e-5


@pytest.mark.parametrize("up", [0, 1], ids=["3x3", "sub-pixel-up"])
def test_bn_small_sums_the_split_k_partials_of_the_convolution_in_front(be, up):
    """a split-K convolution (few output tiles) run with MNK_CONV_DEFER_SPLITK leaves [split][phase][M][ldw] partials; the
    small-layer BatchNorm kernel sums them (+ bias) into y, for the plain 3x3 form and for the sub-pixel form of an
    up-sampled convolution (phase-major partials scattered to (2i + a, 2j + b))."""
    n, hl, wl, cin, cout = 3, 4, 4, 136, 72       # >= 32 K steps in both forms: split along K
    g = torch.Generator().manual_seed(8)
    x = torch.randn(n, cin, hl, wl, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    h, w = (2 * hl, 2 * wl) if up else (hl, wl)
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up else x.double()
    yref = F.conv2d(xin, wt.double(), b.double(), padding=1)
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    rm, rv = torch.zeros(cout).double(), torch.ones(cout).double()
    zref = F.relu(F.batch_norm(yref, rm, rv, gamma.double(), beta.double(), True, 0.1, 1e-5))
    X, W = be.t(to_nhwc(x)), be.t(wt)
    ld = ceil4(cout)
    Y = be.empty(n, h, w, ld)
    if up:
        wp = be.empty(be.query("mnk_conv3x3_up_packed_floats", cout, cin, 0))
        be.call("mnk_conv3x3_up_pack_fwd", W, wp, cout, cin, 0)
        nws = be.query("mnk_conv3x3_up_workspace_floats", n, hl, wl, cin, 0, cout)
        splits = be.query("mnk_conv3x3_up_splits", n, hl, wl, cin, 0, cout)
        ws = be.empty(nws)
        be.call("mnk_conv3x3_up_fwd", X, X.shape[-1], cin, None, 0, 0, 4, wp, be.t(b), Y, ld, n, hl, wl, cout, ws, nws, None)
    else:
        wp = be.empty(be.query("mnk_conv3x3_packed_floats", cout, cin, 0))
        be.call("mnk_conv3x3_pack_fwd", W, wp, cout, cin, 0)
        nws = be.query("mnk_conv3x3_workspace_floats", n, h, w, cin, 0, cout)
        splits = be.query("mnk_conv3x3_splits", n, h, w, cin, 0, cout)
        ws = be.empty(nws)
        be.call("mnk_conv3x3_fwd", X, X.shape[-1], cin, None, 0, 0, 2 | 4, wp, be.t(b), None, 0, Y, ld, n, h, w, cout, ws, nws, None)
    assert splits > 1 and nws > 0
    mean, invstd, scale = be.empty(cout), be.empty(cout), be.empty(cout)
    RM, RV = be.zeros(cout), be.t(torch.ones(cout))
    Z = be.empty(n, h, w, ld)
    be.call("mnk_bn_small_fwd", ws, splits, ld, 4 if up else 1, be.t(b), Y, ld, n, h, w, cout, be.t(gamma), be.t(beta), RM, RV,
            0.1, 1e-5, mean, invstd, scale, Z, ld, 1, 0)
    be.sync()
    assert relerr(from_nhwc(Y.cpu(), cout), yref) < 2e-6
    assert torch.all(Y.cpu()[..., cout:] == 0)
    assert maxerr(from_nhwc(Z.cpu(), cout), zref) < 2e-5


@pytest.mark.parametrize("shape,pool", [((3, 45, 4, 6), 0), ((2, 64, 8, 8), 1), ((2, 5, 6, 4), 1)])
def test_apply_from_finished_sums_equals_finalize_then_apply(be, shape, pool):
    """mnk_bn_act_fwd_sums (the SyncBN path: the sums come out of an all-reduce, the finalisation happens inside the apply
    pass) against mnk_bn_finalize + mnk_bn_act_fwd: outputs, saved statistics and running statistics to the bit."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.3
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    ld, rows = ceil4(c), n * h * w
    X = be.t(to_nhwc(x))
    nws = be.query("mnk_bn_workspace_floats", rows, ld)
    ws, sums = be.empty(nws), be.empty(2 * c)
    be.call("mnk_bn_stats", X, ld, rows, c, sums, ws, nws)
    count = float(3 * rows)                 # as if three ranks had contributed (the sums are whatever the exchange left)
    G, Bt = be.t(gamma), be.t(beta)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    out = []
    for fused in (False, True):
        mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
        RM, RV = be.t(rm0.clone()), be.t(rv0.clone())
        Z = be.zeros(n, ho, wo, ld)
        if fused:
            be.call("mnk_bn_act_fwd_sums", X, ld, sums, count, G, Bt, RM, RV, 0.1, 1e-5, 1, mean, invstd, scale, Z, ld, 0, n, h, w, c,
                    1, pool)
        else:
            be.call("mnk_bn_finalize", sums, count, G, RM, RV, 0.1, 1e-5, c, 1, mean, invstd, scale)
            be.call("mnk_bn_act_fwd", X, ld, mean, scale, Bt, Z, ld, 0, n, h, w, c, 1, pool)
        be.sync()
        out.append([t.cpu().clone() for t in (Z, mean, invstd, scale, RM, RV)])
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("ca,cb", [(3, 11), (8, 12), (4, 6)])
def test_concat_with_an_embedding_shared_by_both_batch_halves(be, ca, cb):
    """ops.Concat2PairFn (the batched discriminator pass [generated | real] with ONE key-point embedding):
    out[n] = [a[n] | b[n mod B]], gradient of b = sum of both halves -- torch.cat / repeat and their autograd."""
    from mnk import ops
    g = torch.Generator().manual_seed(12)
    b2, h, w = 4, 5, 3
    a = torch.randn(b2, ca, h, w, generator=g)
    e = torch.randn(b2 // 2, cb, h, w, generator=g)
    A, E = be.t(to_nhwc(a)).requires_grad_(True), be.t(to_nhwc(e)).requires_grad_(True)
    out = ops.Concat2PairFn.apply(A, ca, E, cb)
    ref = torch.cat([a, e.repeat(2, 1, 1, 1)], dim=1)
    be.sync()
    assert torch.equal(from_nhwc(out.detach().cpu(), ca + cb), ref)
    assert torch.all(out.detach().cpu()[..., ca + cb:] == 0)
    go = torch.randn(b2, ca + cb, h, w, generator=g)
    out.backward(be.t(to_nhwc(go)))
    be.sync()
    assert torch.equal(from_nhwc(A.grad.cpu(), ca), go[:, :ca])
    assert maxerr(from_nhwc(E.grad.cpu(), cb), go[:b2 // 2, ca:] + go[b2 // 2:, ca:]) < 1e-6


# ---- the kernels that carry the SyncBN exchange, on a handle of world size 1 ---------------------------------------------------------
# (round 5: the emulator has stand-ins for the peer-to-peer primitives -- tests/hipemu/include/hip/hip_runtime.h -- so the product
# sources need no emulator switch around these kernels and their indexing / reductions / exchange protocol run in the CPU suite
# too: a rank pushes its sums into its own mailbox and adds "all one" ranks in order: 0.f + x = x, the plain kernels' numbers)
def _one_rank_handle(be):
    import ctypes
    h = ctypes.c_void_p()
    be.lib.call("mnk_p2p_create", 0, 1, ctypes.byref(h))
    return h


@pytest.mark.parametrize("n", [1, 7, 256, 2112])
def test_one_rank_exchange_is_the_identity(be, n):
    import ctypes
    h = _one_rank_handle(be)
    try:
        g = torch.Generator().manual_seed(n)
        x = be.t(torch.randn(n, generator=g))
        for _ in range(6):                       # more exchanges than mailbox slots: the sequence number advances
            out = be.empty(n)
            be.lib.call("mnk_p2p_allreduce", h, x.data_ptr(), out.data_ptr(), n, 2000, be.stream())
            be.sync()
            assert torch.equal(out.cpu(), x.cpu())
        flag = ctypes.c_int(-1)
        be.lib.call("mnk_p2p_error", h, ctypes.byref(flag))
        assert flag.value == 0
    finally:
        be.lib.call("mnk_p2p_destroy", h)


@pytest.mark.parametrize("shape,pool", [((2, 10, 8, 8), 0), ((3, 37, 4, 6), 1), ((2, 64, 16, 16), 0)])
def test_synchronised_statistics_of_one_rank_equal_the_plain_ones(be, shape, pool):
    """mnk_bn_stats_sync / mnk_bn_stats_finish_sync / mnk_bn_act_bwd_stats_sync (the statistics' second stage with the exchange
    inside) against mnk_bn_stats / mnk_bn_stats_finish / mnk_bn_act_bwd_stats: bit for bit on one rank"""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(5)
    ld, rows = ceil4(c), n * h * w
    X = be.t(to_nhwc(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.2))
    nws = be.query("mnk_bn_workspace_floats", rows, ld)
    hnd = _one_rank_handle(be)
    try:
        plain, ws = be.empty(2 * c), be.empty(nws)
        be.call("mnk_bn_stats", X, ld, rows, c, plain, ws, nws)
        loc, glob, ws2 = be.empty(2 * c), be.empty(2 * c), be.empty(nws)
        be.call("mnk_bn_stats_sync", hnd, X, ld, rows, c, loc, glob, ws2, nws, 2000)
        be.sync()
        assert torch.equal(glob.cpu(), plain.cpu()) and torch.equal(loc.cpu(), plain.cpu())
        # second stage alone, on synthetic per-block partials [row_blocks][2][ld]
        rb = 37
        part = be.t(torch.randn(rb, 2, ld, generator=g))
        p1, l2, g2 = be.empty(2 * c), be.empty(2 * c), be.empty(2 * c)
        be.call("mnk_bn_stats_finish", part, rb, ld, c, p1)
        be.call("mnk_bn_stats_finish_sync", hnd, part, rb, ld, c, l2, g2, 2000)
        be.sync()
        assert torch.equal(g2.cpu(), p1.cpu()) and torch.equal(l2.cpu(), p1.cpu())
        # backward statistics
        mean, invstd = be.t(torch.randn(c, generator=g) * 0.1), be.t(torch.rand(c, generator=g) + 0.5)
        scale, beta = be.t(torch.rand(c, generator=g) + 0.5), be.t(torch.randn(c, generator=g) * 0.3)
        ho, wo = (h // 2, w // 2) if pool else (h, w)
        DZ = be.t(to_nhwc(torch.randn(n, c, ho, wo, generator=g)))
        b1, ws3 = be.empty(2 * c), be.empty(nws)
        be.call("mnk_bn_act_bwd_stats", X, ld, DZ, ld, 0, mean, invstd, scale, beta, n, h, w, c, 1, pool, b1, ws3, nws)
        bl, bg, ws4 = be.empty(2 * c), be.empty(2 * c), be.empty(nws)
        be.call("mnk_bn_act_bwd_stats_sync", hnd, X, ld, DZ, ld, 0, mean, invstd, scale, beta, n, h, w, c, 1, pool, bl, bg, ws4,
                nws, 2000)
        be.sync()
        assert torch.equal(bg.cpu(), b1.cpu()) and torch.equal(bl.cpu(), b1.cpu())
    finally:
        be.lib.call("mnk_p2p_destroy", hnd)


@pytest.mark.parametrize("shape,pool", [((2, 10, 4, 4), 0), ((4, 37, 4, 6), 1), ((2, 130, 2, 2), 0)])
def test_one_launch_small_layer_forms_with_the_exchange_inside(be, shape, pool):
    """mnk_bn_small_fwd_sync / mnk_bn_small_bwd_sync on one rank against F.batch_norm(training) + relu + avg_pool2d in fp64 (the
    bounds of the single-process forms) and against mnk_bn_small_fwd / _bwd (another thread map: equal to rounding)"""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    z = F.relu(F.batch_norm(xd, rm, rv, gd, bd, True, 0.1, 1e-5))
    if pool:
        z = F.avg_pool2d(z, 2)
    dz = torch.randn(z.shape, generator=g, dtype=torch.float64)
    z.backward(dz)
    ld = ceil4(c)
    X = be.t(to_nhwc(x))
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    G, Bt = be.t(gamma), be.t(beta)
    hnd = _one_rank_handle(be)
    try:
        outs = {}
        for name in ("mnk_bn_small_fwd", "mnk_bn_small_fwd_sync"):
            mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
            RM, RV, Z = be.t(rm0.clone()), be.t(rv0.clone()), be.empty(n, ho, wo, ld)
            args = (None, 0, ld, 1, None, X, ld, n, h, w, c, G, Bt, RM, RV, 0.1, 1e-5, mean, invstd, scale, Z, ld, 1, pool)
            be.call(name, *(((hnd,) + args + (2000,)) if name.endswith("_sync") else args))
            be.sync()
            outs[name] = (Z.cpu(), RM.cpu(), RV.cpu(), mean, invstd, scale)
            assert maxerr(from_nhwc(Z.cpu(), c), z) < 2e-5 and torch.all(Z.cpu()[..., c:] == 0)
            assert maxerr(RM.cpu(), rm) < 1e-5 and maxerr(RV.cpu(), rv) < 1e-4
        assert maxerr(outs["mnk_bn_small_fwd"][0], outs["mnk_bn_small_fwd_sync"][0]) < 5e-6
        mean, invstd, scale = outs["mnk_bn_small_fwd_sync"][3:]
        DZ = be.t(to_nhwc(dz.float()))
        bs, DY = be.empty(2 * c), be.empty(n, h, w, ld)
        be.call("mnk_bn_small_bwd_sync", hnd, X, ld, DZ, ld, mean, invstd, scale, Bt, float(n * h * w), n, h, w, c, 1, pool, bs, DY,
                ld, 2000)
        be.sync()
        assert relerr(bs.cpu()[:c], bd.grad) < 1e-4 and relerr(bs.cpu()[c:], gd.grad) < 1e-4
        assert relerr(from_nhwc(DY.cpu(), c), xd.grad) < 1e-4 and torch.all(DY.cpu()[..., c:] == 0)
    finally:
        be.lib.call("mnk_p2p_destroy", hnd)


@pytest.mark.parametrize("splits", [1, 2, 3, 4, 5, 8, 13, 33])
@pytest.mark.parametrize("phases,pool,relu,c", [(1, 0, 1, 22), (1, 1, 1, 45), (4, 0, 1, 3), (4, 1, 0, 66), (1, 1, 0, 7)])
def test_eval_norm_layer_on_split_partials(be, splits, phases, pool, relu, c):
    """mnk_bn_eval_split_fwd on synthetic partials ([split][M][ldw], or [split][phase][M][ldw] of a sub-pixel form): z must be,
    BIT FOR BIT, what the library's own two launches make of them -- the split reduction's summation order (four interleaved
    groups, (g0 + g1) + (g2 + g3), then the bias; restated here in torch fp32) followed by mnk_bn_act_fwd on that y -- and close to
    the fp64 evaluation-mode BatchNorm (+ ReLU, + 2x2 average pool) of sync_batchnorm/batchnorm.py:57-59 / util.py:56-57,100-107."""
    n, h, w = 2, 4, 6                                   # (h, w): the convolution's output size
    ld, rows = ceil4(c), n * h * w
    g = torch.Generator().manual_seed(100 * splits + c)
    part = torch.randn(splits, rows, ld, generator=g)   # row index = pixel (phases 1) / phase-major low-resolution pixel (4)
    b = torch.randn(c, generator=g)
    gr = [torch.zeros(rows, ld) for _ in range(4)]
    s = 0
    while s + 4 <= splits:
        for e in range(4):
            gr[e] = gr[e] + part[s + e]
        s += 4
    for e in range(3):
        if s + e < splits:
            gr[e] = gr[e] + part[s + e]
    y = (gr[0] + gr[1]) + (gr[2] + gr[3])
    y[:, :c] = y[:, :c] + b
    y[:, c:] = 0
    if phases == 4:                                     # phase-major rows -> the (n, h, w) pixel order of y
        hl, wl = h // 2, w // 2
        y = y.reshape(2, 2, n, hl, wl, ld).permute(2, 3, 0, 4, 1, 5).reshape(rows, ld)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rm, rv = torch.randn(c, generator=g) * 0.2, torch.rand(c, generator=g) + 0.5
    mean, invstd, scale = be.empty(c), be.empty(c), be.empty(c)
    be.call("mnk_bn_eval_coeffs", be.t(gamma), be.t(rm), be.t(rv), 1e-5, c, mean, invstd, scale)
    ho, wo = (h // 2, w // 2) if pool else (h, w)
    Z, Z2 = be.empty(n, ho, wo, ld), be.empty(n, ho, wo, ld)
    be.call("mnk_bn_eval_split_fwd", be.t(part), splits, ld, phases, be.t(b), mean, scale, be.t(beta), Z, ld, n, h, w, c, relu, pool)
    be.call("mnk_bn_act_fwd", be.t(y.reshape(n, h, w, ld)), ld, mean, scale, be.t(beta), Z2, ld, 0, n, h, w, c, relu, pool)
    be.sync()
    assert torch.equal(Z.cpu(), Z2.cpu())
    assert float(Z.cpu()[..., c:].abs().max()) == 0.0 if ld > c else True
    yd = y[:, :c].double().reshape(n, h, w, c).permute(0, 3, 1, 2)
    zref = F.batch_norm(yd, rm.double(), rv.double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
    if relu:
        zref = F.relu(zref)
    if pool:
        zref = F.avg_pool2d(zref, 2)
    assert maxerr(from_nhwc(Z.cpu(), c), zref) < 2e-5
