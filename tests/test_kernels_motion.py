"""Grouped 1x1 conv, 1x1+sigmoid, motion-field head and the bilinear warp kernels against torch/oracle (fp64)."""
import os

import pytest
import torch

from mnk import knobs
import torch.nn.functional as F

from _util import to_nhwc, from_nhwc, ceil4, relerr, maxerr
from oracle import restate


@pytest.mark.parametrize("G,S", [(11, 4), (5, 6), (3, 1)])
def test_gconv1x1(be, G, S):
    g = torch.Generator().manual_seed(0)
    n, h, w = 2, 5, 7
    c = G * S
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, S, 1, 1, generator=g)
    b = torch.randn(c, generator=g)
    xd, wd, bd = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, bd, groups=G)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    ld = ceil4(c)
    rows = n * h * w
    X = be.t(to_nhwc(x))
    Y = be.empty(n, h, w, ld)
    be.call("mnk_gconv1x1_fwd", X, ld, be.t(wt), be.t(b), Y, ld, rows, G, S)
    DY = be.t(to_nhwc(dy.float()))
    DX = be.empty(n, h, w, ld)
    be.call("mnk_gconv1x1_bwd_data", DY, ld, be.t(wt), DX, ld, rows, G, S)
    nws = be.query("mnk_gconv1x1_workspace_floats", rows, G, S)
    ws, DW, DB = be.empty(nws), be.empty(c, S), be.empty(c)
    be.call("mnk_gconv1x1_bwd_weight", X, ld, DY, ld, DW, DB, rows, G, S, ws, nws)
    be.sync()
    assert relerr(from_nhwc(Y.cpu(), c), ref) < 1e-6 and torch.all(Y.cpu()[..., c:] == 0)
    assert relerr(from_nhwc(DX.cpu(), c), xd.grad) < 1e-6 and torch.all(DX.cpu()[..., c:] == 0)
    assert relerr(DW.cpu().view(c, S, 1, 1), wd.grad) < 1e-5
    assert relerr(DB.cpu(), bd.grad) < 1e-5


@pytest.mark.parametrize("cin,cout,D,hw", [(45, 3, 1, (6, 5)), (13, 1, 2, (6, 5)), (70, 4, 1, (6, 5)), (45, 3, 1, (20, 23)),
                                            (8, 2, 1, (362, 363)), (200, 1, 1, (1, 3)), (35, 2, 1, (6, 5))])
def test_conv1x1_sigmoid(be, cin, cout, D, hw):
    """the row-tile forms (one partial tile, several tiles, and -- 262 812 pixel rows -- blocks that walk more than one
    tile: the 2048-block cap of the weight-gradient partials), the few-rows x many-channels forms (rows wider than 64
    floats: a wavefront per pixel row); the thread-per-pixel kernels remain for tensors that are not 16-byte aligned."""
    g = torch.Generator().manual_seed(1)
    B, (H, W) = 2, hw
    x = torch.randn(B * D, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    xd, wd, bd = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    ref4 = torch.sigmoid(F.conv2d(xd, wd.view(cout, cin, 1, 1), bd))
    ref = restate.unfold(ref4, B)                        # (B,cout,D,H,W)
    dout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dout)
    ld = ceil4(cin)
    X = be.t(to_nhwc(x))
    OUT = be.empty(B, cout, D, H, W)
    be.call("mnk_conv1x1_sigmoid_fwd", X, ld, cin, be.t(wt), be.t(b), OUT, B, D, H, W, cout)
    rows = B * D * H * W
    nws = be.query("mnk_conv1x1_workspace_floats", rows, cin, cout)
    ws, DX, DW, DB = be.empty(nws), be.empty(B * D, H, W, ld), be.empty(cout, cin), be.empty(cout)
    be.call("mnk_conv1x1_sigmoid_bwd", X, ld, cin, be.t(wt), OUT, be.t(dout.float()), DX, ld, DW, DB, B, D, H, W, cout,
            ws, nws)
    be.sync()
    assert maxerr(OUT.cpu(), ref) < 1e-6
    assert relerr(from_nhwc(DX.cpu(), cin), xd.grad) < 1e-5 and torch.all(DX.cpu()[..., cin:] == 0)
    assert relerr(DW.cpu(), wd.grad) < 1e-5 and relerr(DB.cpu(), bd.grad) < 1e-5
    # data gradient only / parameter gradients only (NULL dw / NULL dx)
    DX2, DW2, DB2 = be.empty(B * D, H, W, ld), be.empty(cout, cin), be.empty(cout)
    be.call("mnk_conv1x1_sigmoid_bwd", X, ld, cin, be.t(wt), OUT, be.t(dout.float()), DX2, ld, None, None, B, D, H, W, cout,
            None, 0)
    be.call("mnk_conv1x1_sigmoid_bwd", X, ld, cin, be.t(wt), OUT, be.t(dout.float()), None, ld, DW2, DB2, B, D, H, W, cout,
            ws, nws)
    be.sync()
    assert torch.equal(DX2.cpu(), DX.cpu()) and relerr(DW2.cpu(), wd.grad) < 1e-5 and relerr(DB2.cpu(), bd.grad) < 1e-5


@pytest.mark.parametrize("use_mask,use_corr", [(1, 1), (1, 0), (0, 1)])
def test_motion_field(be, use_mask, use_corr):
    g = torch.Generator().manual_seed(2)
    n, h, w, K = 3, 6, 9, 4
    cpred = (K + 1) * use_mask + 2 * use_corr
    pred = torch.randn(n, cpred, h, w, generator=g) * 2
    delta = torch.randn(n, K + 1, 2, generator=g) * 0.3
    delta[:, 0] = 0
    pd, dd = pred.double().requires_grad_(True), delta.double().requires_grad_(True)
    rel = 0
    if use_mask:
        mask = F.softmax(pd[:, :K + 1], dim=1)
        rel = (dd.view(n, K + 1, 2, 1, 1) * mask.unsqueeze(2)).sum(1)
    if use_corr:
        rel = rel + pd[:, -2:]
    ref = rel.permute(0, 2, 3, 1) + restate.make_coordinate_grid(h, w, torch.float64).view(1, h, w, 2)
    df = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(df)
    ld = ceil4(cpred)
    PR = be.t(to_nhwc(pred, pad_value=float("nan")))
    FLD = be.empty(n, h, w, 2)
    be.call("mnk_motion_field_fwd", PR, ld, be.t(delta), n, h, w, K, use_mask, use_corr, FLD)
    DP, DD = be.empty(n, h, w, ld), be.empty(n, K + 1, 2)
    be.call("mnk_motion_field_bwd", PR, ld, be.t(delta), be.t(df.float()), n, h, w, K, use_mask, use_corr, DP, ld, DD)
    be.sync()
    assert maxerr(FLD.cpu(), ref) < 2e-6
    assert relerr(from_nhwc(DP.cpu(), cpred), pd.grad) < 1e-5 and torch.all(DP.cpu()[..., cpred:] == 0)
    if use_mask:
        assert relerr(DD.cpu(), dd.grad) < 1e-5


@pytest.mark.parametrize("use_corr", [0, 1])
def test_motion_field_from_key_points_equals_the_delta_form(be, use_corr):
    """mnk_motion_field_kp_*: kp_source.mean - kp_driving.mean formed inside the kernels (dense_motion_module.py:52-54) -- the
    field and d pred of the delta form to the bit, the two key-point gradients +- its d delta."""
    g = torch.Generator().manual_seed(4)
    n, h, w, K = 3, 6, 9, 4
    cpred = K + 1 + 2 * use_corr
    ld = ceil4(cpred)
    pred = torch.randn(n, cpred, h, w, generator=g) * 2
    ms, md = torch.rand(n, K, 2, generator=g) * 2 - 1, torch.rand(n, K, 2, generator=g) * 2 - 1
    delta = torch.cat([torch.zeros(n, 1, 2), ms - md], dim=1)
    df = torch.randn(n, h, w, 2, generator=g)
    PR = be.t(to_nhwc(pred))
    F1, F2 = be.empty(n, h, w, 2), be.empty(n, h, w, 2)
    be.call("mnk_motion_field_fwd", PR, ld, be.t(delta), n, h, w, K, 1, use_corr, F1)
    be.call("mnk_motion_field_kp_fwd", PR, ld, be.t(ms), be.t(md), n, h, w, K, use_corr, F2)
    DP1, DP2, DD = be.empty(n, h, w, ld), be.empty(n, h, w, ld), be.empty(n, K + 1, 2)
    GS, GD = be.empty(n, K, 2), be.empty(n, K, 2)
    be.call("mnk_motion_field_bwd", PR, ld, be.t(delta), be.t(df), n, h, w, K, 1, use_corr, DP1, ld, DD)
    be.call("mnk_motion_field_kp_bwd", PR, ld, be.t(ms), be.t(md), be.t(df), n, h, w, K, use_corr, DP2, ld, GS, GD)
    be.sync()
    assert torch.equal(F1.cpu(), F2.cpu()) and torch.equal(DP1.cpu(), DP2.cpu())
    assert torch.equal(GS.cpu(), DD.cpu()[:, 1:]) and torch.equal(GD.cpu(), -DD.cpu()[:, 1:])


DEFORM_CASES = [("same", (2, 5, 16, 16), 0), ("down", (2, 6, 4, 4), 0), ("up", (2, 3, 32, 32), 0),
                ("one", (2, 7, 1, 1), 0), ("tri_down", (2, 6, 8, 8), 1), ("tri_up", (2, 3, 32, 32), 1),
                ("wide", (2, 300, 4, 4), 0)]


@pytest.mark.parametrize("tag,shape,mode", DEFORM_CASES)
def test_deform(be, tag, shape, mode):
    g = torch.Generator().manual_seed(3)
    n, c, h, w = shape
    hf = wf = 16
    inp = torch.rand(n, c, 1, h, w, generator=g)
    field = torch.cat([restate.make_coordinate_grid(hf, wf).view(1, 1, hf, wf, 2).repeat(n, 1, 1, 1, 1) +
                       0.4 * torch.randn(n, 1, hf, wf, 2, generator=g), torch.zeros(n, 1, hf, wf, 1)], -1)
    i64, f64 = inp.double().requires_grad_(True), field.double().requires_grad_(True)
    ref = restate.deform_input(i64, f64, "nearest" if mode == 0 else "trilinear")       # (n,c,1,h,w)
    dout = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dout)
    ld = ceil4(c)
    X = be.t(to_nhwc(inp[:, :, 0]))
    FL = be.t(field[:, 0, :, :, :2])
    ldo, off = ld + 8, 3
    OUT = be.zeros(n, h, w, ldo)
    be.call("mnk_deform_fwd", X, ld, c, h, w, FL, hf, wf, mode, OUT, ldo, off, n)
    DO = be.zeros(n, h, w, ldo)
    DO[..., off:off + c] = be.t(dout[:, :, 0].float().permute(0, 2, 3, 1))
    DI, DF = be.empty(n, h, w, ld), be.zeros(n, hf, wf, 2)
    DI.fill_(float("nan"))                     # d input is WRITTEN by the gather pass (pad channels 0): no zero fill
    nws = be.query("mnk_deform_bwd_workspace_floats", c, h, w, n)
    WS = be.empty(nws)
    be.call("mnk_deform_bwd", X, ld, c, h, w, FL, hf, wf, mode, DO, ldo, off, DI, DF, n, WS, nws)
    be.sync()
    assert torch.all(DI.cpu()[..., c:] == 0)
    # white-noise image x (w-1)/2 coordinate scaling amplifies fp32 coordinate rounding to a few 1e-6
    assert maxerr(OUT.cpu()[..., off:off + c].permute(0, 3, 1, 2), ref[:, :, 0]) < 1e-5
    assert torch.all(OUT.cpu()[..., :off] == 0) and torch.all(OUT.cpu()[..., off + c:] == 0)
    assert relerr(from_nhwc(DI.cpu(), c), i64.grad[:, :, 0]) < 1e-5
    assert relerr(DF.cpu(), f64.grad[:, 0, :, :, :2]) < 1e-5


@pytest.mark.parametrize("kind", ["collapse", "stripes", "far", "random"])
@pytest.mark.parametrize("shape,mode", [((2, 5, 16, 16), 0), ((2, 70, 8, 8), 0), ((1, 4, 40, 24), 0), ((2, 6, 32, 32), 1),
                                        ((1, 3, 1, 1), 0), ((1, 300, 2, 2), 0), ((1, 5, 128, 128), 0)])
def test_deform_backward_is_deterministic_and_order_exact(be, kind, shape, mode):
    """grid_sample's adjoint with colliding sampling points (the reference's CPU backward, generator.py:51-58, is a loop over
    output pixels: a fixed-order sum per source texel).  The gather form must (a) give the same BITS on every run, also for
    fields that send many pixels to one texel, and (b) equal an fp32 restatement of that pixel-ordered loop bit for bit --
    nothing is summed in an order of the hardware's choosing."""
    g = torch.Generator().manual_seed(11)
    n, c, h, w = shape
    hf, wf = (16, 16) if h != 40 else (20, 12)
    grid = restate.make_coordinate_grid(hf, wf).view(1, hf, wf, 2).repeat(n, 1, 1, 1)
    if kind == "collapse":       # every pixel samples (nearly) the same point: one texel quad receives everything
        field = torch.zeros(n, hf, wf, 2) + 0.13 + 1e-3 * torch.randn(n, hf, wf, 2, generator=g)
    elif kind == "stripes":      # all rows sample one source row
        field = grid.clone()
        field[..., 1] = -0.31
    elif kind == "far":          # most points outside [-1, 1] (zeros padding), some non-finite
        field = grid * 3.0 + torch.randn(n, hf, wf, 2, generator=g)
        field[0, 0, 0, 0] = float("inf")
        field[0, 1, 1, 1] = float("nan")
    else:
        field = grid + 0.5 * torch.randn(n, hf, wf, 2, generator=g)
    inp = torch.rand(n, h, w, ceil4(c), generator=g)
    inp[..., c:] = 0
    ld, ldo = ceil4(c), ceil4(c) + 4
    dout = torch.randn(n, h, w, ldo, generator=g)
    X, FL, DO = be.t(inp), be.t(field), be.t(dout)
    nws = be.query("mnk_deform_bwd_workspace_floats", c, h, w, n)
    runs = []
    for _ in range(3):
        DI, DF, WS = be.empty(n, h, w, ld), be.zeros(n, hf, wf, 2), be.empty(nws)
        DI.fill_(float("nan"))
        be.call("mnk_deform_bwd", X, ld, c, h, w, FL, hf, wf, mode, DO, ldo, 0, DI, DF, n, WS, nws)
        be.sync()
        runs.append((DI.cpu(), DF.cpu()))
    for di, df in runs[1:]:
        assert torch.equal(di.view(torch.int32), runs[0][0].view(torch.int32))
        assert torch.equal(df.view(torch.int32), runs[0][1].view(torch.int32))
    if kind == "far" or h * w > 4096 or mode != 0:
        return
    # (b) the pixel-ordered loop in fp32 with the kernels' own sampling arithmetic (nearest pick of the field: exact)
    fl = restate.resize_field(torch.cat([field, torch.zeros(n, hf, wf, 1)], -1).view(n, 1, hf, wf, 3), (h, w),
                              "nearest")[:, 0, :, :, :2].float()
    want = torch.zeros(n, h, w, ld)
    for b in range(n):
        ix = ((fl[b, ..., 0] + 1.0) / 2.0) * float(w - 1)
        iy = ((fl[b, ..., 1] + 1.0) / 2.0) * float(h - 1)
        fx, fy = torch.floor(ix), torch.floor(iy)
        for py in range(h):
            for px in range(w):
                x0, y0 = int(fx[py, px]), int(fy[py, px])
                for dy in (0, 1):
                    for dx in (0, 1):
                        yy, xx = y0 + dy, x0 + dx
                        if 0 <= yy < h and 0 <= xx < w:
                            wx = (ix[py, px] - fx[py, px]) if dx else ((fx[py, px] + 1.0) - ix[py, px])
                            wy = (iy[py, px] - fy[py, px]) if dy else ((fy[py, px] + 1.0) - iy[py, px])
                            want[b, yy, xx, :c] += dout[b, py, px, :c] * (wx * wy)
    got = runs[0][0]
    # fused multiply-add contraction is the compiler's choice: equal to the last bit or two, not necessarily the same bits
    assert maxerr(got, want) <= 4e-7 * float(want.abs().max() + 1)


def test_deform_matches_reference_golden(be):
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "functions.pt"), weights_only=False)
    field = gold["deform_field"]
    for tag, mode in (("same", 0), ("down", 0), ("up", 0), ("one", 0), ("tri_down", 1), ("tri_up", 1)):
        inp, ref = gold["deform_%s_in" % tag], gold["deform_%s_out" % tag]
        n, c, _, h, w = inp.shape
        ld = ceil4(c)
        OUT = be.zeros(n, h, w, ld)
        be.call("mnk_deform_fwd", be.t(to_nhwc(inp[:, :, 0])), ld, c, h, w, be.t(field[:, 0, :, :, :2]), 16, 16, mode,
                OUT, ld, 0, n)
        be.sync()
        assert maxerr(from_nhwc(OUT.cpu(), c), ref[:, :, 0]) < 1e-5, tag


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("emb_ch", [0, 6, 8])
def test_all_warps_in_one_launch_equal_the_per_level_launches(be, monkeypatch, emb_ch, mode):
    """ops.WarpAllFn with mnk_warp_levels_fwd / _bwd (MNK_WARP_LEVELS=1: every level's warp and embedding copy in one launch
    each way) against one launch per level: outputs and the embedding gradient to the bit (same gathers in the same order),
    the scatter-added gradients (atomics) to rounding."""
    from mnk import ops
    g = torch.Generator().manual_seed(6)
    n = 2
    field = be.t((torch.rand(n, 16, 16, 2, generator=g) * 2.4 - 1.2))
    emb = be.t(torch.randn(n, 8, 8, ceil4(emb_ch), generator=g)) if emb_ch else None
    if emb is not None:
        emb[..., emb_ch:] = 0
    shapes = [(8, 4, 4), (5, 8, 8), (4, 16, 16), (3, 32, 32)]
    specs = tuple((c, emb_ch if i % 2 == 0 else 0) for i, (c, _, _) in enumerate(shapes))
    inps = [be.t(torch.randn(n, h, w, ceil4(c), generator=g)) for c, h, w in shapes]
    for t, (c, _, _) in zip(inps, shapes):
        t[..., c:] = 0
    douts = [torch.randn(n, h, w, ceil4(c + ke), generator=g) for (c, h, w), (_, ke) in zip(shapes, specs)]
    for d, (c, _, _), (_, ke) in zip(douts, shapes, specs):
        d[..., c + ke:] = 0                      # gradients of acts: zero pad channels

    def run(flag):
        monkeypatch.setitem(knobs.FORMS, "WARP_LEVELS", flag == "1")
        f = field.clone().requires_grad_(True)
        e = emb.clone().requires_grad_(True) if emb is not None else None
        xs = [t.clone().requires_grad_(True) for t in inps]
        outs = ops.WarpAllFn.apply(f, e, mode, specs, *xs)
        used = [i for i in range(len(outs)) if i != 1]             # level 1's output goes nowhere (no gradient reaches it)
        torch.autograd.backward([outs[i] for i in used], [be.t(douts[i]) for i in used])
        be.sync()
        return ([o.detach().cpu() for o in outs], f.grad.cpu(), e.grad.cpu() if e is not None else None,
                [x.grad.cpu() for i, x in enumerate(xs) if i != 1])

    o0, gf0, ge0, gx0 = run("0")
    o1, gf1, ge1, gx1 = run("1")
    for a, b in zip(o0, o1):          # (mode 1: the bilinear blends are the same expressions in two kernels -- equal up to the
        assert torch.equal(a, b) if mode == 0 else maxerr(a, b) < 1e-6        # compiler's choice of fused multiply-adds)
    if emb is not None:
        assert torch.equal(ge0, ge1) if mode == 0 else relerr(ge1, ge0) < 1e-6
    assert relerr(gf1, gf0) < 1e-5
    for a, b in zip(gx0, gx1):
        assert relerr(b, a) < 1e-5


@pytest.mark.parametrize("ca,cb,halves", [(3, 11, 1), (3, 11, 2), (8, 6, 1), (5, 4, 2), (64, 10, 1)])
def test_concat_of_two_acts_and_its_adjoint_in_one_launch_each(be, ca, cb, halves):
    """torch.cat([a, b], channel) on acts (modules/util.py:185; discriminator.py:50-52) with the pad channels written by the
    same launch; halves = 2: the batched discriminator pass, b holds half the frames and its gradient is the sum of both halves"""
    from mnk import ops
    g = torch.Generator().manual_seed(2)
    n, h, w = 4, 5, 7
    a = torch.zeros(n, h, w, ceil4(ca))
    a[..., :ca] = torch.randn(n, h, w, ca, generator=g)
    b = torch.zeros(n // halves, h, w, ceil4(cb))
    b[..., :cb] = torch.randn(n // halves, h, w, cb, generator=g)
    A, B = be.t(a).requires_grad_(True), be.t(b).requires_grad_(True)
    fn = ops.Concat2PairFn if halves == 2 else ops.Concat2Fn
    out = fn.apply(A, ca, B, cb)
    want = torch.zeros(n, h, w, ceil4(ca + cb))
    want[..., :ca] = a[..., :ca]
    want[..., ca:ca + cb] = torch.cat([b] * halves, 0)[..., :cb]
    be.sync()
    assert torch.equal(out.detach().cpu(), want)
    go = torch.zeros_like(want)
    go[..., :ca + cb] = torch.randn(n, h, w, ca + cb, generator=g)
    out.backward(be.t(go))
    be.sync()
    ga, gb = A.grad.cpu(), B.grad.cpu()
    assert torch.equal(ga[..., :ca], go[..., :ca]) and torch.all(ga[..., ca:] == 0)
    wb = go[..., ca:ca + cb]
    wb = wb[:n // 2] + wb[n // 2:] if halves == 2 else wb
    assert torch.equal(gb[..., :cb], wb) and torch.all(gb[..., cb:] == 0)
