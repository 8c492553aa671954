"""conv3x3 implicit-GEMM kernels (forward, dgrad via packed weights, wgrad) against torch CPU conv2d.
The same bodies run on the CPU emulator build (-m "not gpu") and on the MI355X (-m gpu)."""
import pytest
import torch
import torch.nn.functional as F

from _util import to_nhwc, from_nhwc, ceil4, relerr, maxerr

@pytest.fixture(params=["f32-mfma", "bf16x3"], autouse=True)
def gemm_mode(request, be):
    """every test of this file runs twice: the forward / data-gradient GEMMs on v_mfma_f32_32x32x2_f32 and on the bf16 matrix
    cores through the exact three-way split of both fp32 operands (tuning value gemm_bf16x3; csrc/mnk_common.h) -- the SAME
    tolerances against fp64 in both modes: the split form is an fp32-accurate product, not a reduced-precision one"""
    on = 1 if request.param == "bf16x3" else 0
    be.lib.call("mnk_set_tuning", b"gemm_bf16x3", on)
    be.lib.call("mnk_set_tuning", b"wgrad_bf16x3", on)       # the tap-major weight-gradient kernels' form (transposing loader)
    be.lib.call("mnk_set_tuning", b"gemm16_bf16x3", on)      # the 16x16-tile kernels' form (pairs of K steps)
    yield request.param
    be.lib.call("mnk_set_tuning", b"gemm_bf16x3", 0)
    be.lib.call("mnk_set_tuning", b"wgrad_bf16x3", 0)
    be.lib.call("mnk_set_tuning", b"gemm16_bf16x3", 0)


CASES = [
    # N, H, W, C0, C1, Cout, ups, bias, residual
    (2, 8, 8, 3, 0, 20, 0, True, False),
    (1, 6, 10, 20, 13, 45, 0, True, True),      # two sources, odd channel counts, residual
    (2, 8, 8, 16, 10, 70, 1, True, False),      # nearest x2 up-sampling on both sources, 128-wide tile
    (3, 4, 4, 40, 0, 136, 0, False, False),     # split-K path (few tiles), two N tiles
    (1, 2, 2, 32, 0, 10, 1, True, False),
    (2, 1, 1, 17, 0, 33, 0, True, False),       # 1x1 spatial (moving-gif bottleneck at 64x64 input)
]


def _inputs(case, seed=0):
    n, h, w, c0, c1, cout, ups, bias, res = case
    g = torch.Generator().manual_seed(seed)
    hs, ws = (h // 2, w // 2) if ups else (h, w)
    x0 = torch.randn(n, c0, hs, ws, generator=g)
    x1 = torch.randn(n, c1, hs, ws, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g) if bias else None
    r = torch.randn(n, cout, h, w, generator=g) if res else None
    return x0, x1, wt, b, r


def _ref_fwd(case, x0, x1, wt, b, r):
    ups = case[6]
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), padding=1)
    if r is not None:
        y = y + r.double()
    return y


def _run_fwd(be, case, x0, x1, wt, b, r, clean=False):
    """clean = False: generic loader, pad channels of the sources hold NaN and must be ignored by the kernel;
    clean = True: MNK_CONV_CLEAN_PADS (zero pads vouched for) -> the raw-buffer-load 3x3 loader."""
    n, h, w, c0, c1, cout, ups, _, _ = case
    wp = be.empty(be.query("mnk_conv3x3_packed_floats", cout, c0, c1))
    be.call("mnk_conv3x3_pack_fwd", be.t(wt), wp, cout, c0, c1)
    padv = 0.0 if clean else float("nan")
    ups = int(ups) | (2 if clean else 0)
    X0 = be.t(to_nhwc(x0, pad_value=padv))
    X1 = be.t(to_nhwc(x1, pad_value=padv)) if x1 is not None else None
    R = be.t(to_nhwc(r)) if r is not None else None
    ldy = ceil4(cout)
    Y = be.empty(n, h, w, ldy)
    nws = be.query("mnk_conv3x3_workspace_floats", n, h, w, c0, c1, cout)
    ws = be.empty(max(nws, 1))
    be.call("mnk_conv3x3_fwd", X0, X0.shape[-1], c0, X1, X1.shape[-1] if x1 is not None else 0, c1, ups, wp,
            be.t(b) if b is not None else None, R, R.shape[-1] if r is not None else 0, Y, ldy, n, h, w, cout,
            ws, nws, None)
    be.sync()
    return Y.cpu()


# more shapes for the fast loader: all five igemm tiles + both 16x16 tiles, up-sampled, ragged M, odd sizes, frame borders
FAST_CASES = [
    (3, 6, 10, 33, 0, 129, 0, True, False),      # 128x128 tile (Cout > 128 -> 2 N tiles), odd H / W
    (2, 16, 16, 24, 17, 40, 1, True, True),      # 16x16x4 kernel BN = 48, two up-sampled sources, residual
    (5, 4, 6, 20, 0, 9, 0, False, False),        # BN = 16
    (2, 8, 8, 72, 0, 30, 1, True, False),        # 128x32 tile, up-sampled
    (6, 64, 64, 5, 0, 20, 0, True, False),       # 192 M tiles, no split-K, XCD re-chunked order
]


@pytest.mark.parametrize("clean", [False, True])
@pytest.mark.parametrize("case", CASES + FAST_CASES)
def test_conv3x3_forward(be, case, clean):
    x0, x1, wt, b, r = _inputs(case)
    Y = _run_fwd(be, case, x0, x1, wt, b, r, clean)
    ref = _ref_fwd(case, x0, x1, wt, b, r)
    cout = case[5]
    assert relerr(from_nhwc(Y, cout), ref) < 2e-6
    assert torch.all(Y[..., cout:] == 0), "pad channels of the output must be written as zero"


@pytest.mark.parametrize("case", CASES[:4])
def test_conv3x3_dgrad(be, case):
    """dx = conv(dy, flipped/transposed weights): the forward kernel with mnk_conv3x3_pack_dgrad weights."""
    n, h, w, c0, c1, cout, ups, _, _ = case
    if ups:
        pytest.skip("dgrad of an up-sampled input = dgrad at full resolution + mnk_sumpool2x2 (tested below)")
    x0, x1, wt, b, r = _inputs(case)
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(n, cout, h, w, generator=g)
    x = (x0 if x1 is None else torch.cat([x0, x1], 1)).double().requires_grad_(True)
    F.conv2d(x, wt.double(), None, padding=1).backward(dy.double())
    DY = be.t(to_nhwc(dy))
    for c_start, c_cnt in ((0, c0),) + (((c0, c1),) if c1 else ()):
        wp = be.empty(be.query("mnk_conv3x3_packed_floats", c_cnt, cout, 0))
        be.call("mnk_conv3x3_pack_dgrad", be.t(wt), wp, cout, c0 + c1, c_start, c_cnt)
        ld = ceil4(c_cnt)
        DX = be.empty(n, h, w, ld)
        nws = be.query("mnk_conv3x3_workspace_floats", n, h, w, cout, 0, c_cnt)
        ws = be.empty(max(nws, 1))
        for flags in (0, 2):     # generic / fast loader (to_nhwc pads with zeros)
            be.call("mnk_conv3x3_fwd", DY, DY.shape[-1], cout, None, 0, 0, flags, wp, None, None, 0, DX, ld, n, h, w,
                    c_cnt, ws, nws, None)
            be.sync()
            assert relerr(from_nhwc(DX.cpu(), c_cnt), x.grad[:, c_start:c_start + c_cnt]) < 2e-6


# W >= 16: the LDS-halo weight-gradient kernel (64-pixel tiles, zero-bordered halo, 9 taps from LDS)
HALO_CASES = [
    (1, 16, 16, 40, 0, 60, 0, False, False),
    (1, 32, 32, 60, 50, 64, 1, False, False),    # up-sampled sources, two sources, TC = 32
    (2, 16, 32, 120, 0, 130, 0, False, False),   # 2 ci tiles x 3 co tiles, several splits
    (1, 6, 64, 64, 0, 40, 0, False, False),      # TC = 64 (one row per tile), ragged H
    (1, 16, 16, 20, 0, 45, 0, False, False),     # poorly filled slab -> stays on the gather kernel
    (1, 8, 8, 72, 0, 48, 0, False, False),       # tap-major form, 64 x 128 tile
    (2, 5, 7, 33, 0, 129, 0, False, False),      # tap-major form, odd H / W (magic-number division), ragged tiles
    (1, 8, 8, 80, 0, 24, 1, False, False),       # tap-major form, 32 x 128 tile, up-sampled source
    (2, 32, 32, 72, 0, 40, 0, False, False),     # tap-major form, 16 pixel splits -> two-stage split reduction
    (6, 64, 64, 5, 0, 20, 0, False, False),      # gather kernel, 192 pixel splits -> two-stage split reduction
    (2, 64, 64, 40, 0, 60, 0, False, False),     # narrow-layer (n16) kernel, many tile splits -> two-stage reduction
    (1, 8, 8, 35, 0, 10, 0, False, False),       # n16 kernel tile-count variants <co tiles, ci tiles>: <1,3>
    (1, 8, 8, 12, 0, 12, 0, False, False),       # <1,1>
    (1, 8, 16, 20, 0, 14, 1, False, False),      # <1,2>, up-sampled source
    (2, 8, 8, 30, 0, 30, 0, False, False),       # <2,2>
    (1, 16, 8, 40, 0, 20, 0, False, False),      # <2,3>
    (1, 8, 8, 10, 0, 40, 0, False, False),       # <3,1>
    (1, 8, 8, 50, 0, 56, 0, False, False),       # <3,3> with two tiles along co and ci (48-wide tiles)
]


# tap-major kernel through the fast loader (rows of >= 16 pixels): tiles 128x128 / 128x64 / 64x128 / 32x128, up-sampled
# sources, several frames per split and splits that end inside a row / inside a frame
WFAST_CASES = [
    (3, 16, 16, 130, 0, 140, 0, False, False),
    (2, 32, 16, 20, 0, 136, 1, False, False),
    (2, 16, 48, 80, 0, 40, 1, False, False),
    (3, 6, 20, 72, 0, 24, 0, False, False),
    (1, 32, 128, 64, 0, 128, 0, False, False),   # one image row per pixel split: the last split's buffer of the row below is empty
]


# small maps through the tap-major kernel's COMPACT K (round 6): only the (pixel, tap) pairs inside the source are multiplied --
# 1 x 1 ... 8 x 8 maps, non-square, two sources, several frames per K step, several pixel splits, and the sub-pixel form of
# up-sampled layers down to a 1 x 1 source (12 of its 16 pseudo taps are empty there)
COMPACT_CASES = [
    (4, 2, 2, 80, 0, 72, 0, False, False),
    (3, 1, 1, 70, 0, 130, 0, False, False),
    (5, 4, 4, 96, 0, 80, 0, False, False),
    (2, 2, 4, 40, 70, 68, 0, False, False),
    (70, 2, 2, 72, 0, 48, 0, False, False),
    (6, 2, 2, 72, 0, 40, 1, False, False),
    (3, 4, 4, 130, 0, 72, 1, False, False),
    (2, 8, 4, 70, 0, 130, 1, False, False),
    (20, 4, 4, 72, 0, 48, 1, False, False),
]


@pytest.mark.parametrize("clean", [False, True])
@pytest.mark.parametrize("case", CASES + HALO_CASES + WFAST_CASES + COMPACT_CASES)
def test_conv3x3_wgrad(be, case, clean):
    """clean = False: NaN pad channels in x must be ignored (generic loaders); clean = True: MNK_CONV_CLEAN_PADS, zero
    pads -> the buffer-load loader of the tap-major kernel where the shape allows it (W >= 16)."""
    n, h, w, c0, c1, cout, ups, _, _ = case
    x0, x1, wt, b, r = _inputs(case)
    g = torch.Generator().manual_seed(6)
    dy = torch.randn(n, cout, h, w, generator=g)
    wd = wt.double().requires_grad_(True)
    x = x0 if x1 is None else torch.cat([x0, x1], 1)
    if ups:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    F.conv2d(x.double(), wd, None, padding=1).backward(dy.double())
    DY = be.t(to_nhwc(dy))
    DW = be.empty(cout, c0 + c1, 3, 3)
    for src, c_start, c_cnt in ((x0, 0, c0),) + (((x1, c0, c1),) if c1 else ()):
        X = be.t(to_nhwc(src, pad_value=0.0 if clean else float("nan")))
        nws = be.query("mnk_conv3x3_up_wgrad_workspace_floats" if ups else "mnk_conv3x3_wgrad_workspace_floats", n, h, w, c_cnt,
                       cout)
        ws = be.empty(max(nws, 1))
        be.call("mnk_conv3x3_wgrad", X, X.shape[-1], c_cnt, int(ups) | (2 if clean else 0), DY, DY.shape[-1], cout, DW,
                c0 + c1, c_start, n, h, w, ws, nws)
    be.sync()
    assert relerr(DW.cpu(), wd.grad) < 2e-6


# (n, hi, wi, c0, c1, cout, kh, kw, pad): odd kernels / paddings through the K x K buffer-load loader (MNK_CONV_CLEAN_PADS
# on anything that is not 3x3 pad 1), ragged channel counts, two sources, frames narrower than a 128-pixel tile
KXK_CASES = [(2, 9, 11, 5, 0, 7, 3, 3, 0), (2, 9, 11, 5, 0, 7, 3, 3, 2), (1, 12, 10, 18, 0, 33, 5, 5, 2),
             (3, 7, 6, 16, 3, 9, 4, 4, 1), (2, 6, 5, 3, 0, 70, 2, 2, 1), (5, 4, 4, 20, 0, 12, 4, 4, 3),
             (2, 40, 24, 6, 0, 16, 1, 1, 0), (150, 4, 4, 24, 0, 8, 4, 4, 0), (70, 5, 5, 8, 0, 20, 4, 4, 0)]


@pytest.mark.parametrize("clean", [0, 2], ids=["generic-loader", "kxk-buffer-loader"])
@pytest.mark.parametrize("case", KXK_CASES)
def test_conv2d_any_kernel_and_padding(be, case, clean):
    n, hi, wi, c0, c1, cout, kh, kw, pad = case
    g = torch.Generator().manual_seed(17)
    x0 = torch.randn(n, c0, hi, wi, generator=g)
    x1 = torch.randn(n, c1, hi, wi, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, kh, kw, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    xin = torch.cat([x0, x1], 1) if c1 else x0
    ref = F.conv2d(xin.double(), wt.double(), b.double(), padding=pad)
    ho, wo = ref.shape[2], ref.shape[3]
    nt = kh * kw
    wp = be.empty(be.query("mnk_conv2d_packed_floats", cout, c0, c1, nt))
    be.call("mnk_conv2d_pack_fwd", be.t(wt), wp, cout, c0, c1, nt)
    X0 = be.t(to_nhwc(x0))
    X1 = be.t(to_nhwc(x1)) if c1 else None
    Y = be.empty(n, ho, wo, ceil4(cout)).fill_(float("nan"))
    nws = be.query("mnk_conv2d_workspace_floats", n, ho, wo, c0, c1, cout, nt)
    ws = be.empty(max(nws, 1))
    be.call("mnk_conv2d_fwd", X0, ceil4(c0), c0, X1, ceil4(c1) if c1 else 0, c1, clean, hi, wi, kh, kw, pad, wp, be.t(b),
            None, 0, Y, ceil4(cout), n, ho, wo, cout, ws, nws, None)
    be.sync()
    assert relerr(from_nhwc(Y.cpu(), cout), ref) < 2e-6
    assert torch.all(Y.cpu()[..., cout:] == 0)


def test_pack_all_equals_separate_packs(be):
    """mnk_conv3x3_pack_all (one launch per training forward) == pack_fwd + pack_dgrad of both sources."""
    cout, c0, c1 = 21, 18, 7
    g = torch.Generator().manual_seed(11)
    wt = be.t(torch.randn(cout, c0 + c1, 1, 3, 3, generator=g))
    nf = be.query("mnk_conv3x3_packed_floats", cout, c0, c1)
    n0, n1 = be.query("mnk_conv3x3_packed_floats", c0, cout, 0), be.query("mnk_conv3x3_packed_floats", c1, cout, 0)
    ref_f, ref_0, ref_1 = be.empty(nf), be.empty(n0), be.empty(n1)
    be.call("mnk_conv3x3_pack_fwd", wt, ref_f, cout, c0, c1)
    be.call("mnk_conv3x3_pack_dgrad", wt, ref_0, cout, c0 + c1, 0, c0)
    be.call("mnk_conv3x3_pack_dgrad", wt, ref_1, cout, c0 + c1, c0, c1)
    a_f, a_0, a_1 = be.empty(nf), be.empty(n0), be.empty(n1)
    be.call("mnk_conv3x3_pack_all", wt, a_f, a_0, a_1, cout, c0, c1)
    be.sync()
    assert torch.equal(a_f.cpu(), ref_f.cpu()) and torch.equal(a_0.cpu(), ref_0.cpu()) and torch.equal(a_1.cpu(), ref_1.cpu())
    b_f = be.empty(nf)
    be.call("mnk_conv3x3_pack_all", wt, b_f, None, None, cout, c0, c1)
    be.sync()
    assert torch.equal(b_f.cpu(), ref_f.cpu())


def test_pack_multi_equals_pack_all_per_layer(be):
    """mnk_conv3x3_pack_multi (every layer of a model in one launch, descriptor table in device memory) == one
    mnk_conv3x3_pack_all per layer, including layers without data-gradient layouts and two-source layers; layers flagged
    as up-sampled convolutions get the packs of their sub-pixel forms (== mnk_conv3x3_up_pack_fwd / _up_pack_dgrad)."""
    import numpy as np
    layers = [(21, 18, 7, True, True, 0), (3, 35, 0, True, False, 0), (40, 5, 0, False, False, 0), (16, 16, 16, False, True, 0),
              (70, 33, 0, True, False, 0), (21, 18, 7, True, True, 1), (40, 5, 0, False, False, 1), (33, 70, 20, True, True, 1)]
    g = torch.Generator().manual_seed(12)
    rec = np.zeros(len(layers), dtype=np.dtype([("p", "<u8", 4), ("i", "<i4", 6)]))
    keep, tiles = [], 0
    for k, (cout, c0, c1, d0, d1, up) in enumerate(layers):
        wt = be.t(torch.randn(cout, c0 + c1, 1, 3, 3, generator=g))
        bufs = []
        for _ in range(2):                    # [0]: pack_multi's outputs, [1]: the per-layer reference
            nf = be.query("mnk_conv3x3_up_packed_floats" if up else "mnk_conv3x3_packed_floats", cout, c0, c1)
            n0 = be.query("mnk_conv3x3_up_dgrad_packed_floats", cout, c0) if up else be.query("mnk_conv3x3_packed_floats", c0, cout, 0)
            n1 = (be.query("mnk_conv3x3_up_dgrad_packed_floats", cout, c1) if up else
                  be.query("mnk_conv3x3_packed_floats", c1, cout, 0)) if c1 else 0
            f = be.empty(nf).fill_(float("nan"))
            a0 = be.empty(n0).fill_(float("nan")) if d0 else None
            a1 = be.empty(n1).fill_(float("nan")) if d1 and c1 else None
            bufs.append((f, a0, a1))
        if up:
            be.call("mnk_conv3x3_up_pack_fwd", wt, bufs[1][0], cout, c0, c1)
            if bufs[1][1] is not None:
                be.call("mnk_conv3x3_up_pack_dgrad", wt, bufs[1][1], cout, c0 + c1, 0, c0)
            if bufs[1][2] is not None:
                be.call("mnk_conv3x3_up_pack_dgrad", wt, bufs[1][2], cout, c0 + c1, c0, c1)
        else:
            be.call("mnk_conv3x3_pack_all", wt, *bufs[1], cout, c0, c1)
        f, a0, a1 = bufs[0]
        rec["p"][k] = (wt.data_ptr(), f.data_ptr(), a0.data_ptr() if a0 is not None else 0,
                       a1.data_ptr() if a1 is not None else 0)
        rec["i"][k] = (cout, c0, c1, tiles, up, 0)
        tiles += ((c0 + 15) // 16 + (c1 + 15) // 16) * ((cout + 15) // 16)
        keep.append((wt, bufs, up))
    descs = be.t(torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()))
    be.call("mnk_conv3x3_pack_multi", descs, len(layers), tiles)
    be.sync()
    for wt, (got, ref), up in keep:
        for a, b in zip(got, ref):
            assert (a is None) == (b is None)
            if a is not None:
                if up:      # sums of up to four taps: the tile kernel and the element kernel add in the same order
                    assert float((a.cpu() - b.cpu()).abs().max()) <= 1e-6 * float(b.cpu().abs().max())
                else:
                    assert torch.equal(a.cpu().nan_to_num(nan=7.0), b.cpu().nan_to_num(nan=7.0))


def test_sumpool2x2(be):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 6, 8, generator=g)
    X = be.t(to_nhwc(x))
    Y = be.empty(2, 3, 4, 8)
    be.call("mnk_sumpool2x2", X, 8, Y, 8, 2, 6, 8, 5)
    be.sync()
    ref = F.avg_pool2d(x, 2) * 4
    assert maxerr(from_nhwc(Y.cpu(), 5), ref) < 1e-5
    assert torch.all(Y.cpu()[..., 5:] == 0)


# large enough (>= 192 output tiles) that the forward is not split along K -- only then are the statistics fused
STAT_CASES = [(6, 64, 64, 5, 0, 20, 0, True, False), (6, 64, 64, 4, 3, 40, 1, True, True)]
# few output tiles, deep K: split along K -- the reduction of the partials leaves the statistics (one launch for both)
SPLIT_STAT_CASES = [(2, 8, 8, 64, 0, 22, 0, True, False), (3, 6, 10, 40, 24, 70, 0, False, True), (2, 8, 8, 48, 0, 33, 1, True, False)]


@pytest.mark.parametrize("case", STAT_CASES + SPLIT_STAT_CASES)
def test_conv3x3_fused_bn_statistics(be, case):
    """The conv epilogue's per-block column sums (unsplit launches) or the split-K reduction's (split launches)
    + mnk_bn_stats_finish == sums over the written output; the output does not depend on the statistics being asked for."""
    n, h, w, c0, c1, cout, ups, _, _ = case
    split = be.query("mnk_conv3x3_splits", n, h, w, c0, c1, cout) > 1
    assert split == (case in SPLIT_STAT_CASES)
    x0, x1, wt, b, r = _inputs(case, seed=3)
    wp = be.empty(be.query("mnk_conv3x3_packed_floats", cout, c0, c1))
    be.call("mnk_conv3x3_pack_fwd", be.t(wt), wp, cout, c0, c1)
    X0 = be.t(to_nhwc(x0))
    X1 = be.t(to_nhwc(x1)) if x1 is not None else None
    R = be.t(to_nhwc(r)) if r is not None else None
    ldy = ceil4(cout)
    Y = be.empty(n, h, w, ldy)
    nst = be.query("mnk_conv3x3_stats_floats", n, h, w, c0, c1, cout)
    assert nst > 0 and nst % (2 * ldy) == 0
    st = be.empty(nst).fill_(float("nan"))
    nws = be.query("mnk_conv3x3_workspace_floats", n, h, w, c0, c1, cout)
    assert (nws > 0) == split
    ws = be.empty(nws) if nws else None
    Y.fill_(float("nan"))
    args = (X0, X0.shape[-1], c0, X1, X1.shape[-1] if x1 is not None else 0, c1, ups, wp,
            be.t(b) if b is not None else None, R, R.shape[-1] if r is not None else 0)
    be.call("mnk_conv3x3_fwd", *args, Y, ldy, n, h, w, cout, ws, nws, st)
    sums = be.empty(2 * cout)
    be.call("mnk_bn_stats_finish", st, nst // (2 * ldy), ldy, cout, sums)
    Y2 = be.empty(n, h, w, ldy)
    be.call("mnk_conv3x3_fwd", *args, Y2, ldy, n, h, w, cout, ws, nws, None)
    be.sync()
    assert torch.equal(Y.cpu(), Y2.cpu())
    if split:           # the 64-outputs x 4-split-groups reduction (MNK_REDUCE_V4=0) adds the splits in the same order
        Y3 = be.empty(n, h, w, ldy)
        be.lib.call("mnk_set_tuning", b"reduce_v4", 0)
        try:
            be.call("mnk_conv3x3_fwd", *args, Y3, ldy, n, h, w, cout, ws, nws, None)
            be.sync()
        finally:
            be.lib.call("mnk_set_tuning", b"reduce_v4", 1)
        assert torch.equal(Y.cpu(), Y3.cpu())
    assert relerr(from_nhwc(Y.cpu(), cout), _ref_fwd(case, x0, x1, wt, b, r)) < 2e-6
    assert torch.all(Y.cpu()[..., cout:] == 0)
    y = from_nhwc(Y.cpu(), cout).double()
    ref = torch.cat([y.sum(dim=(0, 2, 3)), (y * y).sum(dim=(0, 2, 3))])
    assert relerr(sums.cpu(), ref) < 1e-5


# ---- general K x K form: the discriminator's (1,4,4) convolutions without padding (modules/discriminator.py:17-18)
K4_CASES = [(2, 16, 16, 13, 32), (1, 13, 13, 32, 64), (3, 6, 6, 64, 20), (2, 5, 4, 7, 1)]


@pytest.mark.parametrize("clean", [0, 2], ids=["generic-loader", "kxk-buffer-loader"])
@pytest.mark.parametrize("case", K4_CASES)
def test_conv4x4_nopad_forward_dgrad_wgrad(be, case, clean):
    n, hi, wi, cin, cout = case
    kh = kw = 4
    ho, wo = hi - 3, wi - 3
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, cin, hi, wi, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    xd, wd = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, b.double())
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    X, W, ldx, ldy = be.t(to_nhwc(x)), be.t(wt), ceil4(cin), ceil4(cout)
    # forward
    wp = be.empty(be.query("mnk_conv2d_packed_floats", cout, cin, 0, 16))
    be.call("mnk_conv2d_pack_fwd", W, wp, cout, cin, 0, 16)
    Y = be.empty(n, ho, wo, ldy)
    nws = be.query("mnk_conv2d_workspace_floats", n, ho, wo, cin, 0, cout, 16)
    ws = be.empty(max(nws, 1))
    be.call("mnk_conv2d_fwd", X, ldx, cin, None, 0, 0, clean, hi, wi, kh, kw, 0, wp, be.t(b), None, 0, Y, ldy, n, ho, wo,
            cout, ws, nws, None)
    # data gradient: the same kernel on dy, pad = k-1, flipped/transposed pack
    DY = be.t(to_nhwc(dy.float()))
    wpd = be.empty(be.query("mnk_conv2d_packed_floats", cin, cout, 0, 16))
    be.call("mnk_conv2d_pack_dgrad", W, wpd, cout, cin, 0, cin, 16)
    DX = be.empty(n, hi, wi, ldx)
    nws2 = be.query("mnk_conv2d_workspace_floats", n, hi, wi, cout, 0, cin, 16)
    ws2 = be.empty(max(nws2, 1))
    be.call("mnk_conv2d_fwd", DY, ldy, cout, None, 0, 0, clean, ho, wo, kh, kw, 3, wpd, None, None, 0, DX, ldx, n, hi, wi,
            cin, ws2, nws2, None)
    # weight gradient
    DW = be.empty(cout, cin, kh, kw)
    nws3 = be.query("mnk_conv2d_wgrad_workspace_floats", n, ho, wo, cin, cout, kh, kw, 0)
    ws3 = be.empty(max(nws3, 1))
    be.call("mnk_conv2d_wgrad", X, ldx, cin, 0, hi, wi, kh, kw, 0, DY, ldy, cout, DW, cin, 0, n, ho, wo, ws3, nws3)
    be.sync()
    assert relerr(from_nhwc(Y.cpu(), cout), ref) < 2e-6
    assert relerr(from_nhwc(DX.cpu(), cin), xd.grad) < 2e-6
    assert relerr(DW.cpu(), wd.grad) < 2e-6


# ---- sub-pixel forms of [nearest x2 up-sampling -> 3x3 / pad 1] (UpBlock3D): (n, h_low, w_low, c0, c1, cout, bias)
UP_CASES = [(2, 4, 4, 16, 10, 70, True),        # two sources, 64x128 tile, split-K (few tiles)
            (1, 1, 1, 32, 0, 24, True),         # 1x1 -> 2x2 (every tap of some phases falls outside)
            (3, 3, 5, 20, 0, 136, False),       # odd sizes, two N tiles
            (2, 16, 16, 24, 17, 40, True),      # no split-K: scattered epilogue + fused statistics
            (6, 32, 32, 5, 0, 20, True),        # many tiles, XCD re-chunked order, 128x32 tile
            (2, 8, 8, 72, 0, 130, True)]


@pytest.mark.parametrize("case", UP_CASES)
def test_conv3x3_upsampled_subpixel_forward_and_dgrad(be, case):
    """mnk_conv3x3_up_fwd (four 2x2 phase convolutions on the low-resolution input) and mnk_conv3x3_up_dgrad (one 4x4 /
    stride 2 convolution over dy) against conv2d(interpolate(x, 2, 'nearest')) and its input gradient in fp64."""
    n, h, w, c0, c1, cout, bias = case
    g = torch.Generator().manual_seed(31)
    x0 = torch.randn(n, c0, h, w, generator=g)
    x1 = torch.randn(n, c1, h, w, generator=g) if c1 else None
    wt = torch.randn(cout, c0 + c1, 3, 3, generator=g) * 0.2
    b = torch.randn(cout, generator=g) if bias else None
    x = (x0 if x1 is None else torch.cat([x0, x1], 1)).double().requires_grad_(True)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt.double(), None if b is None else b.double(), padding=1)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    W = be.t(wt)
    wp = be.empty(be.query("mnk_conv3x3_up_packed_floats", cout, c0, c1))
    be.call("mnk_conv3x3_up_pack_fwd", W, wp, cout, c0, c1)
    X0 = be.t(to_nhwc(x0))
    X1 = be.t(to_nhwc(x1)) if c1 else None
    ldy = ceil4(cout)
    Y = be.empty(n, 2 * h, 2 * w, ldy)
    nws = be.query("mnk_conv3x3_up_workspace_floats", n, h, w, c0, c1, cout)
    ws = be.empty(max(nws, 1))
    nst = be.query("mnk_conv3x3_up_stats_floats", n, h, w, c0, c1, cout)
    st = be.empty(nst) if nst else None
    be.call("mnk_conv3x3_up_fwd", X0, X0.shape[-1], c0, X1, X1.shape[-1] if c1 else 0, c1, 0, wp, be.t(b) if bias else None, Y, ldy,
            n, h, w, cout, ws, nws, st)
    be.sync()
    assert relerr(from_nhwc(Y.cpu(), cout), ref) < 2e-6
    assert torch.all(Y.cpu()[..., cout:] == 0), "pad channels of the output must be written as zero"
    if nst:
        sums = be.empty(2 * cout)
        be.call("mnk_bn_stats_finish", st, nst // (2 * ldy), ldy, cout, sums)
        be.sync()
        yd = ref.detach()
        assert relerr(sums.cpu(), torch.cat([yd.sum(dim=(0, 2, 3)), (yd * yd).sum(dim=(0, 2, 3))])) < 1e-5
    DY = be.t(to_nhwc(dy.float()))
    for c_start, c_cnt in ((0, c0),) + (((c0, c1),) if c1 else ()):
        wd = be.empty(be.query("mnk_conv3x3_up_dgrad_packed_floats", cout, c_cnt))
        be.call("mnk_conv3x3_up_pack_dgrad", W, wd, cout, c0 + c1, c_start, c_cnt)
        ld = ceil4(c_cnt)
        DX = be.empty(n, h, w, ld)
        nws2 = be.query("mnk_conv3x3_up_dgrad_workspace_floats", n, h, w, cout, c_cnt)
        ws2 = be.empty(max(nws2, 1))
        be.call("mnk_conv3x3_up_dgrad", DY, ldy, cout, wd, DX, ld, n, h, w, c_cnt, ws2, nws2)
        be.sync()
        assert relerr(from_nhwc(DX.cpu(), c_cnt), x.grad[:, c_start:c_start + c_cnt]) < 2e-6
        assert torch.all(DX.cpu()[..., c_cnt:] == 0)


# ---- measured / forced launch plans (csrc/plan_table.h, tools/plan_tune.py) -------------------------------------------------
def _last_plan(be):
    import numpy as np
    out = np.zeros(8, dtype=np.int64)
    be.lib.call("mnk_last_plan", out.ctypes.data)
    return tuple(int(v) for v in out)


@pytest.mark.parametrize("case", [(2, 8, 8, 40, 0, 70, 0, True, False), (2, 8, 8, 16, 10, 40, 1, True, False),
                                  (1, 8, 8, 20, 0, 12, 0, True, True)])
def test_forced_launch_plans_compute_the_same_convolution(be, case):
    """every (block tile, split-K) plan the sweep may force -- and so every row csrc/plan_table.h may hold -- is the same
    convolution; a plan the kernels have no instantiation for is refused and the rule's plan runs."""
    x0, x1, wt, b, r = _inputs(case, seed=5)
    ref = _ref_fwd(case, x0, x1, wt, b, r)
    cout = case[5]
    seen = set()
    try:
        for bm, bn, splits in [(0, 0, 0), (64, 128, 1), (128, 128, 2), (64, 64, 3), (128, 64, 1), (128, 32, 4), (128, 48, 1),
                               (128, 16, 2), (64, 32, 1), (64, 16, 1), (0, 0, 7)]:
            for name, v in (("force_bm", bm), ("force_bn", bn), ("force_splits", splits)):
                be.lib.call("mnk_set_tuning", name.encode(), v)
            Y = _run_fwd(be, case, x0, x1, wt, b, r, clean=True)
            plan = _last_plan(be)
            assert plan[1] == cout and plan[3] == 9 and plan[4] == 1
            if bn in (16, 48) and cout > bn or (bm, bn) in ((64, 32), (64, 16)):
                assert plan[5:7] != (bm, bn)                       # refused: no such kernel
            elif bm:
                assert plan[5:7] == (bm, bn), (plan, bm, bn)
            seen.add(plan[5:])
            assert relerr(from_nhwc(Y, cout), ref) < 2e-6, plan
    finally:
        for name in ("force_bm", "force_bn", "force_splits"):
            be.lib.call("mnk_set_tuning", name.encode(), 0)
    assert len(seen) >= 6


# ---- data-gradient launches that also leave the backward statistics of the norm layer in front (round 4) -----------------------
BNSTATS_CASES = [
    # N, H, W, Cout of the forward conv (= channels of dy), C (= channels of dx = of the norm layer), up, residual, slope
    (4, 16, 16, 24, 64, False, False, 0.0),      # 64x64 tiles, ReLU
    (2, 16, 16, 40, 45, False, True, 0.0),       # 16x16-MFMA kernel (BN = 48), a residual (skip gradient) in the epilogue
    (3, 8, 8, 20, 70, False, False, -1.0),       # 128-wide plan / two N tiles, no activation
    (2, 8, 12, 33, 30, False, False, 0.2),       # 128x32 tile, LeakyReLU, ragged M
    (2, 4, 4, 136, 24, False, False, 0.0),       # split-K launch: the statistics come out of the split reduction
    (2, 8, 8, 24, 40, True, False, 0.0),         # sub-pixel data gradient of an up-sampled convolution (4x4 / stride 2)
]


@pytest.mark.parametrize("case", BNSTATS_CASES)
def test_data_gradient_leaves_the_backward_statistics_of_the_norm_layer_in_front(be, case):
    """mnk_conv3x3_dgrad_bnstats / mnk_conv3x3_up_dgrad_bnstats: dx equals the plain data-gradient launch's, and the finished
    column sums equal mnk_bn_act_bwd_stats's pass over (y, dz = dx) -- and an fp64 evaluation of sum g, sum g * xhat."""
    n, h, w, cout, c, up, use_res, slope = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    ho, wo = (2 * h, 2 * w) if up else (h, w)                       # geometry of dy; (h, w) = geometry of dx and of the norm layer
    dy = torch.randn(n, cout, ho, wo, generator=g)
    wt = torch.randn(cout, c, 3, 3, generator=g) * 0.2
    res = torch.randn(n, c, h, w, generator=g) if use_res else None
    y = torch.randn(n, c, h, w, generator=g)                        # the norm layer's input
    mean, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    scale = gamma * invstd
    DY, Y = be.t(to_nhwc(dy)), be.t(to_nhwc(y))
    ldx = ceil4(c)
    dx_a, dx_b = be.empty(n, h, w, ldx), be.empty(n, h, w, ldx)
    if up:
        wp = be.empty(be.query("mnk_conv3x3_up_dgrad_packed_floats", cout, c))
        be.call("mnk_conv3x3_up_pack_dgrad", be.t(wt), wp, cout, c, 0, c)
        nws = be.query("mnk_conv3x3_up_dgrad_workspace_floats", n, h, w, cout, c)
        nst = be.query("mnk_conv3x3_up_dgrad_stats_floats", n, h, w, cout, c)
    else:
        wp = be.empty(be.query("mnk_conv3x3_packed_floats", c, cout, 0))
        be.call("mnk_conv3x3_pack_dgrad", be.t(wt), wp, cout, c, 0, c)
        nws = be.query("mnk_conv3x3_workspace_floats", n, h, w, cout, 0, c)
        nst = be.query("mnk_conv3x3_stats_floats", n, h, w, cout, 0, c)
    assert nst > 0 and nst % (2 * ldx) == 0
    ws = be.empty(max(nws, 1))
    st = be.empty(nst)
    R = be.t(to_nhwc(res)) if use_res else None
    bn = (Y, ldx, be.t(mean), be.t(invstd), be.t(scale), be.t(beta), float(slope))
    if up:
        be.call("mnk_conv3x3_up_dgrad", DY, DY.shape[-1], cout, wp, dx_a, ldx, n, h, w, c, ws, nws)
        be.call("mnk_conv3x3_up_dgrad_bnstats", DY, DY.shape[-1], cout, wp, dx_b, ldx, n, h, w, c, ws, nws, st, *bn)
    else:
        be.call("mnk_conv3x3_fwd", DY, DY.shape[-1], cout, None, 0, 0, 2, wp, None, R, ldx if use_res else 0, dx_a, ldx, n, h, w,
                c, ws, nws, None)
        be.call("mnk_conv3x3_dgrad_bnstats", DY, DY.shape[-1], cout, wp, R, ldx if use_res else 0, dx_b, ldx, n, h, w, c, ws, nws,
                st, *bn)
    sums = be.empty(2 * c)
    be.call("mnk_bn_stats_finish", st, nst // (2 * ldx), ldx, c, sums)
    # the separate pass the hand-over replaces
    ref_sums = be.empty(2 * c)
    nwb = be.query("mnk_bn_workspace_floats", n * h * w, ldx)
    wsb = be.empty(max(nwb, 1))
    relu = 1 if slope == 0.0 else 0
    if slope in (0.0, -1.0):
        be.call("mnk_bn_act_bwd_stats", Y, ldx, dx_a, ldx, 0, be.t(mean), be.t(invstd), be.t(scale), be.t(beta), n, h, w, c, relu, 0,
                ref_sums, wsb, nwb)
    be.sync()
    assert torch.equal(dx_a.cpu(), dx_b.cpu()), "the data gradient itself must not change"
    dz = from_nhwc(dx_a.cpu(), c).double()
    d = y.double() - mean.double()[None, :, None, None]
    pre = d * scale.double()[None, :, None, None] + beta.double()[None, :, None, None]
    gg = dz if slope < 0 else torch.where(pre > 0, dz, dz * slope)
    want = torch.cat([gg.sum(dim=(0, 2, 3)), (gg * d * invstd.double()[None, :, None, None]).sum(dim=(0, 2, 3))])
    got = sums.cpu().double()
    tol = 2e-5 * (float(want.abs().max()) + float(gg.abs().sum(dim=(0, 2, 3)).max()) * 1e-2)
    assert float((got - want).abs().max()) <= tol, (float((got - want).abs().max()), tol)
    if slope in (0.0, -1.0):
        assert float((got - ref_sums.cpu().double()).abs().max()) <= tol

