"""Deferred weight-gradient reductions (mnk_conv2d_wgrad under MNK_WGRAD_DEFER + mnk_wgrad_reduce_multi: the split
partials of MANY layers reduced in one launch) and the one-launch Adam that also emits the packed convolution weights
(mnk_adam_multi), against torch: conv2d's weight gradient in fp64, torch.optim.Adam's trajectory."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import to_nhwc, ceil4, relerr
from test_kernels_conv import CASES, HALO_CASES, WFAST_CASES, K4_CASES, _inputs


class Plan(ctypes.Structure):
    _fields_ = [("layout", ctypes.c_int), ("splits", ctypes.c_int), ("part_floats", ctypes.c_size_t)]


REDUCE_DESC = np.dtype([("part", "<u8"), ("dw", "<u8"), ("layout", "<i4"), ("splits", "<i4"), ("ntaps", "<i4"),
                        ("Cout", "<i4"), ("C", "<i4"), ("Cin_total", "<i4"), ("c_start", "<i4"), ("accumulate", "<i4"),
                        ("block_begin", "<i4"), ("reserved", "<i4")])
ADAM_DESC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("wp_fwd", "<u8"),
                      ("wp_d0", "<u8"), ("wp_d1", "<u8"), ("Cout", "<i4"), ("C0", "<i4"), ("C1", "<i4"),
                      ("block_begin", "<i4"), ("flags", "<i4"), ("reserved", "<i4"), ("gt0", "<u8"), ("gt1", "<u8"),
                      ("gt_splits0", "<i4"), ("gt_splits1", "<i4")])


def test_descriptor_sizes_match_the_header():
    assert REDUCE_DESC.itemsize == 56 and ADAM_DESC.itemsize == 112


def _table(be, rec):
    return be.t(torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()))


@pytest.mark.parametrize("accumulate", [0, 1])
def test_deferred_wgrad_of_many_layers_reduced_in_one_launch(be, accumulate):
    """every weight-gradient form (tap-major, nine-tap 16x16, LDS-halo, gather; 3x3 and the discriminator's 4x4) leaves its
    partials behind; ONE mnk_wgrad_reduce_multi launch produces all gradients (two-source layers: two descriptors into
    one parameter gradient)."""
    layers = []
    for case in CASES + HALO_CASES + WFAST_CASES:
        n, h, w, c0, c1, cout, ups, _, _ = case
        x0, x1, wt, b, r = _inputs(case)
        layers.append((n, h, w, h, w, c0, c1, cout, ups, 3, 3, 1, x0, x1, wt))
    g = torch.Generator().manual_seed(9)
    for n, hi, wi, cin, cout in K4_CASES:
        layers.append((n, hi - 3, wi - 3, hi, wi, cin, 0, cout, 0, 4, 4, 0, torch.randn(n, cin, hi, wi, generator=g), None,
                       torch.randn(cout, cin, 4, 4, generator=g) * 0.2))
    rows, keep, blocks, direct = [], [], 0, 0
    for li, (n, ho, wo, hi, wi, c0, c1, cout, ups, kh, kw, pad, x0, x1, wt) in enumerate(layers):
        gd = torch.Generator().manual_seed(100 + li)
        dy = torch.randn(n, cout, ho, wo, generator=gd)
        wd = wt.double().requires_grad_(True)
        x = x0 if x1 is None else torch.cat([x0, x1], 1)
        if ups:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        F.conv2d(x.double(), wd, None, padding=pad).backward(dy.double())
        DY = be.t(to_nhwc(dy))
        base = torch.randn(cout, c0 + c1, kh, kw, generator=gd) if accumulate else None
        DW = be.t(base.clone()) if accumulate else be.empty(cout, c0 + c1, kh, kw)
        for src, c_start, c_cnt in ((x0, 0, c0),) + (((x1, c0, c1),) if c1 else ()):
            X = be.t(to_nhwc(src))
            plan = Plan()
            his, wis = (hi // 2, wi // 2) if ups else (hi, wi)
            assert be.query("mnk_conv2d_wgrad_plan2", n, ho, wo, c_cnt, cout, kh, kw, pad, X.shape[-1], int(ups) | 2,
                            ctypes.byref(plan)) == 0
            part = be.empty(max(plan.part_floats, 1))
            tgt = DW if not (plan.splits == 0 and accumulate) else be.empty(cout, c0 + c1, kh, kw)
            be.call("mnk_conv2d_wgrad", X, X.shape[-1], c_cnt, int(ups) | 2 | 4, hi, wi, kh, kw, pad, DY, DY.shape[-1], cout,
                    tgt, c0 + c1, c_start, n, ho, wo, part, plan.part_floats)
            if plan.splits > 0:
                rows.append((part.data_ptr(), DW.data_ptr(), plan.layout, plan.splits, kh * kw, cout, c_cnt, c0 + c1,
                             c_start, accumulate, blocks, 0))
                blocks += be.query("mnk_wgrad_reduce_blocks", plan.splits, cout, c_cnt)
            else:
                direct += 1
                if accumulate:      # the direct form overwrites: the caller adds (rare path, not the kernel's business)
                    be.sync()
                    DW[:, c_start:c_start + c_cnt] = be.t(base)[:, c_start:c_start + c_cnt] + tgt[:, c_start:c_start + c_cnt]
            keep.append((X, part, tgt))
        keep.append((DY, DW, wd.grad, base))
    rec = np.array(rows, dtype=REDUCE_DESC)
    layouts = set(int(r["layout"]) for r in rec)
    assert layouts == {0, 1, 2} and direct > 0 and len(rec) > 12, (layouts, direct, len(rec))
    # both thread maps of the reduction are exercised for every layout: the flat float4 map (few splits, C % 4 == 0: the deep
    # levels) and the tile map
    flat = {(int(r["layout"]), int(r["ntaps"])) for r in rec if r["splits"] < 4 and r["C"] % 4 == 0}
    tiled = {int(r["layout"]) for r in rec if not (r["splits"] < 4 and r["C"] % 4 == 0)}
    assert {l for l, _ in flat} >= {0, 2} and tiled >= {0, 1}, (flat, tiled)
    print('flat map:', sorted(flat), 'tile map layouts:', sorted(tiled))
    descs = _table(be, rec)
    be.call("mnk_wgrad_reduce_multi", descs, len(rec), blocks)
    be.sync()
    for item in keep:
        if len(item) == 4:
            DY, DW, ref, base = item
            want = ref if base is None else ref + base.double()
            assert relerr(DW.cpu(), want) < 2e-6


def _fold16(acc):
    """(16, ...) pseudo-tap sums of the sub-pixel form -> (9, ...) kernel taps (csrc/conv3x3.hip: up_fold)"""
    def pairs(k):
        return ((0, 0 if k == 0 else 1), (1, 1 if k == 2 else 0))
    out = []
    for ky in range(3):
        for kx in range(3):
            v = 0
            for ya, yu in pairs(ky):
                for xb, xv in pairs(kx):
                    v = v + acc[4 * (2 * ya + xb) + 2 * yu + xv]
            out.append(v)
    return torch.stack(out)


@pytest.mark.parametrize("accumulate", [0, 1])
def test_wgrad_reduce_multi_thread_maps_on_synthetic_partials(be, accumulate):
    """mnk_wgrad_reduce_multi alone, on made-up partials: every layout x tap count through BOTH thread maps -- the flat float4
    map (splits < 4, C % 4 == 0) with aligned and unaligned gradient slices, and the tile map (many splits / ragged C)."""
    g = torch.Generator().manual_seed(31)
    rows, keep, blocks = [], [], 0
    for layout, nt in ((0, 9), (0, 16), (1, 9), (1, 16), (2, 9)):
        for splits, cout, c, c_start, cin_total in ((1, 33, 20, 0, 20), (3, 5, 8, 4, 16), (2, 7, 8, 3, 12), (5, 9, 8, 0, 8),
                                                    (2, 6, 10, 0, 10), (40, 4, 12, 0, 12)):
            nin = 16 if layout == 2 else nt
            if layout == 1:
                part = torch.randn(splits, cout, c * nt, generator=g)
                want = part.double().sum(0).view(cout, c, nt)
            else:
                part = torch.randn(splits, nin, cout, c, generator=g)
                acc = part.double().sum(0)                               # (nin, cout, c)
                want = (_fold16(acc) if layout == 2 else acc).permute(1, 2, 0)   # (cout, c, nt)
            base = torch.randn(cout, cin_total, nt, generator=g)
            full = base.double().clone() if accumulate else torch.full((cout, cin_total, nt), float("nan"), dtype=torch.float64)
            full[:, c_start:c_start + c] = (full[:, c_start:c_start + c] if accumulate else 0) + want
            P, DW = be.t(part), be.t(base.clone() if accumulate else torch.full_like(base, float("nan")))
            rows.append((P.data_ptr(), DW.data_ptr(), layout, splits, nt, cout, c, cin_total, c_start, accumulate, blocks, 0))
            blocks += be.query("mnk_wgrad_reduce_blocks", splits, cout, c)
            keep.append((P, DW, full, c_start, c))
    rec = np.array(rows, dtype=REDUCE_DESC)
    be.call("mnk_wgrad_reduce_multi", _table(be, rec), len(rec), blocks)
    be.sync()
    for i, (P, DW, full, c_start, c) in enumerate(keep):
        got = DW.cpu().double()
        sl = slice(c_start, c_start + c)
        assert float((got[:, sl] - full[:, sl]).abs().max()) < 1e-5, (i, rows[i][2:10])
        rest = torch.ones(got.shape[1], dtype=torch.bool)
        rest[sl] = False
        if accumulate:       # columns of other sources are not touched
            assert torch.equal(got[:, rest], full[:, rest]), (i, rows[i][2:10])
        else:
            assert bool(torch.isnan(got[:, rest]).all()), (i, rows[i][2:10])


def _adam_reference(params, grads_seq, lr):
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.Adam(ps, lr=lr, betas=(0.5, 0.999))
    for grads in grads_seq:
        for p, g in zip(ps, grads):
            p.grad = g.clone()
        opt.step()
    return [p.detach() for p in ps], opt


@pytest.mark.parametrize("lr_drop", [False, True])
def test_adam_multi_matches_torch_adam_and_emits_the_packs(be, lr_drop):
    """3-step trajectory of plain ranges (odd sizes, unaligned views) and 3x3 convolution weights (one / two sources, with
    and without data-gradient layouts) == torch.optim.Adam(betas=(0.5, 0.999)); the packed layouts written in the same
    launch == mnk_conv3x3_pack_all of the updated weights; a learning-rate change through the device scalars is seen."""
    g = torch.Generator().manual_seed(21)
    convs = [(21, 18, 7, True, True), (3, 35, 0, True, False), (40, 5, 0, False, False), (70, 33, 0, True, False),
             (21, 18, 7, True, True), (33, 20, 0, True, False)]
    ups = [0, 0, 0, 0, 1, 1]          # the last two are up-sampled convolutions: the packs of their sub-pixel forms
    plains = [1, 5, 1000, 4096, 4097, 12345]
    flat = torch.zeros(sum(plains) + 16)
    params, off = [], 3                                   # offset 3: views that are not 16-byte aligned
    for n in plains:
        params.append(torch.randn(n, generator=g))
        off += n
    for cout, c0, c1, _, _ in convs:
        params.append(torch.randn(cout, c0 + c1, 1, 3, 3, generator=g) * 0.1)
    steps = 3
    grads_seq = [[torch.randn(p.shape, generator=g) * (10.0 ** float(torch.randint(-4, 2, (1,), generator=g)))
                  for p in params] for _ in range(steps)]
    lr = 2e-4
    # device state
    P = [be.t(p.clone()) for p in params]
    M = [be.zeros(*p.shape) for p in params]
    V = [be.zeros(*p.shape) for p in params]
    G = [be.zeros(*p.shape) for p in params]
    rows, blocks, packs = [], 0, []
    for k, p in enumerate(params):
        if k < len(plains):
            nb = be.query("mnk_adam_blocks", p.numel(), 0, 0, 0, 0)
            rows.append((P[k].data_ptr(), G[k].data_ptr(), M[k].data_ptr(), V[k].data_ptr(), p.numel(), 0, 0, 0, 0, 0, 0, blocks, 0, 0, 0, 0, 0, 0))
            packs.append(None)
        else:
            cout, c0, c1, d0, d1 = convs[k - len(plains)]
            up = ups[k - len(plains)]
            if up:
                wf = be.empty(be.query("mnk_conv3x3_up_packed_floats", cout, c0, c1))
                w0 = be.empty(be.query("mnk_conv3x3_up_dgrad_packed_floats", cout, c0)) if d0 else None
                w1 = be.empty(be.query("mnk_conv3x3_up_dgrad_packed_floats", cout, c1)) if d1 and c1 else None
            else:
                wf = be.empty(be.query("mnk_conv3x3_packed_floats", cout, c0, c1))
                w0 = be.empty(be.query("mnk_conv3x3_packed_floats", c0, cout, 0)) if d0 else None
                w1 = be.empty(be.query("mnk_conv3x3_packed_floats", c1, cout, 0)) if d1 and c1 else None
            nb = be.query("mnk_adam_blocks", 0, cout, c0, c1, 1)
            rows.append((P[k].data_ptr(), G[k].data_ptr(), M[k].data_ptr(), V[k].data_ptr(), p.numel(), wf.data_ptr(),
                         w0.data_ptr() if w0 is not None else 0, w1.data_ptr() if w1 is not None else 0, cout, c0, c1, blocks,
                         up, 0, 0, 0, 0, 0))
            packs.append((wf, w0, w1))
        assert nb > 0
        blocks += nb
    descs = _table(be, np.array(rows, dtype=ADAM_DESC))
    hyper = be.t(torch.tensor([lr, 0.5, 0.999, 1e-8, 0.0, 0.0, 1.0, 0.0, 1 - 0.5, 1 - 0.999], dtype=torch.float64).float())
    ref_ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.Adam(ref_ps, lr=lr, betas=(0.5, 0.999))
    for it, grads in enumerate(grads_seq):
        if lr_drop and it == 2:
            lr *= 0.1
            opt.param_groups[0]["lr"] = lr
            hyper[0:1].copy_(be.t(torch.tensor([lr])))
        for k, gr in enumerate(grads):
            G[k].copy_(be.t(gr))
            ref_ps[k].grad = gr.clone()
        opt.step()
        be.call("mnk_adam_tick", hyper)
        be.call("mnk_adam_multi", descs, len(rows), blocks, hyper)
        be.sync()
        for k in range(len(params)):
            scale = float(ref_ps[k].detach().abs().max()) + 1e-12
            assert float((P[k].cpu() - ref_ps[k].detach()).abs().max()) <= 2e-6 * scale + 4e-7 * lr / 2e-4 * (it + 1), (it, k)
            st = opt.state[ref_ps[k]]
            assert relerr(M[k].cpu(), st["exp_avg"]) < 1e-5 and relerr(V[k].cpu(), st["exp_avg_sq"]) < 1e-5
    assert float(hyper.cpu()[7]) == steps
    for k, pk in enumerate(packs):
        if pk is None:
            continue
        cout, c0, c1, d0, d1 = convs[k - len(plains)]
        wf, w0, w1 = pk
        rf = be.empty(wf.numel())
        r0 = be.empty(w0.numel()) if w0 is not None else None
        r1 = be.empty(w1.numel()) if w1 is not None else None
        if ups[k - len(plains)]:
            be.call("mnk_conv3x3_up_pack_fwd", P[k], rf, cout, c0, c1)
            if r0 is not None:
                be.call("mnk_conv3x3_up_pack_dgrad", P[k], r0, cout, c0 + c1, 0, c0)
            if r1 is not None:
                be.call("mnk_conv3x3_up_pack_dgrad", P[k], r1, cout, c0 + c1, c0, c1)
        else:
            be.call("mnk_conv3x3_pack_all", P[k], rf, r0, r1, cout, c0, c1)
        be.sync()
        for a, b in ((wf, rf), (w0, r0), (w1, r1)):
            if a is not None:
                assert float((a.cpu().nan_to_num(nan=7.0) - b.cpu().nan_to_num(nan=7.0)).abs().max()) <= \
                    1e-6 * float(b.cpu().nan_to_num(nan=7.0).abs().max())


JOB = np.dtype([("x", "<u8"), ("dy", "<u8"), ("part", "<u8"), ("part_floats", "<u8"), ("ld_x", "<i4"), ("C", "<i4"),
                ("flags", "<i4"), ("ld_dy", "<i4"), ("Cout", "<i4"), ("N", "<i4"), ("Ho", "<i4"), ("Wo", "<i4"), ("Hi", "<i4"),
                ("Wi", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("pad", "<i4"), ("variant", "<i4"), ("splits", "<i4"),
                ("reserved", "<i4")])


@pytest.mark.parametrize("subpixel", [1, 0], ids=["up-layers-subpixel", "up-layers-upsampled-view"])
def test_grouped_weight_gradients_of_many_layers(be, subpixel):
    try:
        be.lib.call("mnk_set_tuning", b"up_subpixel", subpixel)
        _grouped_weight_gradients(be, subpixel)
    finally:
        be.lib.call("mnk_set_tuning", b"up_subpixel", 1)


def _grouped_weight_gradients(be, subpixel):
    """mnk_wgrad_grouped_*: the tap-major weight-gradient GEMMs of many layers (all four tile shapes, the three loaders, two
    sources, up-sampled views, the 4x4 discriminator kernels, several pixel chunks per layer) in one launch per tile
    shape + ONE reduction launch, against conv2d's weight gradient in fp64."""
    assert JOB.itemsize == 96
    layers = []
    for case in CASES + HALO_CASES + WFAST_CASES:
        n, h, w, c0, c1, cout, ups, _, _ = case
        x0, x1, wt, b, r = _inputs(case)
        layers.append((n, h, w, h, w, c0, c1, cout, ups, 3, 3, 1, x0, x1, wt))
    g = torch.Generator().manual_seed(9)
    for n, hi, wi, cin, cout in K4_CASES:
        layers.append((n, hi - 3, wi - 3, hi, wi, cin, 0, cout, 0, 4, 4, 0, torch.randn(n, cin, hi, wi, generator=g), None,
                       torch.randn(cout, cin, 4, 4, generator=g) * 0.2))
    # a long pixel range: several chunks of the default 1024 pixels
    gl = torch.Generator().manual_seed(10)
    layers.append((3, 32, 48, 32, 48, 70, 0, 24, 0, 3, 3, 1, torch.randn(3, 70, 32, 48, generator=gl), None,
                   torch.randn(24, 70, 3, 3, generator=gl) * 0.2))
    # narrow layers on 8-aligned maps: the nine-tap 16x16 kernel, grouped (variants >= 16), plain and up-sampled
    layers.append((4, 16, 24, 16, 24, 20, 0, 45, 0, 3, 3, 1, torch.randn(4, 20, 16, 24, generator=gl), None,
                   torch.randn(45, 20, 3, 3, generator=gl) * 0.2))
    layers.append((2, 16, 16, 16, 16, 45, 0, 45, 0, 3, 3, 1, torch.randn(2, 45, 16, 16, generator=gl), None,
                   torch.randn(45, 45, 3, 3, generator=gl) * 0.2))
    jobs, meta, keep = [], [], []
    for li, (n, ho, wo, hi, wi, c0, c1, cout, ups, kh, kw, pad, x0, x1, wt) in enumerate(layers):
        gd = torch.Generator().manual_seed(100 + li)
        dy = torch.randn(n, cout, ho, wo, generator=gd)
        wd = wt.double().requires_grad_(True)
        x = x0 if x1 is None else torch.cat([x0, x1], 1)
        if ups:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        F.conv2d(x.double(), wd, None, padding=pad).backward(dy.double())
        DY = be.t(to_nhwc(dy))
        DW = be.empty(cout, c0 + c1, kh, kw)
        keep.append((DY, DW, wd.grad))
        for src, c_start, c_cnt in ((x0, 0, c0),) + (((x1, c0, c1),) if c1 else ()):
            X = be.t(to_nhwc(src))
            clean = 2 if li % 2 == 0 else 0            # both loader families
            jobs.append((X.data_ptr(), DY.data_ptr(), 0, 0, X.shape[-1], c_cnt, int(ups) | clean, DY.shape[-1], cout, n, ho, wo,
                         hi, wi, kh, kw, pad, 0, 0, 0))
            meta.append((DW, c0 + c1, c_start, c_cnt, cout, kh * kw, X))
    rec = np.array(jobs, dtype=JOB)
    assert be.query("mnk_wgrad_grouped_plan", rec.ctypes.data, len(rec)) == 0
    sel = [i for i in range(len(rec)) if rec["variant"][i] >= 0]
    tap = [i for i in sel if rec["variant"][i] < 16]
    n16 = [i for i in sel if rec["variant"][i] >= 16]
    assert len(tap) >= 12 and len(set(int(rec["variant"][i]) // 4 for i in tap)) == 4, rec["variant"]
    assert len(set(int(rec["variant"][i]) for i in n16)) >= 2 and all(int(rec["splits"][i]) > 1 for i in n16), rec["variant"]
    modes = set(int(rec["variant"][i]) % 4 for i in tap)       # loaders: generic, 3x3 buffer loads, + up-sampled view / sub-pixel
    assert modes == ({0, 1, 3} if subpixel else {0, 1, 2}) and max(int(rec["splits"][i]) for i in tap) >= 4
    grouped = rec[sel].copy()
    parts, rows, blocks = [], [], 0
    for k, i in enumerate(sel):
        part = be.empty(int(grouped["part_floats"][k]))
        parts.append(part)
        grouped["part"][k] = part.data_ptr()
        DW, cin_total, c_start, c_cnt, cout, ntaps, _ = meta[i]
        v = int(grouped["variant"][k])
        rows.append((part.data_ptr(), DW.data_ptr(), 2 if (v < 16 and v % 4 == 3) else 0, int(grouped["splits"][k]),
                     ntaps, cout, c_cnt, cin_total, c_start, 0, blocks, 0))
        blocks += be.query("mnk_wgrad_reduce_blocks", int(grouped["splits"][k]), cout, c_cnt)
    nbytes = be.query("mnk_wgrad_grouped_table_bytes", len(grouped))
    host = torch.zeros(nbytes, dtype=torch.uint8)
    assert be.query("mnk_wgrad_grouped_build", grouped.ctypes.data, len(grouped), host.data_ptr(), nbytes) == 0
    dev = be.t(host)
    be.call("mnk_wgrad_grouped_launch", dev, host.data_ptr())
    # the layers that are not tap-major shapes go through the single-layer entry (deferred reduction, as before)
    for i in range(len(rec)):
        if i in sel:
            continue
        DW, cin_total, c_start, c_cnt, cout, ntaps, X = meta[i]
        j = rec[i]
        plan = Plan()
        assert be.query("mnk_conv2d_wgrad_plan2", int(j["N"]), int(j["Ho"]), int(j["Wo"]), c_cnt, cout, int(j["kh"]), int(j["kw"]),
                        int(j["pad"]), int(j["ld_x"]), int(j["flags"]) | 2, ctypes.byref(plan)) == 0
        part = be.empty(max(plan.part_floats, 1))
        parts.append(part)
        be.lib.call("mnk_conv2d_wgrad", int(j["x"]), int(j["ld_x"]), c_cnt, int(j["flags"]) | 2 | 4, int(j["Hi"]), int(j["Wi"]),
                    int(j["kh"]), int(j["kw"]), int(j["pad"]), int(j["dy"]), int(j["ld_dy"]), cout, DW.data_ptr(), cin_total,
                    c_start, int(j["N"]), int(j["Ho"]), int(j["Wo"]), part.data_ptr(), plan.part_floats, be.stream())
        if plan.splits > 0:
            rows.append((part.data_ptr(), DW.data_ptr(), plan.layout, plan.splits, ntaps, cout, c_cnt, cin_total, c_start, 0,
                         blocks, 0))
            blocks += be.query("mnk_wgrad_reduce_blocks", plan.splits, cout, c_cnt)
    descs = _table(be, np.array(rows, dtype=REDUCE_DESC))
    be.call("mnk_wgrad_reduce_multi", descs, len(rows), blocks)
    be.sync()
    for li, (DY, DW, ref) in enumerate(keep):
        assert relerr(DW.cpu(), ref) < 2e-6, (li, layers[li][:12], relerr(DW.cpu(), ref))


def test_table_upload_through_kernel_arguments(be):
    """mnk_table_upload: a host table reaches device memory as kernel arguments (3.5 KB per launch, several launches)."""
    g = torch.Generator().manual_seed(1)
    for nbytes in (16, 3584, 3600, 12000):
        host = torch.randint(0, 256, (nbytes,), generator=g, dtype=torch.uint8).numpy()
        dev = be.zeros((nbytes + 15) // 16 * 16 // 4).view(torch.uint8) if be.kind == "emu" else \
            torch.zeros((nbytes + 15) // 16 * 16, dtype=torch.uint8, device=be.device)
        be.lib.call("mnk_table_upload", host.ctypes.data, dev.data_ptr(), nbytes, be.stream())
        be.sync()
        assert bytes(dev.cpu().numpy()[:nbytes]) == bytes(host)


@pytest.mark.parametrize("batch,twice", [(2, False), (5, False), (2, True)])
def test_adam_reading_tap_major_partials_equals_reduce_then_adam(be, batch, twice):
    """MnkAdam.tap_direct (what a captured iteration of one process uses): the step kernel takes the gradients of the few-split
    tap-major layers -- plain, sub-pixel (16 pseudo taps folded) and second-source ones -- straight from the partials of the
    grouped weight-gradient GEMMs, and mnk_wgrad_reduce_multi skips them.  Same sums in the same order: the parameters after
    two steps are the reduce-then-Adam ones to the bit."""
    from modules.util import Hourglass
    from mnk import optim as moptim

    def run(direct):
        torch.manual_seed(3)
        hg = Hourglass(block_expansion=64, in_features=3, out_features=4, num_blocks=2, max_features=256).to(be.device)
        hg.train()
        opt = moptim.MnkAdam(hg.parameters(), lr=1e-3, betas=(0.5, 0.999))
        opt.tap_direct = direct
        g = torch.Generator().manual_seed(4)
        x = be.t(torch.rand(batch, 3, 1, 16, 16, generator=g))
        ndirect, most = 0, 0
        for _ in range(2):
            out = hg(x)
            (out * out).mean().backward()
            opt.materialize_grads()          # (as mnk.engine.TrainStep does before it steps: a second flush finds nothing pending)
            ndirect = max(ndirect, len(opt.reducer.direct))
            most = max([most] + [opt.reducer.recs[k]["splits"] for k in opt.reducer.direct])
            if twice:                        # a second backward pass before the step: the slow path adds to the sinks, so the
                (hg(x * 0.5).sum() * 1e-3).backward()        # first contribution must be reduced into them after all
            opt.step()
            opt.zero_grad()
        be.sync()
        return [p.detach().cpu().clone() for p in hg.parameters()], ndirect, most

    base, n0, _ = run(False)
    got, n1, most = run(True)
    assert n0 == 0 and n1 >= 3, (n0, n1)          # plain, up-sampled and two-source layers took the direct path
    assert most >= (2 if batch == 5 else 1)       # ... batch 5: with more than one split to add up
    for a, b in zip(base, got):
        assert torch.equal(a, b)
