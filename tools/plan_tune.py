#!/usr/bin/env python
"""Sweep of the forward / data-gradient GEMM launch plans (block tile x split-K) over the 3x3 layer shapes of a config, on the
GPU box.  For every distinct layer shape and direction each candidate plan is forced (mnk_set_tuning force_*), the launch is
timed cold (a cache-sized buffer is rewritten before every timed launch) together with what the plan costs downstream -- the
split-K reduction, and for a layer in front of a BatchNorm the statistics pass a split plan needs because its epilogue cannot
produce the sums -- and the best plan is compared with make_plan's rule.  Rows that beat the rule by more than --gain go to
stdout in csrc/plan_table.h's format.
Usage:  python tools/plan_tune.py --config moving-gif --batch 32 [--size 64] > gpurun_out/plan_tune_moving-gif.txt"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import torch  # noqa: E402

from mnk import configs, ops, _lib, workload  # noqa: E402

FLUSH = None


def force(bm, bn, splits):
    lib = _lib.lib()
    lib.call("mnk_set_tuning", b"force_bm", bm)
    lib.call("mnk_set_tuning", b"force_bn", bn)
    lib.call("mnk_set_tuning", b"force_splits", splits)


def last_plan():
    out = np.zeros(8, dtype=np.int64)
    _lib.lib().call("mnk_last_plan", out.ctypes.data)
    return tuple(int(v) for v in out)


def time_cold(fn, iters, reps=3):
    """median over reps of the mean time of `iters` launches, each after the caches were overwritten"""
    res = []
    for _ in range(reps):
        fn()
        tot = 0.0
        evs = []
        for _ in range(iters):
            FLUSH.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        tot = sum(a.elapsed_time(b) for a, b in evs)
        res.append(tot / iters * 1e-3)
    return sorted(res)[len(res) // 2]


def candidates(m_tiles128, cout, ksteps, phases):
    bns = [128, 64] if cout > 64 else ([64, 32] + ([48] if cout <= 48 and phases == 1 else []) if cout > 32 else
                                       [32] + ([16] if cout <= 16 and phases == 1 else []))
    out = []
    for bn in bns:
        for bm in ((64, 128) if bn in (64, 128) else (128,)):
            tiles = m_tiles128 * (128 // bm) * ((cout + bn - 1) // bn) * phases
            for s in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64):
                if s > 1 and (tiles >= 768 or tiles * s > 3072 or ksteps // s < 3):
                    continue
                out.append((bm, bn, s))
    return out


def main():
    global FLUSH
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="moving-gif")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--gain", type=float, default=0.03)
    ap.add_argument("--eval", action="store_true",
                    help="evaluation-mode shapes: the key-point detector sees `batch` frames (not source + driving), no BatchNorm "
                         "statistics are asked of the convolutions, forward only (reconstruction.py:45-62 at batch 1)")
    args = ap.parse_args()
    cfg = configs.get(args.config)
    layers = workload.conv_flops_hot_path(cfg, args.size, args.size)["layers"]
    dev = torch.device("cuda:0")
    FLUSH = torch.empty(320 << 18, device=dev)           # 320 MB > the 256 MB memory-side cache
    seen = set()
    rows = {}
    print("# %s batch %d @ %d: cold launch times; rule = make_plan's choice" % (args.config, args.batch, args.size))
    for name, cin, cout, h, w, k, flops in layers:
        if k != 3:
            continue
        frames = args.batch * (2 if name.startswith("kp") and not args.eval else 1)
        ups = ".dec" in name
        key = (cin, cout, h, w, frames, ups)
        if key in seen:
            continue
        seen.add(key)
        hs, ws_ = (h // 2, w // 2) if ups else (h, w)
        x = torch.randn(frames, hs, ws_, ops.ceil4(cin), device=dev)
        wt = torch.randn(cout, cin, 1, 3, 3, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        dy = torch.randn(frames, h, w, ops.ceil4(cout), device=dev)
        up = ops.subpixel(ups)
        wp = ops._packed_fwd_weight(wt, cout, cin, 0, up)
        want_stats = not name.endswith(".last") and not args.eval
        small = want_stats and ops.small_bn(frames * h * w)

        def fwd():
            y, sums = ops._conv_launch(x, cin, None, 0, ups, wp, bias, None, frames, h, w, cout, want_stats, up)
            if want_stats and sums is not None and sums.numel() == 0:
                ops.channel_sums(y, cout)          # a split plan leaves the statistics to a pass over y

        if up:
            wpd = torch.empty(ops._query("mnk_conv3x3_up_dgrad_packed_floats", cout, cin), device=dev)
            ops._call("mnk_conv3x3_up_pack_dgrad", dy, wt.data_ptr(), wpd.data_ptr(), cout, cin, 0, cin)
            dxl = torch.empty(frames, hs, ws_, ops.ceil4(cin), device=dev)

            def dgrad():
                nwd = ops._query("mnk_conv3x3_up_dgrad_workspace_floats", frames, hs, ws_, cout, cin)
                wsd = ops.SCRATCH.get("ws", nwd, dy) if nwd else None
                ops._call("mnk_conv3x3_up_dgrad", dy, dy.data_ptr(), dy.shape[-1], cout, wpd.data_ptr(), dxl.data_ptr(),
                          dxl.shape[-1], frames, hs, ws_, cin, wsd.data_ptr() if nwd else 0, nwd)
        else:
            wpd = torch.empty(ops._query("mnk_conv3x3_packed_floats", cin, cout, 0), device=dev)
            ops._call("mnk_conv3x3_pack_dgrad", dy, wt.data_ptr(), wpd.data_ptr(), cout, cin, 0, cin)

            def dgrad():
                ops._conv_launch(dy, cout, None, 0, 0, wpd, None, None, frames, h, w, cin)

        for direction, fn, n_out in (("fwd", fwd, cout), ("dgrad", dgrad, cin)):
            if (direction == "fwd" and small) or (args.eval and direction == "dgrad"):
                continue                      # the one-launch BatchNorm sums these layers' split partials itself
            force(0, 0, 0)
            _lib.lib().call("mnk_set_tuning", b"plan_table", 0)
            t_rule = time_cold(fn, args.iters)
            rule = last_plan()
            m, co, chunks, taps, phases = rule[:5]
            best, t_best = rule[5:], t_rule
            tried = {rule[5:]: t_rule}
            for bm, bn, s in candidates((m + 127) // 128, co, taps * chunks, phases):
                force(bm, bn, s)
                fn()
                eff = last_plan()[5:]
                if eff in tried:
                    continue
                tried[eff] = time_cold(fn, args.iters)
                if tried[eff] < t_best:
                    best, t_best = eff, tried[eff]
            force(0, 0, 0)
            gain = 1.0 - t_best / t_rule
            top = sorted(tried.items(), key=lambda kv: kv[1])[:4]
            print("# %-14s %-5s M=%-7d Cout=%-4d chunks=%-3d taps=%-2d ph=%d  rule %s %.1f us | best %s %.1f us (%+.1f%%) | %s" % (
                name, direction, m, co, chunks, taps, phases, rule[5:], t_rule * 1e6, best, t_best * 1e6, -gain * 100,
                " ".join("%s=%.1f" % (kk, v * 1e6) for kk, v in top)))
            sys.stdout.flush()
            if best != rule[5:] and gain > args.gain:
                pk = (m, co, chunks, taps, phases)
                if pk in rows and rows[pk][0] >= t_rule - t_best:
                    print("# (same plan key as an earlier row with a larger gain: kept that one)")
                    continue
                rows[pk] = (t_rule - t_best, "    {%d, %d, %d, %d, %d, %d, %d, %d},   // %s %s: %.1f -> %.1f us" % (
                    m, co, chunks, taps, phases, best[0], best[1], best[2], name, direction, t_rule * 1e6, t_best * 1e6))
    print("# rows: %d, summed gain %.1f us (layers counted once per distinct shape)" % (
        len(rows), sum(v[0] for v in rows.values()) * 1e6))
    for v in rows.values():
        print(v[1])


if __name__ == "__main__":
    main()
