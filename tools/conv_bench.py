#!/usr/bin/env python
"""Per-layer micro-benchmark of the conv3x3 kernels (forward / dgrad / wgrad) on the layer shapes of one config.
Usage on the GPU box:  python tools/conv_bench.py --config taichi --batch 32 [--size 64]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import torch  # noqa: E402

from mnk import configs, ops, _lib, workload  # noqa: E402


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


STATS = bool(int(os.environ.get('CB_STATS', '0')))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="taichi")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    cfg = configs.get(args.config)
    layers = workload.conv_flops_hot_path(cfg, args.size, args.size)["layers"]
    dev = torch.device("cuda:0")
    tot = {"fwd": [0.0, 0.0, 0.0], "dgrad": [0.0, 0.0, 0.0], "wgrad": [0.0, 0.0, 0.0]}
    print("algorithmic TFLOP/s (the reference's 3x3 convolution, whatever form computes it) | executed TFLOP/s (multiply-adds issued:\n"
          "the sub-pixel forms of the up-sampled layers run 4/9 of them) -- the executed column is what the 157.3 TFLOP/s pipe does")
    print("%-16s %5s %5s %4s %7s | %8s %8s %8s  (alg. TFLOP/s, ms) | executed: %6s %6s %6s" % (
        "layer", "cin", "cout", "hw", "frames", "fwd", "dgrad", "wgrad", "fwd", "dgrad", "wgrad"))
    seen = {}
    for name, cin, cout, h, w, k, flops in layers:
        if k != 3:
            continue
        frames = args.batch * (2 if name.startswith("kp") else 1)
        ups = ".dec" in name
        key = (cin, cout, h, w, frames, ups)
        if key not in seen:
            hs, ws_ = (h // 2, w // 2) if ups else (h, w)
            x = torch.randn(frames, hs, ws_, ops.ceil4(cin), device=dev)
            wt = torch.randn(cout, cin, 1, 3, 3, device=dev) * 0.05
            bias = torch.randn(cout, device=dev)
            dy = torch.randn(frames, h, w, ops.ceil4(cout), device=dev)
            fl = 2.0 * 9 * cin * cout * h * w * frames
            up = ops.subpixel(ups)          # UpBlock3D layers: the sub-pixel forms (MNK_UP_SUBPIXEL=0: the up-sampled view)
            wp = ops._packed_fwd_weight(wt, cout, cin, 0, up)
            t_f = timeit(lambda: ops._conv_launch(x, cin, None, 0, ups, wp, bias, None, frames, h, w, cout, STATS, up), args.iters)
            if up:
                wpd = torch.empty(ops._query("mnk_conv3x3_up_dgrad_packed_floats", cout, cin), device=dev)
                ops._call("mnk_conv3x3_up_pack_dgrad", dy, wt.data_ptr(), wpd.data_ptr(), cout, cin, 0, cin)
                dxl = torch.empty(frames, hs, ws_, ops.ceil4(cin), device=dev)
                nwd = ops._query("mnk_conv3x3_up_dgrad_workspace_floats", frames, hs, ws_, cout, cin)
                wsd = torch.empty(max(nwd, 1), device=dev)
                t_d = timeit(lambda: ops._call("mnk_conv3x3_up_dgrad", dy, dy.data_ptr(), dy.shape[-1], cout, wpd.data_ptr(),
                                               dxl.data_ptr(), dxl.shape[-1], frames, hs, ws_, cin, wsd.data_ptr(), nwd), args.iters)
            else:
                npk = ops._query("mnk_conv3x3_packed_floats", cin, cout, 0)
                wpd = torch.empty(npk, device=dev)
                ops._call("mnk_conv3x3_pack_dgrad", dy, wt.data_ptr(), wpd.data_ptr(), cout, cin, 0, cin)
                t_d = timeit(lambda: ops._conv_launch(dy, cout, None, 0, 0, wpd, None, None, frames, h, w, cin), args.iters)
            dw = torch.empty_like(wt)
            nws = ops._query("mnk_conv3x3_up_wgrad_workspace_floats" if ups else "mnk_conv3x3_wgrad_workspace_floats", frames, h, w,
                             cin, cout)
            ws = torch.empty(max(nws, 1), device=dev)

            def wg():
                ops._call("mnk_conv3x3_wgrad", dy, x.data_ptr(), x.shape[-1], cin, int(ups) | 2, dy.data_ptr(), dy.shape[-1], cout,
                          dw.data_ptr(), cin, 0, frames, h, w, ws.data_ptr(), nws)
            t_w = timeit(wg, args.iters)
            seen[key] = (fl, t_f, t_d, t_w, fl * (4.0 / 9.0 if up else 1.0))
        fl, t_f, t_d, t_w, ex = seen[key]
        print("%-16s %5d %5d %4d %7d | %5.1f %5.2f  %5.1f %5.2f  %5.1f %5.2f | %15.1f %6.1f %6.1f" % (
            name, cin, cout, h, frames, fl / t_f / 1e12, t_f * 1e3, fl / t_d / 1e12, t_d * 1e3, fl / t_w / 1e12, t_w * 1e3,
            ex / t_f / 1e12, ex / t_d / 1e12, ex / t_w / 1e12))
        for kk, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
            tot[kk][0] += fl
            tot[kk][1] += t
            tot[kk][2] += ex
    for kk, (fl, t, ex) in tot.items():
        print("TOTAL %-6s %.1f GFLOP in %.2f ms = %.1f TFLOP/s (%.1f%% of 157.3) algorithmic; executed %.1f GFLOP = %.1f TFLOP/s "
              "(%.1f%% of 157.3)" % (kk, fl / 1e9, t * 1e3, fl / t / 1e12, fl / t / 1e12 / 157.3 * 100, ex / 1e9, ex / t / 1e12,
                                     ex / t / 1e12 / 157.3 * 100))


if __name__ == "__main__":
    main()
