#!/usr/bin/env python
"""Fixed cost vs per-K-step cost of the forward implicit GEMM: the same layer (pixels, Cout) with growing Cin.
Usage (GPU box): python tools/kstep_probe.py [--cout 45] [--hw 64] [--frames 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import torch  # noqa: E402

from mnk import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cout", type=int, default=45)
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--frames", type=int, default=32)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, h, w, cout = args.frames, args.hw, args.hw, args.cout
    rows = []
    for cin in (16, 32, 48, 96, 192, 384):
        x = torch.randn(n, h, w, ops.ceil4(cin), device=dev)
        wt = torch.randn(cout, cin, 1, 3, 3, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        wp = ops._packed_fwd_weight(wt, cout, cin, 0, False)
        t = timeit(lambda: ops._conv_launch(x, cin, None, 0, False, wp, bias, None, n, h, w, cout, False, False))
        steps = 9 * ((cin + 15) // 16)
        fl = 2.0 * 9 * cin * cout * n * h * w
        rows.append((cin, steps, t, fl / t / 1e6))
        print("cin %4d  K steps %4d  %7.1f us  %6.1f TFLOP/s" % (cin, steps, t, fl / t / 1e6))
    (c0, s0, t0, _), (c1, s1, t1, _) = rows[2], rows[-1]
    b = (t1 - t0) / (s1 - s0)
    print("marginal %.3f us per K step, fixed %.1f us (from the %d- and %d-step points)" % (b, t0 - b * s0, s0, s1))


if __name__ == "__main__":
    main()
