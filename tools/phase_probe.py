#!/usr/bin/env python
"""Where a forward / data-gradient GEMM launch spends its time, block by block (needs the -DMNK_PHASE_CLOCKS experiment build:
MNK_BUILD_TAG=_clk MNK_EXTRA_FLAGS=-DMNK_PHASE_CLOCKS monkey-net_amd/csrc/build.sh; run with
MNK_LIBRARY=monkey-net_amd/libmonkeynet_hip_clk.so).  Thread 0 of every block of conv3x3_igemm_kernel stamps the 100 MHz wall
clock at entry, in front of its K loop, behind it and at its exit (after its stores have left the wave).  Printed per layer:
when blocks start (relative to the first), how long prologue / loop / epilogue take, when the last block ends, against the
launch duration that HIP events see for back-to-back launches of the same kernel.
Usage (GPU box): python tools/phase_probe.py [--config moving-gif] [--batch 32]"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mnk import configs, ops, _lib, workload  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
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
    ap.add_argument("--config", default="moving-gif")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    args = ap.parse_args()
    lib = _lib.lib()
    assert hasattr(lib.cdll, "mnk_phase_log_read"), "needs the -DMNK_PHASE_CLOCKS build (see the docstring)"
    cfg = configs.get(args.config)
    layers = workload.conv_flops_hot_path(cfg, args.size, args.size)["layers"]
    dev = torch.device("cuda:0")
    log = np.zeros(4 * 16384, dtype=np.uint64)
    sclk = np.zeros(2 * 16384, dtype=np.uint64)
    print("%-14s %5s %5s %3s | %6s blocks | event us | start p50 / max | prologue p50 | loop p50 (min..max) | epilogue p50 | "
          "last end | first end" % ("layer", "cin", "cout", "hw", ""))
    seen = set()
    for name, cin, cout, h, w, k, flops in layers:
        if k != 3 or ".dec" in name:
            continue
        frames = args.batch * (2 if name.startswith("kp") else 1)
        key = (cin, cout, h, w, frames)
        if key in seen or cout <= 48:
            continue
        seen.add(key)
        x = torch.randn(frames, h, w, ops.ceil4(cin), device=dev)
        wt = torch.randn(cout, cin, 1, 3, 3, device=dev) * 0.05
        bias = torch.randn(cout, device=dev)
        wp = ops._packed_fwd_weight(wt, cout, cin, 0, False)
        fn = lambda: ops._conv_launch(x, cin, None, 0, False, wp, bias, None, frames, h, w, cout, True, False)
        t = timeit(fn)
        torch.cuda.synchronize()
        lib.cdll.mnk_phase_log_read(log.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(log.nbytes), 1)
        fn()
        torch.cuda.synchronize()
        lib.cdll.mnk_phase_sclk_read(sclk.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(sclk.nbytes))
        lib.cdll.mnk_phase_log_read(log.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(log.nbytes), 1)
        sc = sclk.reshape(-1, 2)[log.reshape(-1, 4)[:, 0] > 0].astype(np.int64)
        st = log.reshape(-1, 4)
        st = st[st[:, 0] > 0].astype(np.int64)
        if not len(st):
            print("%-14s no stamps (another kernel took this shape)" % name)
            continue
        t0 = st[:, 0].min()
        us = lambda v: v * 0.01            # 100 MHz ticks
        start = us(st[:, 0] - t0)
        pro = us(st[:, 1] - st[:, 0])
        loop = us(st[:, 2] - st[:, 1])
        epi = us(st[:, 3] - st[:, 2])
        end = us(st[:, 3] - t0)
        # shader clock during the K loop: s_memtime ticks per 100 MHz wall-clock tick
        ghz = np.median((sc[:, 1] - sc[:, 0]) / np.maximum(st[:, 2] - st[:, 1], 1)) * 0.1
        ksteps = 9 * ((cin + 15) // 16)
        print("%-14s %5d %5d %3d | %6d blocks | %8.1f | %6.1f / %6.1f | %8.1f | %6.1f (%5.1f..%5.1f) | %8.1f | %7.1f | %7.1f | "
              "s_memtime/wall in the loop: %.3f GHz-equivalent, %d K steps"
              % (name, cin, cout, h, len(st), t, np.median(start), start.max(), np.median(pro), np.median(loop), loop.min(),
                 loop.max(), np.median(epi), end.max(), end.min(), ghz, ksteps))


if __name__ == "__main__":
    main()
