#!/usr/bin/env python
"""Where the HOST time of the drop-in loop goes (bench.py's `dropin` object: the reference's own train.py loop on the drop-in
modules, eager launches): cProfile over a few iterations, top functions by own and by cumulative time.
Usage (GPU box): python tools/dropin_profile.py [--config moving-gif] [--batch 32] [--iters 10]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts monkey-net_amd on sys.path)
import torch  # noqa: E402
from mnk import configs, workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="moving-gif")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--mnk-adam", type=int, default=0, help="1: the loop with mnk.optim.MnkAdam in place of torch.optim.Adam")
    a = ap.parse_args()
    cfg = configs.get(a.config)
    dev = torch.device("cuda:0")
    src, drv = workload.synthetic_pair(a.batch, a.size, a.size)
    x = {"source": src.to(dev), "video": drv.to(dev)}
    r = bench.dropin_loop(cfg, x, dev, a.iters, 3, mnk_adam=bool(a.mnk_adam))
    print("unprofiled: %.3f ms per iteration" % r["ms_per_step"])
    pr = cProfile.Profile()
    pr.enable()
    r = bench.dropin_loop(cfg, x, dev, a.iters, 3, mnk_adam=bool(a.mnk_adam))
    pr.disable()
    print("under cProfile: %.3f ms per iteration (%d timed + 3 warm-up iterations + construction in the profile)" % (r["ms_per_step"], a.iters))
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
        print("==== by %s\n%s" % (key, s.getvalue()[:9000]))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().print_callers("split_with_sizes|method 'cpu'|method 'to' of")
    print("==== callers\n%s" % s.getvalue()[:6000])


if __name__ == "__main__":
    main()
