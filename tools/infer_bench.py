#!/usr/bin/env python
"""BASELINE config 5: bair 64x64 reconstruction-mode inference, batch 512, hipGraph-captured forward (kp detector on
source + driving frames, generator).  Prints frames/s for eager launches and for graph replay."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import torch
from mnk import configs, engine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="bair"); ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--size", type=int, default=64); ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
gen, disc, kpd = bench.build_models(configs.get(args.config), dev)
src = torch.rand(args.batch, 3, 1, args.size, args.size, device=dev)
drv = torch.rand(args.batch, 3, 1, args.size, args.size, device=dev)
res = {}
for mode in ("eager", "graph"):
    r = engine.Reconstructor(kpd, gen, use_graph=(mode == "graph"))
    for _ in range(3):
        r(src, drv)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.iters):
        r(src, drv)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.iters
    res[mode] = {"ms_per_batch": round(dt * 1e3, 3), "frames_per_s": round(args.batch / dt, 1)}
print(json.dumps({"workload": "%s eval forward (kp x2 + generator), batch %d @ %dx%d" % (args.config, args.batch, args.size, args.size), **res}))
