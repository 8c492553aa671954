#!/usr/bin/env python
"""tests/test_train_sanity.py's fixed-batch run, repeated: at which iteration does a loss first go non-finite, per repetition?
(the warp backward accumulates with atomics, so two runs of a borderline initialisation need not agree)
Usage (GPU box): python tools/sanity_repeat.py --seed 11 --graph 1 --reps 4 [--iters 40]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import torch  # noqa: E402

from mnk import configs, engine, workload  # noqa: E402
from modules.generator import MotionTransferGenerator  # noqa: E402
from modules.discriminator import Discriminator  # noqa: E402
from modules.keypoint_detector import KPDetector  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--nosync", type=int, default=0, help="1: no host synchronisation inside the loop (the test's form)")
    args = ap.parse_args()
    cfg = configs.get("moving-gif")
    mp = cfg["model_params"]
    src, drv = workload.synthetic_pair(32, 64, 64)
    blur = torch.nn.AvgPool2d(9, stride=1, padding=4)
    x = {"source": blur(src[:, :, 0]).unsqueeze(2).contiguous().cuda(), "video": blur(drv[:, :, 0]).unsqueeze(2).contiguous().cuda()}
    for rep in range(args.reps):
        torch.manual_seed(args.seed)
        gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]).cuda()
        disc = Discriminator(**mp["discriminator_params"], **mp["common_params"]).cuda()
        kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"]).cuda()
        step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=bool(args.graph))
        first_bad, hist = None, []
        if args.nosync:
            for it in range(args.iters):
                g_l, d_l, out = step.step(x)
            vals = [float(v) for v in g_l] + [float(v) for v in d_l]
            ok = all(v == v and abs(v) < 1e6 for v in vals) and all(bool(torch.isfinite(p).all()) for m in (gen, disc, kpd)
                                                                    for p in m.parameters())
            print("seed %d graph %d rep %d nosync: %s %s" % (args.seed, args.graph, rep, "finite" if ok else "NON-FINITE",
                                                          [round(v, 4) for v in vals]), flush=True)
            continue
        for it in range(args.iters):
            g_l, d_l, out = step.step(x)
            vals = [float(v) for v in g_l] + [float(v) for v in d_l]
            kv = out["kp_driving"]["var"]
            det = (kv[..., 0, 0] * kv[..., 1, 1] - kv[..., 0, 1] * kv[..., 1, 0])
            hist.append((round(sum(vals[:-2]), 4), float(det.min()), float(kv.abs().max())))
            if first_bad is None and not all(v == v and abs(v) < 1e6 for v in vals):
                first_bad = it
        print("seed %d graph %d rep %d: first non-finite iteration %s; (sum rec losses, min det(var), max |var|) at 0/1/2/3/5/10/last: %s" % (
            args.seed, args.graph, rep, first_bad, [hist[i] for i in (0, 1, 2, 3, 5, 10, len(hist) - 1) if i < len(hist)]), flush=True)


if __name__ == "__main__":
    main()
