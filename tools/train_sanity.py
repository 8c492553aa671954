#!/usr/bin/env python
"""Longer run of the captured training iteration on one fixed synthetic batch (moving-gif parameters, batch 32 @ 64x64): the
losses must stay finite and the reconstruction terms must fall (the network can over-fit one batch).
Usage (GPU box): python tools/train_sanity.py [--steps 400]"""
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
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--graph", type=int, default=1)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--config", default="moving-gif")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--watch", type=int, default=0, help="1: check every parameter and the key points after every iteration")
    args = ap.parse_args()
    cfg = configs.get(args.config)
    torch.manual_seed(args.seed)
    mp = cfg["model_params"]
    gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]).cuda()
    disc = Discriminator(**mp["discriminator_params"], **mp["common_params"]).cuda()
    kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"]).cuda()
    src, drv = workload.synthetic_pair(args.batch, args.size, args.size)
    # smooth frames (the uniform-noise bench input has nothing to learn): blur the noise
    blur = torch.nn.AvgPool2d(9, stride=1, padding=4)
    src = blur(src[:, :, 0]).unsqueeze(2).contiguous()
    drv = blur(drv[:, :, 0]).unsqueeze(2).contiguous()
    x = {"source": src.cuda(), "video": drv.cuda()}
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=bool(args.graph))
    first = None
    for it in range(args.steps):
        g_l, d_l, gen_out = step.step(x)
        if args.watch:
            bad = [n for m, tag in ((kpd, "kp"), (gen, "gen"), (disc, "disc")) for n, p in m.named_parameters()
                   if not torch.isfinite(p).all()]
            kp_bad = {k: int((~torch.isfinite(v)).sum()) for k, v in gen_out.items()
                      if torch.is_tensor(v) and not torch.isfinite(v).all()}
            if bad or kp_bad or not all(torch.isfinite(v).all() for v in list(g_l) + list(d_l)):
                print("iteration %d: first non-finite parameters %s; outputs %s; losses %s" % (
                    it, bad[:6], kp_bad, [float(v) for v in list(g_l) + list(d_l)]))
                kd = gen_out.get("kp_driving", {})
                if "var" in kd:
                    v = kd["var"].reshape(-1, 2, 2)
                    det = v[:, 0, 0] * v[:, 1, 1] - v[:, 0, 1] * v[:, 1, 0]
                    print("driving var: min det %.3e, min diag %.3e, non-finite %d" % (
                        float(det[torch.isfinite(det)].min()) if torch.isfinite(det).any() else float("nan"),
                        float(torch.minimum(v[:, 0, 0], v[:, 1, 1]).nan_to_num(1.0).min()), int((~torch.isfinite(v)).sum())))
                sys.exit(3)
        if it % args.every == 0 or it == args.steps - 1:
            vals = [float(v) for v in g_l] + [float(v) for v in d_l]
            first = first or vals
            print("iteration %4d  generator terms %s  discriminator %s" % (
                it, " ".join("%.4f" % v for v in vals[:-1]), "%.4f" % vals[-1]))
    last = vals
    assert all(v == v and abs(v) < 1e6 for v in last), last
    rec_first, rec_last = sum(first[:-2]), sum(last[:-2])
    print("reconstruction terms %.4f -> %.4f" % (rec_first, rec_last))
    assert rec_last < rec_first, "the reconstruction terms did not fall on a fixed batch"
    # evaluation-mode forward with the trained weights (running statistics): reconstruction of the batch
    rec = engine.Reconstructor(kpd, gen, use_graph=False)
    with torch.no_grad():
        out = rec(x["source"], x["video"])
    pred = out["video_prediction"] if isinstance(out, dict) else out
    assert torch.isfinite(pred).all(), "non-finite evaluation forward"
    print("evaluation forward: L1 to the driving frames %.4f" % float((pred - x["video"]).abs().mean()))
    print("ok")


if __name__ == "__main__":
    main()
