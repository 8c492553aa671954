#!/usr/bin/env python
"""Which gradient goes non-finite first?  tools/train_sanity.py's run (seed, fixed blurred batch), eager, with every
optimiser's gradients checked right before its step.  Usage (GPU box): python tools/nan_diag.py --seed 11 [--steps 8]"""
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
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--fused", type=int, default=-1, help="-1 default pipeline, 0 torch.optim.Adam pipeline")
    ap.add_argument("--wrap", default="check", help="check: gradients checked before every optimiser step; sync: only a device "
                    "synchronisation there; materialize: only materialize_grads(); none")
    ap.add_argument("--poison", type=int, default=0, help="1: every torch.empty / empty_like of mnk.ops, mnk.optim and the "
                    "discriminator is pre-filled with NaN: an unwritten element that is read shows up at once")
    args = ap.parse_args()
    cfg = configs.get("moving-gif")
    if args.poison:
        from mnk import ops, optim, discriminator_hip

        def nan_empty(*a, **k):
            t = torch.zeros(*a, **k)
            return t.fill_(float("nan")) if t.is_floating_point() else t

        def nan_empty_like(t0, **k):
            t = torch.zeros_like(t0, **k)
            return t.fill_(float("nan")) if t.is_floating_point() else t

        class TorchProxy:
            def __getattr__(self, item):
                if item == "empty":
                    return nan_empty
                if item == "empty_like":
                    return nan_empty_like
                return getattr(torch, item)

        for m in (ops, optim, discriminator_hip):
            m.torch = TorchProxy()
    torch.manual_seed(args.seed)
    mp = cfg["model_params"]
    gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]).cuda()
    disc = Discriminator(**mp["discriminator_params"], **mp["common_params"]).cuda()
    kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"]).cuda()
    src, drv = workload.synthetic_pair(32, 64, 64)
    blur = torch.nn.AvgPool2d(9, stride=1, padding=4)
    src = blur(src[:, :, 0]).unsqueeze(2).contiguous()
    drv = blur(drv[:, :, 0]).unsqueeze(2).contiguous()
    x = {"source": src.cuda(), "video": drv.cuda()}
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=None if args.fused < 0 else bool(args.fused))
    state = {"it": 0}
    for name, opt, mod in (("generator", step.opt_g, gen), ("kp_detector", step.opt_k, kpd), ("discriminator", step.opt_d, disc)):
        def wrapped(real=opt.step, name=name, mod=mod, opt=opt):
            if args.wrap == "sync":
                torch.cuda.synchronize()
                return real()
            if args.wrap == "materialize":
                opt.materialize_grads()
                return real()
            if hasattr(opt, "materialize_grads"):
                opt.materialize_grads()
            bad = [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in mod.named_parameters()
                   if p.grad is not None and not torch.isfinite(p.grad).all()]
            big = sorted(((float(p.grad.abs().max()), n) for n, p in mod.named_parameters() if p.grad is not None), reverse=True)[:3]
            print("iteration %d %-13s: %d tensors with non-finite gradients %s | largest |g|: %s" % (
                state["it"], name, len(bad), bad[:4], ["%s %.3e" % (n, v) for v, n in big]), flush=True)
            return real()
        if args.wrap != "none":
            opt.step = wrapped
    for it in range(args.steps):
        state["it"] = it
        g_l, d_l, out = step._eager_step(x)
        v = out["kp_driving"]["var"].reshape(-1, 2, 2)
        vs = out["kp_source"]["var"].reshape(-1, 2, 2)
        def cond(v):
            det = v[:, 0, 0] * v[:, 1, 1] - v[:, 0, 1] * v[:, 1, 0]
            return "min det %.3e min diag %.3e max |offdiag| %.3e" % (float(det.min()), float(torch.minimum(v[:, 0, 0], v[:, 1, 1]).min()),
                                                                     float(v[:, 0, 1].abs().max()))
        print("iteration %d losses %s | driving var: %s | source var: %s" % (
            it, " ".join("%.4f" % float(t) for t in list(g_l) + list(d_l)), cond(v), cond(vs)), flush=True)
        if not all(torch.isfinite(t).all() for t in list(g_l) + list(d_l)):
            break


if __name__ == "__main__":
    main()
