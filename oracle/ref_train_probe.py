#!/usr/bin/env python
"""TEST INFRASTRUCTURE (authoring container only: needs /root/reference).  Does the REFERENCE itself go non-finite on the
fixed batch of tools/train_sanity.py for a given initialisation seed?  Runs train.py:110-136 (the reference's own
GeneratorFullModel / DiscriminatorFullModel, torch.optim.Adam betas (0.5, 0.999)) for a few iterations on the CPU in fp32 and
fp64 and reports the first non-finite loss / parameter and the conditioning of the key-point covariances.
Usage: python oracle/ref_train_probe.py --seed 11 [--steps 8]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import cases, ref_shim  # noqa: E402
from oracle.make_golden import load_cfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=11)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--dtypes", default="float32,float64")
    args = ap.parse_args()
    ref = ref_shim.load()
    cfg = load_cfg("moving-gif")
    tp, mp = cfg["train_params"], cfg["model_params"]
    g = torch.Generator().manual_seed(1234)                 # mnk.workload.synthetic_pair(32, 64, 64)
    src = torch.rand(32, 3, 1, 64, 64, generator=g)
    drv = torch.rand(32, 3, 1, 64, 64, generator=g)
    blur = torch.nn.AvgPool2d(9, stride=1, padding=4)
    src = blur(src[:, :, 0]).unsqueeze(2).contiguous()
    drv = blur(drv[:, :, 0]).unsqueeze(2).contiguous()
    for name in args.dtypes.split(","):
        dtype = getattr(torch, name)
        torch.manual_seed(args.seed)
        gen = ref.MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]).to(dtype)
        disc = ref.Discriminator(**mp["discriminator_params"], **mp["common_params"]).to(dtype)
        kpd = ref.KPDetector(**mp["kp_detector_params"], **mp["common_params"]).to(dtype)
        opts = [torch.optim.Adam(m.parameters(), lr=tp["lr"], betas=(0.5, 0.999)) for m in (gen, disc, kpd)]
        gfull = ref.GeneratorFullModel(kpd, gen, disc, tp)
        dfull = ref.DiscriminatorFullModel(kpd, gen, disc, tp)
        x = {"source": src.to(dtype), "video": drv.to(dtype)}
        for it in range(args.steps):
            outs = gfull(x)
            lv = [v.mean() for v in outs[:-2]]
            generated, kp_joined = outs[-2], outs[-1]
            sum(lv).backward(retain_graph=not tp["detach_kp_discriminator"])
            opts[0].step(), opts[0].zero_grad(), opts[1].zero_grad()
            if tp["detach_kp_discriminator"]:
                opts[2].step(), opts[2].zero_grad()
            dl = [v.mean() for v in dfull(x, kp_joined, generated)]
            sum(dl).backward()
            opts[1].step(), opts[1].zero_grad()
            if not tp["detach_kp_discriminator"]:
                opts[2].step(), opts[2].zero_grad()
            v = kp_joined["var"].detach().reshape(-1, 2, 2).double()
            det = v[:, 0, 0] * v[:, 1, 1] - v[:, 0, 1] * v[:, 1, 0]
            bad = [n for m in (kpd, gen, disc) for n, p in m.named_parameters() if not torch.isfinite(p).all()]
            print("%s iteration %d: losses %s | min det(var) %.3e, min diag %.3e | non-finite parameters: %d %s" % (
                name, it, " ".join("%.4f" % float(t) for t in lv + dl), float(det.min()),
                float(torch.minimum(v[:, 0, 0], v[:, 1, 1]).min()), len(bad), bad[:2]), flush=True)
            if bad:
                break


if __name__ == "__main__":
    main()
