#!/usr/bin/env python
"""Time the generator optimiser's mnk_adam_multi launch alone (after one eager training iteration built its descriptor table).
Usage on the GPU box: MNK_WGRAD_BG=0 [MNK_LIBRARY=variant.so] python tools/adam_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import torch  # noqa: E402

from mnk import configs, engine, ops as mops, workload  # noqa: E402
from modules.generator import MotionTransferGenerator  # noqa: E402
from modules.discriminator import Discriminator  # noqa: E402
from modules.keypoint_detector import KPDetector  # noqa: E402

cfg = configs.get("moving-gif")
mp = cfg["model_params"]
torch.manual_seed(0)
gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]).cuda()
disc = Discriminator(**mp["discriminator_params"], **mp["common_params"]).cuda()
kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"]).cuda()
src, drv = workload.synthetic_pair(32, 64, 64)
x = {"source": src.cuda(), "video": drv.cuda()}
step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
for _ in range(2):
    step.step(x)
torch.cuda.synchronize()
for name, opt in (("generator", step.opt_g), ("kp_detector", step.opt_k), ("discriminator", step.opt_d)):
    _, tab, n, blocks, entries = opt._table
    nparam = sum(p.numel() for p in opt._params)
    for _ in range(3):
        mops._call("mnk_adam_multi", opt.hyper, mops._p(tab), n, blocks, mops._p(opt.hyper))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        mops._call("mnk_adam_multi", opt.hyper, mops._p(tab), n, blocks, mops._p(opt.hyper))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%-14s %4d tensors %9d parameters %6d blocks  %7.1f us  (%.1f B/parameter at 6 TB/s would be %.1f us)" % (
        name, n, nparam, blocks, us, 36.0, nparam * 36.0 / 6e12 * 1e6))
