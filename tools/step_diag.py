#!/usr/bin/env python
"""GPU diagnostic: how far does the 3-iteration loss history separate from the fp64 reference under different stock-op
settings (MIOpen on/off, fused Adam on/off)?  Explains the coarse band of tests/test_step.py on the device."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "monkey-net_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from oracle import cases  # noqa: E402
from test_modules import build, load  # noqa: E402
from mnk import engine  # noqa: E402

gold = load("step_tiny")
cfg = gold["cfg"]
dev = torch.device("cuda:0")
for cudnn in (True, False):
    for fused in (True, False):
        torch.backends.cudnn.enabled = cudnn
        gen, disc, kpd = build(cfg)
        gen.load_state_dict(gold["state"]["generator"]); disc.load_state_dict(gold["state"]["discriminator"])
        kpd.load_state_dict(gold["state"]["kp_detector"])
        gen.to(dev), disc.to(dev), kpd.to(dev)
        step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=fused)
        src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
        x = {"source": src.to(dev), "video": drv.to(dev)}
        rep = []
        for it, (ref, ref64) in enumerate(zip(gold["history"], gold["history64"])):
            g, d, _ = step.step(x)
            mine = [float(v) for v in g] + [float(v) for v in d]
            r64 = ref64["generator"] + ref64["discriminator"]
            r32 = ref["generator"] + ref["discriminator"]
            rep.append(("%.2e" % max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(mine, r64)),
                        "%.2e" % max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(r32, r64))))
        print("miopen=%s fused_adam=%s  (|hip-ref64|, |ref32-ref64|) per iteration: %s" % (cudnn, fused, rep))
