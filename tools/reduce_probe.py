#!/usr/bin/env python
"""Where the time of mnk_wgrad_reduce_multi goes: one eager training iteration fills the generator optimiser's partial buffers,
then the reduction is timed for subsets of its layer table (stale partials: timing only).
Usage on the GPU box: MNK_WGRAD_BG=0 python tools/reduce_probe.py [--config moving-gif --batch 32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mnk import configs, engine, ops as mops, optim, workload  # noqa: E402
from modules.generator import MotionTransferGenerator  # noqa: E402
from modules.discriminator import Discriminator  # noqa: E402
from modules.keypoint_detector import KPDetector  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="moving-gif")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--opt", default="g", choices=["g", "d", "kp"], help="whose optimiser: generator / discriminator / key points")
    a = ap.parse_args()
    cfg = configs.get(a.config)
    mp = cfg["model_params"]
    torch.manual_seed(0)
    gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"]).cuda()
    disc = Discriminator(**mp["discriminator_params"], **mp["common_params"]).cuda()
    kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"]).cuda()
    src, drv = workload.synthetic_pair(a.batch, a.size, a.size)
    x = {"source": src.cuda(), "video": drv.cuda()}
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=False)
    for _ in range(2):
        step.step(x)
    torch.cuda.synchronize()
    red = {"g": step.opt_g, "d": step.opt_d, "kp": step.opt_k}[a.opt].reducer
    names = {id(p): n for n, p in {"g": gen, "d": disc, "kp": kpd}[a.opt].named_parameters()}
    recs = [(k, r) for k, r in red.recs.items() if r["splits"] > 0]

    def table(sel):
        rec = np.zeros(len(sel), dtype=optim.REDUCE_DESC)
        blocks = 0
        for i, (k, r) in enumerate(sel):
            rec[i] = r["row"] + (blocks, 0)
            blocks += r["blocks"]
        t = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda()
        return t, len(sel), blocks

    def timeit(sel, iters=20):
        if not sel:
            return 0.0, 0
        t, n, blocks = table(sel)
        for _ in range(3):
            mops._call("mnk_wgrad_reduce_multi", t, mops._p(t), n, blocks)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            mops._call("mnk_wgrad_reduce_multi", t, mops._p(t), n, blocks)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3, blocks

    def mb(sel):
        return sum(r["nfloats"] for _, r in sel) * 4 / 1e6

    groups = {"all": recs,
              "splits < 4 (the gradient itself, tap-major)": [x for x in recs if x[1]["splits"] < 4],
              "4 <= splits < 32": [x for x in recs if 4 <= x[1]["splits"] < 32],
              "splits >= 32": [x for x in recs if x[1]["splits"] >= 32]}
    for name, sel in groups.items():
        us, blocks = timeit(sel)
        print("%-46s layers %3d  partials %7.1f MB  blocks %6d  %7.1f us  %.2f TB/s read" % (
            name, len(sel), mb(sel), blocks, us, mb(sel) / max(us, 1e-9) / 1e6 * 1e6 / 1e6))
    print("-- per layer (alone)")
    for k, r in sorted(recs, key=lambda kr: -kr[1]["nfloats"]):
        us, blocks = timeit([(k, r)], iters=10)
        row = r["row"]
        td, red.owner.tap_direct = red.owner.tap_direct, True
        direct = red.is_direct(k)
        red.owner.tap_direct = td
        print("%-46s splits %4d layout %d partial %7.2f MB blocks %5d %7.1f us  C %4d Cout %4d%s" % (
            names.get(k[0], "?")[:46], r["splits"], row[2], r["nfloats"] * 4 / 1e6, blocks, us, row[6], row[5],
            "  direct (read by the optimiser kernel in a captured iteration)" if direct else ""))


if __name__ == "__main__":
    main()
