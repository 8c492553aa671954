#!/usr/bin/env python
"""TEST INFRASTRUCTURE (authoring container only: needs /root/reference).  Wall-clock time of the UNMODIFIED reference's
training iteration on the host CPU -- BASELINE.md section 2, rows 2 and 4 ("to measure in build").

What is timed is train.py:110-136 as the reference runs it: its own GeneratorFullModel / DiscriminatorFullModel
(train.py:24-75) around its own Conv3d modules, three torch.optim.Adam(betas=(0.5, 0.999)) steps, the per-iteration
loss means -- imported through oracle/ref_shim.py (torch.gesv / grid_sample pins only), synthetic U[0,1) pairs of the bench
protocol (mnk.workload.synthetic_pair = oracle.cases.synthetic_pair), one warm-up iteration, then `--steps` timed ones.

    python oracle/time_reference.py --config moving-gif --batch 32 --size 64 [--steps 3] [--threads N]

Prints one JSON line; BASELINE.md quotes it and bench.py's cpu_baseline.sample cites BASELINE.md's row."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import cases, ref_shim  # noqa: E402
from oracle.make_golden import load_cfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="moving-gif")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    ref = ref_shim.load()
    cfg = load_cfg(args.config)
    tp, mp = cfg["train_params"], cfg["model_params"]
    torch.manual_seed(0)                                                         # run.py:50-62 construction order
    gen = ref.MotionTransferGenerator(**mp["generator_params"], **mp["common_params"])
    disc = ref.Discriminator(**mp["discriminator_params"], **mp["common_params"])
    kpd = ref.KPDetector(**mp["kp_detector_params"], **mp["common_params"])
    for m in (gen, disc, kpd):
        m.train()
    opts = [torch.optim.Adam(m.parameters(), lr=tp["lr"], betas=(0.5, 0.999)) for m in (gen, disc, kpd)]   # train.py:81-83
    gfull = ref.GeneratorFullModel(kpd, gen, disc, tp)
    dfull = ref.DiscriminatorFullModel(kpd, gen, disc, tp)
    src, drv = cases.synthetic_pair(args.batch, args.size, args.size)
    x = {"source": src, "video": drv}

    def iteration():                                                             # train.py:110-136
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
        return [float(v) for v in lv + dl]                                       # train.py:138 (.cpu() of the loss values)

    iteration()
    times = []
    for _ in range(args.steps):
        t0 = time.perf_counter()
        losses = iteration()
        times.append(time.perf_counter() - t0)
    best = min(times)
    mean = sum(times) / len(times)
    cpu = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    print(json.dumps({"what": "unmodified reference, train.py:110-136 iteration, CPU", "config": args.config,
                      "batch": args.batch, "size": args.size, "threads": args.threads, "host_cores": os.cpu_count(),
                      "cpu": cpu, "torch": torch.__version__, "steps": args.steps,
                      "s_per_step_mean": round(mean, 4), "s_per_step_best": round(best, 4),
                      "frames_per_s": round(args.batch / mean, 3), "finite": all(v == v for v in losses)}))


if __name__ == "__main__":
    main()
