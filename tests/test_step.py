"""Step-level parity: three full training iterations (generator step + discriminator step, 3x Adam) through
mnk.engine.TrainStep against the loss history recorded from the reference's own GeneratorFullModel /
DiscriminatorFullModel + torch.optim.Adam (tests/golden/step_tiny.pt, train.py:110-136)."""
import os

import torch

from oracle import cases
from test_modules import build, load


def test_three_training_steps_match_reference_history(be):
    from mnk import engine
    gold = load("step_tiny")
    cfg = gold["cfg"]
    gen, disc, kpd = build(cfg)
    gen.load_state_dict(gold["state"]["generator"])
    disc.load_state_dict(gold["state"]["discriminator"])
    kpd.load_state_dict(gold["state"]["kp_detector"])
    gen.to(be.device), disc.to(be.device), kpd.to(be.device)
    # non-fused Adam: same update formula as the optimiser the golden history was recorded with; the fused ROCm
    # kernel separates 10x faster from the fp64 trajectory (tools/step_diag.py on the MI355X: 7e-2 vs 6e-3 at
    # iteration 1, independent of MIOpen being on or off)
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=False)
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
    x = {"source": be.t(src), "video": be.t(drv)}
    report = []
    for it, (ref, ref64) in enumerate(zip(gold["history"], gold["history64"])):
        g_losses, d_losses, generated = step.step(x)
        be.sync()
        mine = [float(v) for v in g_losses] + [float(v) for v in d_losses]
        r32 = ref["generator"] + ref["discriminator"]
        r64 = ref64["generator"] + ref64["discriminator"]
        # Adam's first updates are sign-like, so rounding noise on near-zero gradient elements moves parameters by
        # +-lr and the trajectories of ANY two fp32 implementations separate step by step.  The yard-stick is the
        # reference's own fp32-vs-fp64 separation at the same iteration (recorded by oracle/make_golden.py).
        spread = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(r32, r64))
        err = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(mine, r64))
        report.append((it, err, spread))
        # factor 16: the yard-stick is ONE realisation of the reference's fp32 noise; on the MI355X (non-fused Adam)
        # iteration 1 measured 8.7e-3 vs 9.0e-4, on the CPU emulator 6.5e-3
        bound = 16.0 * spread + 2e-5
        assert err <= bound, "iteration %d: |hip - ref64| = %.3e vs reference fp32 noise %.3e" % (it, err, spread)
    print("step parity (iteration, |hip-ref64|, |ref32-ref64|):", report)
