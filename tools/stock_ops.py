#!/usr/bin/env python
"""Which stock PyTorch operators does one training iteration still run, and where from?  (no stock ops on
the hot path.)  Runs TrainStep on the CPU emulator build of the kernels (tests/hipemu: same Python, same autograd graph as on
the MI355X) under a TorchDispatchMode and prints every aten operator that would be a kernel launch on the device, with the
innermost frame of this repository that caused it (autograd's own accumulation shows up as `<autograd engine>`).

    python tools/stock_ops.py [--config moving-gif] [--batch 2] [--size 64] [--torch-adam]"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "monkey-net_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

# metadata-only operators: no launch on the device
FREE = ("view", "reshape", "_unsafe_view", "alias", "detach", "as_strided", "t.", "transpose", "permute", "expand", "slice",
        "select", "unsqueeze", "squeeze", "empty", "_local_scalar_dense", "is_same_size", "unbind", "split", "narrow",
        "new_empty", "lift_fresh", "_to_copy", "stride", "sym_", "size", "numel", "is_pinned", "record_stream", "prim.")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        short = name.replace("aten.", "")
        if not any(short.startswith(f) or ("." + f) in name for f in FREE):
            site = "<autograd engine>"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if fn.startswith(ROOT) and "tools/stock_ops.py" not in fn:
                    site = "%s:%d %s" % (os.path.relpath(fn, ROOT), fr.lineno, fr.name)
                    break
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            self.sites[(short, site, str(shapes))] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="tiny")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--torch-adam", action="store_true")
    a = ap.parse_args()
    import conftest
    import _util
    _util.set_library(conftest.emu_library_path(), strict=False)
    from oracle import cases
    from mnk import configs, engine
    from test_modules import build
    cfg = cases.TINY if a.config == "tiny" else configs.get(a.config)
    gen, disc, kpd = build(cfg)
    src, drv = cases.smooth_pair(a.batch, a.size, a.size)
    x = {"source": src, "video": drv}
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=not a.torch_adam)
    for _ in range(2):
        step.step(x)
    log = Log()
    with log:
        step.step(x)
    total = 0
    for (op, site, shapes), n in sorted(log.sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print("%3d  %-28s %-70s %s" % (n, op, site, shapes))
        total += n
    print("total stock-operator calls in one iteration:", total)


if __name__ == "__main__":
    main()
