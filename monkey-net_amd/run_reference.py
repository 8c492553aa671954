#!/usr/bin/env python
"""Run one of the reference's own entry scripts (run.py, demo.py) on top of the MI355X drop-in packages.

    python monkey-net_amd/run_reference.py /path/to/monkey-net/run.py --config config/taichi.yaml --device_ids 0

`python run.py` puts the script's own directory at sys.path[0], so a PYTHONPATH entry can never win over the reference's
`modules/` and `sync_batchnorm/`; this launcher runs the script with sys.path = [monkey-net_amd, <script dir>, ...]:
`modules.*` / `sync_batchnorm.*` resolve here, everything else (train.py, logger.py, frames_dataset.py,
modules.prediction_module) in the reference tree.  Under torch.distributed.run (one process per GPU) it also joins the
process group before the script starts, so SyncBN statistics and gradients are exchanged over RCCL without editing
train.py (sync_batchnorm.DataParallelWithCallback installs the gradient averaging, see replicate.py).  Batch semantics are
DataParallel's: the YAML batch_size is the GLOBAL batch -- every rank's DataLoader produces a batch, rank 0's is broadcast and
each rank computes its slice (replicate.py, MNK_DP_SCATTER) -- so epochs, learning-rate milestones and logged iteration
counts mean what they mean in the reference; ranks > 0 load data they do not use (seed the loaders equally and set
MNK_DP_SCATTER=slice to skip the broadcast)."""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        sys.exit("usage: run_reference.py <reference script> [script arguments]")
    script = os.path.abspath(sys.argv[1])
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:] = [here, os.path.dirname(script)] + [p for p in sys.path[1:] if os.path.abspath(p or ".") != here]
    sys.argv = [script] + sys.argv[2:]
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group("nccl")            # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
