"""`DataParallelWithCallback` / `patch_replication_callback` (sync_batchnorm/replicate.py:50-94) for the
one-process-per-GPU design.

The reference wraps its full models in a single-process `DataParallel` that scatters the batch, re-broadcasts all
parameters on every forward and runs one thread per GPU (train.py:104-105).  Here every rank already owns a resident
replica and its shard of the batch, so the wrapper only (a) moves the inputs to this rank's device and (b) calls the
wrapped module; statistics and gradients are exchanged by `mnk.dist` (RCCL).  The class keeps the reference's name,
constructor and call signature so train.py / reconstruction.py / transfer.py / demo.py run unchanged."""
import warnings

import torch
from torch import nn


def _to_device(obj, device):
    if torch.is_tensor(obj):
        return obj.to(device, non_blocking=True)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(v, device) for v in obj)
    return obj


class DataParallelWithCallback(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super(DataParallelWithCallback, self).__init__()
        self.module = module
        self.dim = dim
        self.device_ids = list(device_ids) if device_ids is not None else None
        self.output_device = output_device
        from mnk import dist as mdist
        if mdist.initialized() and mdist.world_size() > 1:
            # one process per GPU under torch.distributed.run: SyncBN statistics are exchanged inside the BN wrappers
            # (mnk.dist), and any optimiser's gradients are averaged right before its step -- train.py runs unchanged
            mdist.install_grad_averaging()
        elif self.device_ids is not None and len(self.device_ids) > 1:
            warnings.warn("DataParallelWithCallback(device_ids=%r): this implementation runs ONE process per GPU -- launch "
                          "the script under `python -m torch.distributed.run --nproc-per-node %d ...` (monkey-net_amd/"
                          "run_reference.py joins the process group); without a process group the whole batch runs on "
                          "%s alone." % (self.device_ids, len(self.device_ids), self._device()), stacklevel=2)

    def _device(self):
        for p in self.module.parameters():
            return p.device
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")

    def forward(self, *inputs, **kwargs):
        dev = self._device()
        return self.module(*_to_device(inputs, dev), **_to_device(kwargs, dev))

    def replicate(self, module, device_ids):  # kept for API compatibility; replicas are separate processes here
        return [module]


def patch_replication_callback(data_parallel):
    """No-op: there is no in-process replication to hook into (one process drives one GPU)."""
    return data_parallel
