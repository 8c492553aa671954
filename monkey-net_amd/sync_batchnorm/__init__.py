"""Drop-in for the reference's `sync_batchnorm` package (sync_batchnorm/__init__.py:11-12): same public names,
MI355X-native underneath -- one process per GPU, statistics exchanged with an RCCL all-reduce (mnk.dist) instead
of DataParallel replica threads talking through SyncMaster queues."""
from .batchnorm import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d, SynchronizedBatchNorm3d
from .replicate import DataParallelWithCallback, patch_replication_callback

__all__ = ["SynchronizedBatchNorm1d", "SynchronizedBatchNorm2d", "SynchronizedBatchNorm3d",
           "DataParallelWithCallback", "patch_replication_callback"]
