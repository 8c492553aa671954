"""SynchronizedBatchNorm{1,2,3}d with the reference's constructor, buffers and state_dict keys
(sync_batchnorm/batchnorm.py:38-46,128-315).  They are parameter holders for the fused kernels in `mnk.ops`
(conv -> statistics -> BN+ReLU(+pool) apply); calling one directly runs the same kernels without the fusion.

Numerics: (var + eps)^-1/2 like the reference's CPU / single-device branch (batchnorm.py:50-53), biased variance
for normalisation, unbiased for running_var, momentum 0.1 (batchnorm.py:113-125).  Under torch.distributed the
per-rank sums are all-reduced, so N ranks x B/N samples reproduce one rank x B samples."""
import torch
from torch.nn.modules.batchnorm import _BatchNorm

from mnk import ops


def _check_exchange(module, prefix, keep_vars):
    from mnk import dist as mdist
    if mdist._P2P["handle"] is not None:
        mdist.check_p2p()


class _SynchronizedBatchNorm(_BatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super(_SynchronizedBatchNorm, self).__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        # a checkpoint written by a user-owned loop (train.py:138-143 -> logger.py:51-55: state_dict() of every network) must not
        # carry running statistics that a given-up peer-to-peer exchange poisoned with NaN: raise instead (mnk.dist.check_p2p)
        self.register_state_dict_pre_hook(_check_exchange)

    def forward(self, input):
        self._check_input_dim(input)
        if not self.affine:
            raise NotImplementedError("affine=False is not used by the reference's modules")
        shape = input.shape
        x5 = input.reshape(shape[0], shape[1], 1, -1, 1)
        act = ops.to_act(x5)
        out = ops.bn_act(act, self.num_features, self, relu=False, pool=False)
        return ops.from_act(out, self.num_features, shape[0]).reshape(shape)


class SynchronizedBatchNorm1d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 2 and input.dim() != 3:
            raise ValueError('expected 2D or 3D input (got {}D input)'.format(input.dim()))


class SynchronizedBatchNorm2d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(input.dim()))


class SynchronizedBatchNorm3d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 5:
            raise ValueError('expected 5D input (got {}D input)'.format(input.dim()))
