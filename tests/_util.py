"""Helpers shared by the kernel tests."""
import torch


def ceil4(c):
    return (c + 3) // 4 * 4


def to_nhwc(x4, ld=None, pad_value=0.0):
    """(N,C,H,W) -> (N,H,W,ld) with pad channels = pad_value."""
    n, c, h, w = x4.shape
    ld = ld or ceil4(c)
    out = torch.full((n, h, w, ld), pad_value, dtype=x4.dtype)
    out[..., :c] = x4.permute(0, 2, 3, 1)
    return out


def from_nhwc(x, c):
    return x[..., :c].permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxerr(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max())


def set_library(path, strict=True):
    """TEST INFRASTRUCTURE: bind the C-ABI of another build of the same kernel sources (the CPU emulator build used by the
    `-m "not gpu"` tests) by replacing the product's process-wide library handle; None: back to the in-tree gfx950 build on
    next use.  The product has no such switch (its only library override is the MNK_LIBRARY path)."""
    from mnk import _lib
    _lib._LIB = _lib.Library(path, strict=strict) if path is not None else None
    return _lib._LIB
