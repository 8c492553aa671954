"""Feature-matching L1 kernels (mnk_pair_l1_fwd / _bwd) against modules/losses.py::reconstruction_loss semantics:
weight * mean over (C, H, W) of |generated - real| per sample, and its gradient (torch.abs' sign convention, 0 at 0)."""
import pytest
import torch

from _util import to_nhwc, from_nhwc, ceil4, relerr, maxerr


@pytest.mark.parametrize("shape", [(3, 5, 7, 6), (2, 64, 30, 30), (1, 13, 4, 4), (4, 256, 2, 2)])
def test_pair_l1_forward_backward(be, shape):
    b, c, h, w = shape
    g = torch.Generator().manual_seed(3)
    fake = torch.randn(b, c, h, w, generator=g)
    real = torch.randn(b, c, h, w, generator=g)
    real[0, 0, 0, :2] = fake[0, 0, 0, :2]                  # exact ties: the gradient there is 0
    weight = 10.0
    f64, r64 = fake.double().requires_grad_(True), real.double().requires_grad_(True)
    ref = weight * (f64 - r64).abs().reshape(b, -1).mean(-1)
    gout = torch.randn(b, generator=g).double()
    (ref * gout).sum().backward()
    # pad channels of both halves hold the same garbage-free zeros (acts of this library); poison nothing else
    A = be.t(to_nhwc(torch.cat([fake, real], 0)))
    out = be.empty(b)
    be.call("mnk_pair_l1_fwd", A, ceil4(c), h * w, c, b, weight, out)
    DA = be.empty(2 * b, h, w, ceil4(c)).fill_(float("nan"))
    be.call("mnk_pair_l1_bwd", A, ceil4(c), h * w, c, b, weight, be.t(gout.float()), DA)
    be.sync()
    assert relerr(out.cpu(), ref.detach()) < 2e-6
    d = DA.cpu()
    assert maxerr(from_nhwc(d[:b], c), f64.grad) < 1e-6 * (1 + float(f64.grad.abs().max()))
    assert maxerr(from_nhwc(d[b:], c), r64.grad) < 1e-6 * (1 + float(r64.grad.abs().max()))
    assert torch.all(d[..., c:] == 0)
    assert float(from_nhwc(d[:b], c)[0, 0, 0, 0]) == 0.0


def test_pair_l1_autograd_function(be):
    from mnk import ops
    b, c, h, w = 2, 10, 6, 5
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2 * b, c, h, w, generator=g)
    act = ops.to_act(be.t(x.unsqueeze(2).clone())).detach().requires_grad_(True)
    out = ops.PairL1Fn.apply(act, c, b, 2.5)
    gout = torch.randn(b, generator=g)
    (out * be.t(gout)).sum().backward()
    be.sync()
    x64 = x.double().requires_grad_(True)
    ref = 2.5 * (x64[:b] - x64[b:]).abs().reshape(b, -1).mean(-1)
    (ref * gout.double()).sum().backward()
    assert relerr(out.detach().cpu(), ref.detach()) < 2e-6
    assert maxerr(from_nhwc(act.grad.cpu(), c), x64.grad) < 1e-6
