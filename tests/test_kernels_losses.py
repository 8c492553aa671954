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


@pytest.mark.parametrize("shape", [(3, 3, 1, 9, 7), (2, 3, 1, 64, 64), (4, 1, 1, 1, 5)])
def test_l1_mean_and_its_autograd_function(be, shape):
    """mnk_l1_mean_fwd / _bwd == weight * mean_batch(|prediction - target|) of modules/losses.py:8-12 on NCDHW frames."""
    from mnk import ops
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    b.view(-1)[:3] = a.view(-1)[:3]                         # exact ties: gradient 0
    n = shape[0]
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = 7.0 * (a64 - b64).abs().reshape(n, -1).mean(-1)
    gout = torch.randn(n, generator=g)
    (ref * gout.double()).sum().backward()
    A, B = be.t(a).requires_grad_(True), be.t(b).requires_grad_(True)
    out = ops.L1MeanFn.apply(A, B, 7)
    (out * be.t(gout)).sum().backward()
    be.sync()
    assert relerr(out.detach().cpu(), ref.detach()) < 2e-6
    assert maxerr(A.grad.cpu(), a64.grad) < 1e-6 * (1 + float(a64.grad.abs().max()))
    assert maxerr(B.grad.cpu(), b64.grad) < 1e-6 * (1 + float(b64.grad.abs().max()))
    assert float(A.grad.cpu().view(-1)[0]) == 0.0
    # only one side asks for a gradient (reconstruction_deformed: the real frames are data)
    A2, B2 = be.t(a), be.t(b).requires_grad_(True)
    (ops.L1MeanFn.apply(A2, B2, 7) * be.t(gout)).sum().backward()
    be.sync()
    assert torch.equal(B2.grad.cpu(), B.grad.cpu())


@pytest.mark.parametrize("shape", [(3, 1, 1, 1, 1), (2, 1, 1, 5, 4)])
def test_gan_terms_match_the_loss_module(be, shape):
    """GanTermsFn == (generator_gan_loss, discriminator_gan_loss) of modules/losses.py on the batched score maps, for the
    three ways the step differentiates them: the generator term alone, the discriminator term alone, both."""
    from mnk import ops
    from modules import losses
    b = shape[0]
    g = torch.Generator().manual_seed(6)
    score = torch.randn((2 * b,) + tuple(shape[1:]), generator=g)
    s64 = score.double().requires_grad_(True)
    ref_g = losses.generator_gan_loss([s64[:b]], 1.5)
    ref_d = losses.discriminator_gan_loss([s64[:b]], [s64[b:]], 0.75)
    gg, gd = torch.randn(b, generator=g), torch.randn(b, generator=g)
    S = be.t(score).requires_grad_(True)
    gen, disc = ops.GanTermsFn.apply(S, b, 1.5, 0.75)
    be.sync()
    assert relerr(gen.detach().cpu(), ref_g.detach()) < 2e-6 and relerr(disc.detach().cpu(), ref_d.detach()) < 2e-6
    for use_g, use_d in ((True, False), (False, True), (True, True)):
        S.grad, s64.grad = None, None
        tot = (gen * be.t(gg)).sum() * float(use_g) + (disc * be.t(gd)).sum() * float(use_d)
        ref = (ref_g * gg.double()).sum() * float(use_g) + (ref_d * gd.double()).sum() * float(use_d)
        if use_g and not use_d:
            tot, ref = (gen * be.t(gg)).sum(), (ref_g * gg.double()).sum()
        if use_d and not use_g:
            tot, ref = (disc * be.t(gd)).sum(), (ref_d * gd.double()).sum()
        tot.backward(retain_graph=True)
        ref.backward(retain_graph=True)
        be.sync()
        assert maxerr(S.grad.cpu(), s64.grad) < 1e-6 * (1 + float(s64.grad.abs().max()))


def test_loss_module_runs_the_l1_kernels_on_the_library_device(be):
    """modules.losses.reconstruction_loss (the function the reference's train.py calls, losses.py:8-12) on fp32 tensors of the
    library's device is the one-launch kernel pair (an L1MeanFn node), equal to the stock arithmetic it replaces; fp64 / CPU
    checks keep the stock arithmetic."""
    from modules import losses
    g = torch.Generator().manual_seed(7)
    a, b = torch.randn(3, 4, 1, 6, 5, generator=g), torch.randn(3, 4, 1, 6, 5, generator=g)
    A, B = be.t(a).requires_grad_(True), be.t(b).requires_grad_(True)
    out = losses.reconstruction_loss(A, B, 10)
    assert "L1MeanFn" in type(out.grad_fn).__name__
    gout = torch.randn(3, generator=g)
    (out * be.t(gout)).sum().backward()
    be.sync()
    a64, b64 = a.double().detach().requires_grad_(True), b.double().detach().requires_grad_(True)
    ref = losses.reconstruction_loss(a64, b64, 10)
    assert "L1MeanFn" not in type(ref.grad_fn).__name__
    (ref * gout.double()).sum().backward()
    assert relerr(out.detach().cpu(), ref.detach()) < 2e-6
    assert maxerr(A.grad.cpu(), a64.grad) < 1e-6 and maxerr(B.grad.cpu(), b64.grad) < 1e-6
    assert losses.reconstruction_loss(A, B, 0) == 0


@pytest.mark.parametrize("nvec,length", [(1, 4), (6, 32), (16, 70)])
def test_batch_means_of_the_loss_vectors(be, nvec, length):
    """mnk_vec_means_fwd / _bwd and ops.LossMeansFn against `[v.mean() for v in losses]`, `sum(loss_values)` (train.py:114,116)
    and autograd's gradients of both outputs."""
    import numpy as np
    from mnk import ops
    g = torch.Generator().manual_seed(5)
    vecs = [torch.randn(length, generator=g) * (i + 1) for i in range(nvec)]
    V = [be.t(v) for v in vecs]
    means = be.empty(nvec + 1)
    ptrs = np.array([v.data_ptr() for v in V], dtype=np.uint64)
    be.call("mnk_vec_means_fwd", ptrs.ctypes.data, nvec, length, means)
    ref = torch.stack([v.double().mean() for v in vecs])
    gm, gt = torch.randn(nvec, generator=g), torch.randn(1, generator=g)
    GV = be.empty(nvec, length).fill_(float("nan"))
    be.call("mnk_vec_means_bwd", be.t(gm), be.t(gt), nvec, length, GV)
    GV2 = be.empty(nvec, length)
    be.call("mnk_vec_means_bwd", None, be.t(gt), nvec, length, GV2)
    be.sync()
    assert maxerr(means.cpu()[:nvec], ref) < 1e-6 * (1 + float(ref.abs().max()))
    assert abs(float(means.cpu()[nvec]) - float(ref.sum())) < 1e-5 * (1 + float(ref.abs().sum()))
    want = ((gm.double() + gt.double()) / length).unsqueeze(1).expand(nvec, length)
    assert maxerr(GV.cpu(), want) < 1e-6
    assert maxerr(GV2.cpu(), (gt.double() / length).expand(nvec, length)) < 1e-6
    # the autograd wrapper: same gradients as the stock formulation
    leaves = [be.t(v.clone()).requires_grad_(True) for v in vecs]
    m, tot = ops.LossMeansFn.apply(*leaves)
    (tot * 2.0 + (m * be.t(gm)).sum()).backward()
    leaves2 = [v.detach().double().requires_grad_(True) for v in vecs]
    m2 = torch.stack([v.mean() for v in leaves2])
    (m2.sum() * 2.0 + (m2 * gm.double()).sum()).backward()
    be.sync()
    for a, b in zip(leaves, leaves2):
        assert maxerr(a.grad.cpu(), b.grad) < 1e-6


def test_feature_matching_tap_adds_the_next_blocks_gradient(be):
    """ops.PairL1TapFn: the map goes on to the next block unchanged, and its backward is the L1 gradient PLUS the gradient the
    next block sends (mnk_pair_l1_bwd_add) -- what autograd's accumulation of two consumers gives."""
    from mnk import ops
    b, c, h, w = 2, 10, 6, 5
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2 * b, c, h, w, generator=g)
    nxt = torch.randn(2 * b, h, w, ceil4(c), generator=g)
    nxt[..., c:] = 0
    gl = torch.randn(b, generator=g)
    A = be.t(to_nhwc(x)).requires_grad_(True)
    act, loss = ops.PairL1TapFn.apply(A, c, b, 3.0)
    assert torch.equal(act.detach().cpu(), A.detach().cpu())
    ((act * be.t(nxt)).sum() + (loss * be.t(gl)).sum()).backward()
    A2 = be.t(to_nhwc(x)).requires_grad_(True)
    loss2 = ops.PairL1Fn.apply(A2, c, b, 3.0)
    ((A2 * be.t(nxt)).sum() + (loss2 * be.t(gl)).sum()).backward()
    be.sync()
    assert torch.equal(loss.detach().cpu(), loss2.detach().cpu())
    assert maxerr(A.grad.cpu(), A2.grad.cpu()) < 1e-6
    # only the next block's gradient (the discriminator-loss backward): handed through untouched
    A3 = be.t(to_nhwc(x)).requires_grad_(True)
    act3, _ = ops.PairL1TapFn.apply(A3, c, b, 3.0)
    (act3 * be.t(nxt)).sum().backward()
    be.sync()
    assert torch.equal(A3.grad.cpu(), nxt)
