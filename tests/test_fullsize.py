"""Parity at BASELINE.json's full sizes (moving-gif parameters, batch 32 @ 64x64) through size-independent properties --
the oracle takes seconds per pair at this size, so it checks small cases (test_modules / test_step) and these
identities cover the large launch plans (un-split tiles, many pixel splits, tap-major / nine-tap weight gradients):

* adjoint identities of the convolution: <conv(u; w), g> = <u, dgrad(g; w)> and <conv(x; v), g> = <v, wgrad(x, g)>
  tie the forward, data-gradient and weight-gradient kernels to each other on every layer shape of the config;
* eval-mode generation is frame-independent (running BatchNorm statistics), so the batch-32 launch plan must give the
  frames of four batch-8 launches;
* one captured training iteration replayed as a hipGraph reproduces the eager iteration it was recorded from."""
import pytest
import torch

from oracle import cases

pytestmark = pytest.mark.gpu


def _layer_shapes():
    from mnk import configs, workload
    cfg = configs.get("moving-gif")
    seen, out = set(), []
    for name, cin, cout, h, w, k, _ in workload.conv_flops_hot_path(cfg, 64, 64)["layers"]:
        if k != 3:
            continue
        frames = 32 * (2 if name.startswith("kp") else 1)
        ups = ".dec" in name
        key = (cin, cout, h, w, frames, ups)
        if key not in seen:
            seen.add(key)
            out.append(pytest.param(key, id="%s-%dto%d@%d" % (name, cin, cout, h)))
    return out


@pytest.mark.parametrize("shape", _layer_shapes())
def test_conv_adjoint_identities_at_full_layer_sizes(shape):
    from mnk import ops
    cin, cout, h, w, frames, ups = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(hash(shape) % (2 ** 31))
    hs, ws = (h // 2, w // 2) if ups else (h, w)

    def act(c, hh, ww):      # an act with zero pad channels, as every producer of the package writes it
        t = torch.zeros(frames, hh, ww, ops.ceil4(c))
        t[..., :c] = torch.randn(frames, hh, ww, c, generator=g)
        return t.to(dev)

    x, u = act(cin, hs, ws).requires_grad_(True), act(cin, hs, ws)
    wgt = (torch.randn(cout, cin, 1, 3, 3, generator=g) * (1.0 / (3 * cin ** 0.5))).to(dev).requires_grad_(True)
    v = (torch.randn(cout, cin, 1, 3, 3, generator=g) * (1.0 / (3 * cin ** 0.5))).to(dev)
    gy = act(cout, h, w)
    y, _ = ops.conv3x3(x, cin, wgt, ups=ups)
    (y * gy).sum().backward()                         # x.grad = dgrad(gy; w), wgt.grad = wgrad(x, gy)
    with torch.no_grad():
        yu, _ = ops.conv3x3(u, cin, wgt, ups=ups)     # linear in the input ...
        yv, _ = ops.conv3x3(x.detach(), cin, v, ups=ups)   # ... and in the weights
    torch.cuda.synchronize()

    def dot(a, b):
        return float((a.double() * b.double()).sum())

    def close(a, b, scale):
        return abs(a - b) <= 2e-5 * scale

    n_u = (float(yu.double().pow(2).sum()) * float(gy.double().pow(2).sum())) ** 0.5
    n_v = (float(yv.double().pow(2).sum()) * float(gy.double().pow(2).sum())) ** 0.5
    assert close(dot(yu, gy), dot(u, x.grad), n_u), ("dgrad adjoint", dot(yu, gy), dot(u, x.grad))
    assert close(dot(yv, gy), dot(v, wgt.grad), n_v), ("wgrad adjoint", dot(yv, gy), dot(v, wgt.grad))
    assert torch.all(y.detach()[..., cout:] == 0) and torch.all(x.grad[..., cin:] == 0)


def _models(name="moving-gif"):
    from mnk import configs
    from test_modules import build
    gen, disc, kpd = build(configs.get(name))
    for i, m in enumerate((gen, disc, kpd)):
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 7 + i)
        m.load_state_dict(sd)
    dev = torch.device("cuda:0")
    return gen.to(dev), disc.to(dev), kpd.to(dev)


def test_eval_generation_is_frame_independent_at_batch_32():
    from mnk import engine
    gen, _, kpd = _models()
    src, drv = cases.smooth_pair(32, 64, 64)
    src, drv = src.cuda(), drv.cuda()
    rec = engine.Reconstructor(kpd, gen)
    full = rec(src, drv)
    parts = [rec(src[i:i + 8], drv[i:i + 8]) for i in range(0, 32, 8)]
    torch.cuda.synchronize()
    for k in ("video_prediction", "video_deformed", "kp_driving_mean"):
        joined = torch.cat([p[k] for p in parts], 0)
        assert float((full[k] - joined).abs().max()) < 2e-5, k


def _param_sample(mods):
    """A fixed 256-element sample of every parameter and buffer (fp32 bits)."""
    out = []
    for m in mods:
        for t in list(m.parameters()) + list(m.buffers()):
            flat = t.detach().reshape(-1)
            if flat.dtype != torch.float32:
                continue
            out.append(flat[:: max(1, flat.numel() // 256)][:256].clone())
    return torch.cat(out).cpu()


@pytest.mark.trajectory
@pytest.mark.parametrize("mnk_adam", [True, False], ids=["mnk-adam", "torch-adam-capturable"])
def test_full_size_training_iterations_graph_replay_equals_eager(mnk_adam):
    """Same initial weights and inputs: TrainStep(use_graph=True) runs three eager warm-up iterations, captures the
    iteration, puts parameters / running statistics / optimiser state back and replays -- so replay k must BE eager
    iteration k: the same kernels in the same order on the same data.  Every sum of the library has a fixed order (split-K
    and weight-gradient reductions, the statistics' two stages, and -- since round 5 -- the gather-form warp backward, which
    was the one place fp32 atomics chose the order), so there is no noise to allow for:
      * two eager runs are bit-identical (losses of every iteration, a sample of every parameter after every step);
      * replay 0 equals eager iteration 0 BIT FOR BIT (losses and the parameters after the three Adam steps);
      * replays 1-3 equal eager iterations 1-3 to 1e-5 relative (they are bit-equal too wherever the captured iteration and
        the eager one run the same reduction forms; the bound only leaves room for MnkAdam.tap_direct, which reads the same
        partial sums through another kernel)."""
    from mnk import engine, configs
    cfg = configs.get("moving-gif")
    src, drv = cases.synthetic_pair(32, 64, 64)
    x = {"source": src.cuda(), "video": drv.cuda()}

    def run(use_graph, n):
        gen, disc, kpd = _models()
        # the same Adam in both runs -- only the launch mechanism differs
        step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=mnk_adam, use_graph=True)
        losses, params = [], []
        for _ in range(n):
            g_l, d_l, _ = step.step(x) if use_graph else step._eager_step(x)
            torch.cuda.synchronize()
            losses.append(torch.stack([v.detach().reshape(()).float() for v in list(g_l) + list(d_l)]).cpu())
            params.append(_param_sample((gen, disc, kpd)))
        return losses, params

    def bits(t):
        return t.contiguous().view(torch.int32)

    def rel(p, q):
        return float(((p.double() - q.double()).abs() / q.double().abs().clamp_min(1.0)).max())

    (e_l, e_p), (e2_l, e2_p) = run(False, 4), run(False, 4)
    for k in range(4):
        assert torch.equal(bits(e_l[k]), bits(e2_l[k])), ("two eager runs differ in the losses of iteration", k, e_l[k], e2_l[k])
        assert torch.equal(bits(e_p[k]), bits(e2_p[k])), ("two eager runs differ in the parameters after iteration", k)
    g_l, g_p = run(True, 4)
    for k in range(4):
        assert bool(torch.isfinite(g_l[k]).all()) and float(g_l[k].abs().max()) < 1e6, ("non-finite loss in replay", k, g_l[k])
    assert torch.equal(bits(g_l[0]), bits(e_l[0])), ("replay 0 != eager iteration 0 (losses)", g_l[0], e_l[0])
    assert torch.equal(bits(g_p[0]), bits(e_p[0])), \
        ("replay 0 != eager iteration 0 (parameters after the step)", float((g_p[0] - e_p[0]).abs().max()))
    for k in range(1, 4):
        assert rel(g_l[k], e_l[k]) <= 1e-5, ("replay k != eager iteration k (losses)", k, g_l[k], e_l[k])
        assert rel(g_l[k], e_l[k]) < rel(g_l[k], e_l[k - 1]) or torch.equal(bits(g_l[k]), bits(e_l[k])), \
            "replay k must be iteration k: the warm-up updates are undone"
