"""Module-level parity of the drop-in `modules.*` (HIP kernels) against the goldens produced by the REAL reference
(tests/golden/*.pt, made by oracle/make_golden.py).  Tolerances are stated relative to the spread between the
reference run in fp32 and in fp64 (the reference's own rounding noise): the HIP path must be as close to the fp64
reference as the fp32 reference is, within a small factor."""
import copy
import os

import pytest
import torch

from oracle import cases
from mnk import knobs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)


def build(cfg, seed=0):
    """run.py:50-62 construction order -> identical RNG draws as the reference."""
    from modules.generator import MotionTransferGenerator
    from modules.discriminator import Discriminator
    from modules.keypoint_detector import KPDetector
    mp = cfg["model_params"]
    torch.manual_seed(seed)
    gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"])
    disc = Discriminator(**mp["discriminator_params"], **mp["common_params"])
    kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"])
    return gen, disc, kpd


@pytest.mark.parametrize("name", ["shapes", "taichi", "moving-gif", "bair", "vox", "tiny"])
def test_init_matches_reference_rng_order(name):
    """Same seed => same initial weights and the same state_dict keys as the reference (checkpoint contract)."""
    gold = load(name)
    gen, disc, kpd = build(gold["cfg"])
    for key, m in (("generator", gen), ("discriminator", disc), ("kp_detector", kpd)):
        s = float(sum(v.double().abs().sum() for v in m.state_dict().values()))
        assert abs(s - gold["init_sums"][key]) <= 1e-9 * max(1.0, gold["init_sums"][key]), key
    if "state" in gold:
        for key, m in (("generator", gen), ("discriminator", disc), ("kp_detector", kpd)):
            sd = m.state_dict()
            assert list(sd.keys()) == list(gold["state"][key].keys())
            for k in sd:
                assert sd[k].shape == gold["state"][key][k].shape, k


def run_case(be, gold, train, backward):
    cfg = gold["cfg"]
    gen, disc, kpd = build(cfg)
    if "state" in gold:
        gen.load_state_dict(gold["state"]["generator"])
        kpd.load_state_dict(gold["state"]["kp_detector"])
    else:   # weights are reproduced from the seed + the shared deterministic perturbation
        for i, m in enumerate((gen, disc, kpd)):
            sd = m.state_dict()
            cases.perturb_state_dict(sd, 7 + i)
            m.load_state_dict(sd)
    gen.to(be.device).train(train)
    kpd.to(be.device).train(train)
    src, drv = (cases.smooth_pair if gold["smooth"] else cases.synthetic_pair)(gold["batch"], gold["size"], gold["size"])
    src, drv = be.t(src), be.t(drv)
    kp = kpd(torch.cat([src, drv], dim=2))
    res = gen(src, kp_driving={k: v[:, 1:] for k, v in kp.items()}, kp_source={k: v[:, :1] for k, v in kp.items()})
    out = {"kp_mean": kp["mean"], "kp_var": kp["var"], "video_prediction": res["video_prediction"],
           "video_deformed": res["video_deformed"]}
    grads = None
    if backward:
        r1, r2 = gold["loss_weights"]
        loss = (res["video_prediction"] * be.t(r1)).sum() + (res["video_deformed"] * be.t(r2)).sum()
        loss.backward()
        grads = {"generator": {k: p.grad.cpu() for k, p in gen.named_parameters() if p.grad is not None},
                 "kp_detector": {k: p.grad.cpu() for k, p in kpd.named_parameters() if p.grad is not None}}
    be.sync()
    return {k: v.detach().cpu() for k, v in out.items()}, grads, gen, kpd


def check_outputs(out, gold, mode, factor=4.0, floor=2e-6):
    for k in ("kp_mean", "kp_var", "video_prediction", "video_deformed"):
        ref64 = gold[mode + "64"][k].double()
        if mode + "_spread" in gold:                                        # compact goldens store the spread itself
            spread = gold[mode + "_spread"][k]
        else:
            spread = float((gold[mode][k].double() - ref64).abs().max())   # reference fp32 vs reference fp64
        err = float((out[k].double() - ref64).abs().max())
        assert err <= factor * spread + floor, "%s.%s: |hip - ref64| = %.3e, reference's own fp32 noise %.3e" % (
            mode, k, err, spread)
    # reconstruction L1 criterion of BASELINE.md: |mean abs error difference| <= 1e-4
    l1 = float((out["video_prediction"].double() - gold[mode + "64"]["video_prediction"].double()).abs().mean())
    assert l1 < 1e-4


def check_grads(grads, gold, factor=6.0, floor=1e-4):
    worst = []
    for m in ("generator", "kp_detector"):
        for k, g64 in gold["grad64"][m].items():
            if cases.is_noise_bias(k):
                continue
            ref_spread = gold["grad_ref32_vs_ref64_rel"][m][k]
            g = grads[m][k].double()
            err = float((g - g64.double()).norm() / (g64.double().norm() + 1e-6))
            worst.append((err / (factor * ref_spread + floor), m, k, err, ref_spread))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, "gradient %s.%s: rel err %.3e vs the reference's own fp32 noise %.3e" % worst[0][1:]
    return worst


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
def test_tiny_forward_backward_train(be, name):
    gold = load(name)
    out, grads, gen, kpd = run_case(be, gold, train=True, backward=True)
    check_outputs(out, gold, "train")
    for m in ("generator", "kp_detector"):     # a bias in front of a training-mode BatchNorm may get no gradient (exactly zero)
        assert all(cases.is_noise_bias(k) for k in set(grads[m]) ^ set(gold["grad64"][m])), m
    check_grads(grads, gold)
    # running statistics after one training forward (batchnorm.py:119-123)
    for m, mod in (("generator", gen), ("kp_detector", kpd)):
        sd = mod.state_dict()
        for k, v in gold["running_after_train"][m].items():
            assert float((sd[k].cpu() - v).abs().max()) < 1e-4 * (1 + float(v.abs().max())), k


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
def test_tiny_forward_eval(be, name):
    gold = load(name)
    with torch.no_grad():
        out, _, _, _ = run_case(be, gold, train=False, backward=False)
    check_outputs(out, gold, "eval", factor=8.0, floor=5e-6)


def test_reference_config_shapes_on_the_emulator():
    """config/shapes.yaml (BASELINE configs[0], the reference's own CPU-runnable case) at 64x64, batch 2, weights
    rebuilt from the seed: training-mode outputs and every parameter gradient against the golden recorded from the real
    reference -- on the CPU emulator build of the kernels (the other YAML configs take minutes there; they run on the
    MI355X below)."""
    from conftest import Backend
    be = Backend("emu")
    gold = load("shapes")
    out, grads, _, _ = run_case(be, gold, train=True, backward=True)
    check_outputs(out, gold, "train")
    check_grads(grads, gold, factor=8.0, floor=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["shapes", "taichi", "moving-gif", "bair", "vox"])
def test_reference_configs_on_gpu(name):
    """The reference's own YAML configs (64x64, batch 2), weights rebuilt from the seed, against the goldens."""
    from conftest import Backend
    be = Backend("hip")
    gold = load(name)
    out, grads, _, _ = run_case(be, gold, train=True, backward=True)
    check_outputs(out, gold, "train")
    check_grads(grads, gold, factor=8.0, floor=1e-3)
    with torch.no_grad():
        out, _, _, _ = run_case(be, gold, train=False, backward=False)
    check_outputs(out, gold, "eval", factor=8.0, floor=5e-6)


def test_weights_updated_without_version_bump_are_seen(be):
    """Fused optimisers (torch.optim.Adam(fused=True), the default of mnk.engine.TrainStep on the GPU) write the
    parameters without bumping Tensor._version.  Neither the training forward (packs per call) nor the no-grad
    forward (cache keyed by version AND optimiser epoch) may keep using the old packed weights."""
    from mnk import ops
    torch.manual_seed(5)
    import torch.nn.functional as F
    w = torch.nn.Parameter(be.t(torch.randn(9, 6, 1, 3, 3) * 0.2))
    x = torch.rand(2, 6, 1, 8, 8)
    xa = ops.to_act(be.t(x))

    def run(grad):
        with torch.enable_grad() if grad else torch.no_grad():
            y, _ = ops.conv3x3(xa, 6, w)
        return ops.from_act(y.detach(), 9, 2).cpu()

    def ref():
        return F.conv2d(x[:, :, 0].double(), w.detach().cpu()[:, :, 0].double(), padding=1).float().unsqueeze(2)

    for grad in (True, False):
        assert float((run(grad) - ref()).abs().max()) < 1e-5
        ver = w._version
        w.data.mul_(-1.5)                      # what a fused optimiser step looks like to the version counter
        assert w._version == ver
        ops.invalidate_packed_weights()        # what the optimiser-step hook / TrainStep replay do
        assert float((run(grad) - ref()).abs().max()) < 1e-5, "stale packed weights (grad=%s)" % grad
    # and the hook itself: any optimiser step invalidates
    e0 = ops._PACK_EPOCH[0]
    opt = torch.optim.SGD([w], lr=0.1)
    w.grad = torch.zeros_like(w)
    opt.step()
    assert ops._PACK_EPOCH[0] > e0


def test_repack_registered_serves_training_forwards(be):
    """mnk.ops.repack_registered() (start of a mnk.engine.TrainStep iteration) packs every conv parameter seen so far
    in one launch; training forwards then use those buffers without a per-layer pack -- until the next optimiser step.
    Forward AND both gradients must match torch on the re-packed weights."""
    from mnk import ops
    import torch.nn.functional as F
    torch.manual_seed(6)
    ws = [torch.nn.Parameter(be.t(torch.randn(co, ci, 1, 3, 3) * 0.2)) for co, ci in ((9, 6), (5, 9), (20, 5))]
    x = torch.rand(2, 6, 1, 8, 8)

    def run():
        a = ops.to_act(be.t(x)).requires_grad_(True)
        h, c = a, 6
        for w in ws:
            h, _ = ops.conv3x3(h, c, w)
            c = w.shape[0]
        (h * h).sum().backward()
        out = (ops.from_act(h.detach(), c, 2).cpu(), a.grad.cpu().clone(), [w.grad.cpu().clone() for w in ws])
        for w in ws:
            w.grad = None
        return out

    def ref():
        xr = x[:, :, 0].double().requires_grad_(True)
        wr = [w.detach().cpu()[:, :, 0].double().requires_grad_(True) for w in ws]
        h = xr
        for w in wr:
            h = F.conv2d(h, w, padding=1)
        (h * h).sum().backward()
        return h.detach(), xr.grad, [w.grad for w in wr]

    def check(tag):
        y, dx, dws = run()
        ry, rdx, rdws = ref()
        def rel(a, b):
            return float((a - b).abs().max() / b.abs().max())
        assert rel(y[:, :, 0], ry) < 1e-5, tag
        assert rel(dx.permute(0, 3, 1, 2)[:, :6], rdx) < 1e-5, tag
        for a, b in zip(dws, rdws):
            assert rel(a[:, :, 0], b) < 1e-5, tag

    launches = []
    real = ops._call

    def counting(name, *a, **k):
        launches.append(name)
        return real(name, *a, **k)

    check("first use: per-layer packs register the parameters")
    for w in ws:
        w.data.mul_(-0.7)                      # a fused optimiser step: no version bump ...
    ops.invalidate_packed_weights()            # ... but the step hook fires
    ops._call = counting
    try:
        assert ops.repack_registered()
        check("after repack_registered")
        assert launches.count("mnk_conv3x3_pack_multi") == 1 and "mnk_conv3x3_pack_all" not in launches
        del launches[:]
        for w in ws:
            w.data.add_(0.05)
        ops.invalidate_packed_weights()        # stale again, and nobody calls repack_registered() (a user-owned loop: the
        check("the first stale layer re-packs every registered parameter")      # reference's train.py on these modules)
        assert launches.count("mnk_conv3x3_pack_multi") == 1 and "mnk_conv3x3_pack_all" not in launches
    finally:
        ops._call = real


def test_evaluation_coefficients_of_a_norm_layer_are_kept_until_something_writes_it(be):
    """The reference's evaluation loops call the networks frame by frame with constant weights (reconstruction.py:52-62): the
    (mean, invstd, scale) launch of an eval-mode norm layer runs once, not per call -- and again after ANY write to the layer:
    an in-place change of a buffer, load_state_dict, an optimiser step (fused steps do not bump tensor versions: the global
    post-step hook does it), a training-mode forward (kernels update the running statistics through raw pointers)."""
    from mnk import ops
    from modules.util import SameBlock3D
    torch.manual_seed(3)
    blk = SameBlock3D(6, 6, groups=1, kernel_size=(1, 1, 1), padding=(0, 0, 0)).to(be.device)
    blk.norm.running_mean.uniform_(-0.5, 0.5), blk.norm.running_var.uniform_(0.5, 2.0)
    x = be.t(torch.rand(2, 6, 1, 8, 8))
    launches = []
    real = ops._call

    def counting(name, *a, **k):
        launches.append(name)
        return real(name, *a, **k)

    def ref():
        bn = blk.norm
        y = torch.nn.functional.conv3d(x.cpu().double(), blk.conv.weight.detach().cpu().double(), blk.conv.bias.detach().cpu().double())
        y = (y - bn.running_mean.cpu().double().view(1, -1, 1, 1, 1)) / torch.sqrt(bn.running_var.cpu().double().view(1, -1, 1, 1, 1) + bn.eps)
        return torch.relu(y * bn.weight.detach().cpu().double().view(1, -1, 1, 1, 1) + bn.bias.detach().cpu().double().view(1, -1, 1, 1, 1))

    def run(expect):
        del launches[:]
        with torch.no_grad():
            out = blk(x)
        be.sync()
        assert launches.count("mnk_bn_eval_coeffs") == expect, (expect, launches)
        assert float((out.cpu().double() - ref()).abs().max()) < 1e-5

    ops._call = counting
    try:
        blk.eval()
        run(1)
        run(0), run(0)
        blk.norm.running_mean.add_(0.25)                       # an in-place write: the tensor's version moves
        run(1), run(0)
        sd = {k: v.clone() for k, v in blk.state_dict().items()}
        sd["norm.running_var"] *= 1.5
        blk.load_state_dict(sd)
        run(1), run(0)
        opt = torch.optim.SGD(blk.parameters(), lr=0.1)
        for p in blk.parameters():
            p.grad = torch.ones_like(p)
        opt.step()                                             # the norm layer's gamma changed
        run(1), run(0)
        blk.train()
        with torch.no_grad():
            blk(x)                                             # updates the running statistics on the device
        blk.eval()
        run(1), run(0)
    finally:
        ops._call = real


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
def test_pad_channels_are_written(be, name, monkeypatch):
    """The 3x3 fast loader (MNK_CONV_CLEAN_PADS) relies on every activation this package produces having ZERO pad
    channels.  Here every torch.empty / empty_like of mnk.ops is pre-filled with NaN: a producer that leaves its pad
    channels (or anything else a consumer reads) unwritten turns the losses / gradients into NaN."""
    from mnk import ops

    def nan_empty(*a, **k):
        t = torch.zeros(*a, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t

    def nan_empty_like(x, **k):
        t = torch.zeros_like(x, **k)
        return t.fill_(float("nan")) if t.is_floating_point() else t

    class TorchProxy:
        def __getattr__(self, item):
            if item == "empty":
                return nan_empty
            if item == "empty_like":
                return nan_empty_like
            return getattr(torch, item)

    monkeypatch.setattr(ops, "torch", TorchProxy())
    ops.SCRATCH.bufs.clear()
    gold = load(name)
    out, grads, _, _ = run_case(be, gold, train=True, backward=True)
    ops.SCRATCH.bufs.clear()
    for k, v in out.items():
        assert torch.isfinite(v).all(), "non-finite output %s" % k
    for grp, d in grads.items():
        for k, v in d.items():
            assert torch.isfinite(v).all(), "non-finite gradient %s.%s" % (grp, k)


def test_down_block_with_fused_statistics(be):
    """A block large enough that the conv is not split along K, so the BatchNorm statistics come out of the conv
    epilogue (mnk_conv3x3_stats_floats > 0); forward, running stats and all gradients against the oracle in fp64."""
    from modules.util import DownBlock3D
    from oracle import restate
    torch.manual_seed(3)
    blk = DownBlock3D(5, 12, kernel_size=(1, 3, 3), padding=(0, 1, 1))
    with torch.no_grad():
        blk.norm.weight.add_(0.3 * torch.randn(12))
        blk.norm.bias.add_(0.3 * torch.randn(12))
    sd = {("blk." + k): v.detach().clone().double() for k, v in blk.state_dict().items()}
    x = torch.rand(6, 5, 1, 64, 64)
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    ctx = restate.Ctx(sd, True)
    ref = restate.down_block(ctx, restate.fold(x.double()), "blk")
    g = torch.randn(ref.shape, dtype=torch.float64)
    (ref * g).sum().backward()
    blk.to(be.device).train()
    out = blk(be.t(x))
    (out * be.t(restate.unfold(g.float(), 6))).sum().backward()
    be.sync()
    assert float((out.detach().cpu()[:, :, 0].double() - ref).abs().max()) < 2e-5
    for k in ("conv.weight", "norm.weight", "norm.bias"):
        a, b = dict(blk.named_parameters())[k].grad.cpu().double(), sd["blk." + k].grad
        assert float((a - b).norm() / b.norm()) < 1e-4, k
    assert float((blk.norm.running_var.cpu().double() - ctx.new_stats["blk.norm.running_var"]).abs().max()) < 1e-5


@pytest.mark.parametrize("kind,cfg_name,size", [("emu", "tiny", 32), pytest.param("hip", "tiny", 32, marks=pytest.mark.gpu),
                                                pytest.param("hip", "taichi", 64, marks=pytest.mark.gpu)])
def test_discriminator_matches_oracle(make_backend, kind, cfg_name, size):
    """modules.discriminator: the gfx950-kernel Discriminator (4x4 no-pad convs, InstanceNorm, LeakyReLU, avg-pool, 1x1
    head) against oracle/restate.py::discriminator_forward in fp64: every returned feature map and all gradients
    (parameters, input frame, key-points)."""
    import modules.discriminator as md
    from mnk import discriminator_hip
    Discriminator = md.Discriminator
    assert md.Discriminator is discriminator_hip.Discriminator, "the drop-in name must resolve to the gfx950-kernel class"
    from oracle import restate
    # (explicit backend list: no gpu-marked instance touches the CPU emulator, so a GPU-box run maps libmonkeynet_hip.so only)
    be = make_backend(kind)
    cfg = load(cfg_name)["cfg"]
    mp = cfg["model_params"]
    common = mp["common_params"]
    torch.manual_seed(5)
    disc = Discriminator(**mp["discriminator_params"], **common)
    sd = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    cases.perturb_state_dict(sd, 11)
    disc.load_state_dict(sd)
    b = 2
    g = torch.Generator().manual_seed(4)
    x = torch.rand(b, 3, 1, size, size, generator=g)
    kd, ks = cases.random_kp(b, 1, common["num_kp"], seed=1), cases.random_kp(b, 1, common["num_kp"], seed=2)
    # oracle, fp64
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x64 = x.double().requires_grad_(True)
    kd64 = {k: v.double().requires_grad_(True) for k, v in kd.items()}
    ks64 = {k: v.double() for k, v in ks.items()}
    maps64 = restate.discriminator_forward(sd64, mp["discriminator_params"], common, x64, kd64, ks64)
    ws = [torch.randn(m.shape, generator=g, dtype=torch.float64) for m in maps64]
    sum((m * w).sum() for m, w in zip(maps64[1:], ws[1:])).backward()
    # HIP
    disc.to(be.device)
    xh = be.t(x).requires_grad_(True)
    kdh = {k: be.t(v).requires_grad_(True) for k, v in kd.items()}
    ksh = {k: be.t(v) for k, v in ks.items()}
    maps = disc(xh, kdh, ksh)
    sum((m * be.t(w.float())).sum() for m, w in zip(maps[1:], ws[1:])).backward()
    be.sync()
    assert len(maps) == len(maps64)
    for i, (a, r) in enumerate(zip(maps, maps64)):
        assert a.shape == r.shape
        assert float((a.detach().cpu().double() - r.detach()).abs().max()) < 2e-5 * (1 + float(r.abs().max())), i
    for k, p in disc.named_parameters():
        ref = sd64[k].grad
        if k.endswith("conv.bias") and "down_blocks.0" not in k and k != "conv.bias":
            continue      # bias in front of an InstanceNorm: analytically zero gradient (noise in the reference too)
        err = float((p.grad.cpu().double() - ref).norm() / (ref.norm() + 1e-9))
        assert err < 2e-3, (k, err)   # InstanceNorm over 2x2..5x5 maps amplifies fp32 rounding
    assert float((xh.grad.cpu().double() - x64.grad).norm() / x64.grad.norm()) < 2e-3
    assert float((kdh["mean"].grad.cpu().double() - kd64["mean"].grad).norm() / kd64["mean"].grad.norm()) < 2e-3


def test_discriminate_pair_batched_equals_two_calls(be, monkeypatch):
    """mnk.engine.discriminate_pair: D(fake) and D(real) as one pass over [fake; real] (every layer of the
    discriminator is per sample) == the reference's two calls (train.py:43-45): feature maps and every gradient."""
    import modules.discriminator as md
    from mnk import engine
    cfg = load("tiny")["cfg"]
    mp = cfg["model_params"]
    common = mp["common_params"]
    torch.manual_seed(8)
    disc = md.Discriminator(**mp["discriminator_params"], **common)
    sd = {k: v.detach().clone() for k, v in disc.state_dict().items()}
    cases.perturb_state_dict(sd, 13)
    disc.load_state_dict(sd)
    disc.to(be.device)
    g = torch.Generator().manual_seed(9)
    fake0, real0 = torch.rand(3, 3, 1, 32, 32, generator=g), torch.rand(3, 3, 1, 32, 32, generator=g)
    kd, ks = cases.random_kp(3, 1, common["num_kp"], seed=3), cases.random_kp(3, 1, common["num_kp"], seed=4)
    ws = None

    def run(batched):
        nonlocal ws
        monkeypatch.setitem(knobs.FORMS, "DISC_BATCHED", bool(batched))
        fake = be.t(fake0.clone()).requires_grad_(True)         # (be.t is the identity on the emulator backend)
        kp = {"kp_driving": {k: be.t(v.clone()).requires_grad_(True) for k, v in kd.items()},
              "kp_source": {k: be.t(v) for k, v in ks.items()}}
        mf, mr = engine.discriminate_pair(disc, fake, be.t(real0), kp)
        if ws is None:
            ws = [torch.randn(m.shape, generator=g) for m in mf]
        sum((a * be.t(w)).sum() + (b * be.t(w)).sum() * 0.5 for a, b, w in zip(mf, mr, ws)).backward()
        be.sync()
        out = ([m.detach().cpu() for m in mf + mr], fake.grad.cpu().clone(), kp["kp_driving"]["mean"].grad.cpu().clone(),
               {k: p.grad.cpu().clone() for k, p in disc.named_parameters()})
        disc.zero_grad()
        return out

    two, one = run(False), run(True)
    for a, b in zip(one[0], two[0]):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-5 * (1 + float(b.abs().max()))
    for a, b in ((one[1], two[1]), (one[2], two[2])):
        assert float((a - b).norm() / b.norm()) < 1e-4
    for k in two[3]:
        if k.endswith("conv.bias") and "down_blocks.0" not in k and k != "conv.bias":
            continue      # bias in front of an InstanceNorm: rounding noise around an analytically zero gradient
        assert float((one[3][k] - two[3][k]).norm() / (two[3][k].norm() + 1e-12)) < 1e-3, k


@pytest.mark.parametrize("hw", [(4, 4), (3, 3), (5, 6), (1, 1)])
def test_down_block_on_small_and_odd_maps(be, hw):
    """A small layer whose convolution is split along K (136 input channels): even maps take the one-launch BatchNorm that
    sums the partials; odd maps are refused with the pooling kernel's own message (H % 2 == 0 && W % 2 == 0) -- the
    reference's hourglass cannot pool an odd map either (the decoder's torch.cat with the skip then fails, modules/util.py:185)
    -- and not, as before, with "split-K partials were deferred to a BatchNorm that does not take the small-layer path"."""
    from modules.util import DownBlock3D
    from oracle import restate
    from mnk import _lib
    h, w = hw
    torch.manual_seed(5)
    blk = DownBlock3D(136, 24, kernel_size=(1, 3, 3), padding=(0, 1, 1))
    sd = {("blk." + k): v.detach().clone().double() for k, v in blk.state_dict().items()}
    x = torch.rand(4, 136, 1, h, w)
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    ctx = restate.Ctx(sd, True)
    blk.to(be.device).train()
    if h % 2 or w % 2:
        with pytest.raises(_lib.MnkError, match="H % 2 == 0"):
            blk(be.t(x))
        from mnk import ops
        assert not ops._SPLIT_PENDING
        return
    ref = restate.down_block(ctx, restate.fold(x.double()), "blk")
    g = torch.randn(ref.shape, dtype=torch.float64)
    (ref * g).sum().backward()
    out = blk(be.t(x))
    (out * be.t(restate.unfold(g.float(), 4))).sum().backward()
    be.sync()
    assert float((out.detach().cpu()[:, :, 0].double() - ref).abs().max()) < 2e-5
    for k in ("conv.weight", "norm.weight", "norm.bias"):
        a, b = dict(blk.named_parameters())[k].grad.cpu().double(), sd["blk." + k].grad
        assert float((a - b).norm() / b.norm()) < 1e-4, k


@pytest.mark.parametrize("fused", ["1", "0"], ids=["skip-gradient-in-the-norm-pass", "autograd-accumulates"])
@pytest.mark.parametrize("hw", [24, 8], ids=["general-kernels", "one-launch-norm"])
def test_residual_blocks_and_the_gradient_of_their_skip_path(be, monkeypatch, fused, hw):
    """conv (with bias) -> ResBlock3D -> ResBlock3D in training mode (modules/util.py:45-68) against the oracle in fp64: the
    output, the gradient of the input and of EVERY parameter -- among them the biases of the convolutions that produce a
    block's input, whose gradient is the column sum of (BatchNorm backward + skip gradient) and comes out of the next
    block's norm pass when the skip gradient is added there (MNK_RES_SKIP_FUSED, ops.BNActSkipFn)."""
    from torch import nn
    from modules.util import ResBlock3D
    from mnk import ops
    from oracle import restate
    monkeypatch.setitem(knobs.FORMS, "RES_SKIP_FUSED", fused == "1")
    torch.manual_seed(5)
    cin, c, n = 3, 5, 2
    front = nn.Conv3d(cin, c, kernel_size=(1, 3, 3), padding=(0, 1, 1))
    blocks = [ResBlock3D(c, kernel_size=(1, 3, 3), padding=(0, 1, 1)) for _ in range(2)]
    with torch.no_grad():
        for b in blocks:
            for norm in (b.norm1, b.norm2):
                norm.weight.add_(0.3 * torch.randn(c))
                norm.bias.add_(0.3 * torch.randn(c))
    sd = {"front." + k: v.detach().clone().double() for k, v in front.state_dict().items()}
    for i, b in enumerate(blocks):
        sd.update({"r%d.%s" % (i, k): v.detach().clone().double() for k, v in b.state_dict().items()})
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    x = torch.rand(n, cin, 1, hw, hw)
    xd = restate.fold(x.double()).requires_grad_(True)
    ctx = restate.Ctx(sd, True)
    ref = restate.conv(ctx, xd, "front")
    for i in range(2):
        ref = restate.res_block(ctx, ref, "r%d" % i)
    g = torch.randn(ref.shape, dtype=torch.float64)
    (ref * g).sum().backward()

    front.to(be.device)
    for b in blocks:
        b.to(be.device).train()
    X = be.t(x).requires_grad_(True)
    act, s = ops.conv3x3(ops.to_act(X), cin, front.weight, front.bias, want_stats=True)
    act, _, s = blocks[0].forward_act(act, c, x_sums=s, want_stats=True)
    act, _ = blocks[1].forward_act(act, c, x_sums=s)
    out = ops.from_act(act, c, n)
    (out * be.t(restate.unfold(g.float(), n))).sum().backward()
    be.sync()
    assert float((out.detach().cpu()[:, :, 0].double() - ref).abs().max()) < 2e-5
    assert float((X.grad.cpu()[:, :, 0].double() - xd.grad).norm() / xd.grad.norm()) < 1e-4
    named = [("front." + k, p) for k, p in front.named_parameters()]
    for i, b in enumerate(blocks):
        named += [("r%d.%s" % (i, k), p) for k, p in b.named_parameters()]
    for k, p in named:
        if k in ("r0.conv1.bias", "r1.conv1.bias"):      # in front of a training BatchNorm: exactly zero (no pass is spent)
            assert p.grad is None or float(p.grad.abs().max()) < 1e-3 * float(sd[k].grad.abs().max() + 1)
            continue
        a, b = p.grad.cpu().double().reshape(-1), sd[k].grad.reshape(-1)
        assert float((a - b).norm() / (b.norm() + 1e-12)) < 2e-4, k


@pytest.mark.parametrize("fused", ["1", "0"], ids=["skip-gradient-in-the-dgrad-epilogue", "autograd-accumulates"])
def test_hourglass_levels_with_two_consumers(be, monkeypatch, fused):
    """Hourglass (modules/util.py:129-203) on an input that needs a gradient, against the oracle in fp64: output, input gradient
    and every parameter gradient.  Every encoder level feeds the next down block AND the decoder; with MNK_SKIP_GRAD_FUSED the
    down block's convolution hands its input through (ops.Conv3x3SkipFn) and its data-gradient launch adds the decoder's
    gradient as its residual operand."""
    from modules.util import Hourglass
    from oracle import restate
    monkeypatch.setitem(knobs.FORMS, "SKIP_GRAD_FUSED", fused == "1")
    torch.manual_seed(9)
    hg = Hourglass(block_expansion=8, in_features=6, out_features=5, num_blocks=3, max_features=32)
    sd = {"hg." + k: v.detach().clone().double() for k, v in hg.state_dict().items()}
    for t in sd.values():
        if t.is_floating_point():
            t.requires_grad_(True)
    n, hw = 3, 32
    x = torch.rand(n, 6, 1, hw, hw)
    xd = restate.fold(x.double()).requires_grad_(True)
    ctx = restate.Ctx(sd, True)
    ref = restate.hourglass(ctx, xd, "hg", 3)
    g = torch.randn(ref.shape, dtype=torch.float64)
    (ref * g).sum().backward()
    hg.to(be.device).train()
    X = be.t(x).requires_grad_(True)
    out = hg(X)
    (out * be.t(restate.unfold(g.float(), n))).sum().backward()
    be.sync()
    assert float((out.detach().cpu()[:, :, 0].double() - ref).abs().max()) < 5e-5
    assert float((X.grad.cpu()[:, :, 0].double() - xd.grad).norm() / xd.grad.norm()) < 2e-4
    for k, p in hg.named_parameters():
        r = sd["hg." + k].grad
        if p.grad is None:          # a convolution bias in front of a training BatchNorm: exactly zero, no pass is spent
            assert k.endswith("conv.bias") and float(r.abs().max()) < 1e-6 * float(g.abs().sum())
            continue
        a = p.grad.cpu().double().reshape(-1)
        assert float((a - r.reshape(-1)).norm() / (r.norm() + 1e-12)) < 5e-4, k


def test_norm_layers_take_their_backward_statistics_from_the_data_gradient_launch(be, monkeypatch):
    """round 4 (verdict r3 item 6a): a non-pooling BatchNorm whose output feeds ONE convolution gets sum g, sum g * xhat from
    that convolution's data-gradient epilogue (ops._DZ_STATS) instead of a pass of its own over (y, dz).  Same gradients as
    with the form switched off, and the hand-over is really used (decoder up-blocks, residual blocks)."""
    from mnk import ops
    from modules.util import Hourglass, ResBlock3D
    torch.manual_seed(11)
    hg = Hourglass(block_expansion=16, in_features=3, out_features=5, max_features=64, num_blocks=2)
    r1, r2 = ResBlock3D(21, kernel_size=(1, 3, 3), padding=(0, 1, 1)), ResBlock3D(21, kernel_size=(1, 3, 3), padding=(0, 1, 1))
    # every level above the one-launch small-layer path (<= 512 pixel rows): 8 x 8 maps x 9 frames = 576 rows at the bottom of
    # the hourglass on the emulator, 24 frames on the GPU
    nb = 24 if be.kind == "hip" else 9
    x = torch.rand(nb, 3, 1, 32, 32)
    rs = 32 if be.kind == "hip" else 16                  # (the residual blocks: 16 x 16 x 9 = 2 304 rows on the emulator)
    xr = torch.rand(nb, 21, 1, rs, rs)
    w1, w2 = torch.randn(nb, 5, 1, 32, 32), torch.randn(nb, 21, 1, rs, rs)
    for m in (hg, r1, r2):
        m.to(be.device).train()

    def run(on):
        monkeypatch.setitem(knobs.FORMS, "DGRAD_BN_STATS", on)
        ops.DZ_STATS_COUNT[0] = ops.DZ_STATS_COUNT[1] = 0
        for m in (hg, r1, r2):
            m.zero_grad()
        xin = be.t(xr.clone()).requires_grad_(True)        # (be.t is the identity on the emulator: a fresh leaf per run)
        out = hg(be.t(x))
        res = r2(r1(xin))
        ((out * be.t(w1)).sum() + (res * be.t(w2)).sum()).backward()
        be.sync()
        grads = {"%d.%s" % (i, k): p.grad.detach().cpu().clone() for i, m in enumerate((hg, r1, r2)) for k, p in m.named_parameters()
                 if p.grad is not None}
        grads["x"] = xin.grad.cpu().clone()
        return grads, tuple(ops.DZ_STATS_COUNT)

    g_on, used = run(True)
    g_off, unused = run(False)
    assert used[0] >= 4, used                       # the hourglass decoder's up-block in front of the last one + the residual blocks' norms
    assert unused[0] == 0
    assert set(g_on) == set(g_off)
    for k in g_on:
        err = float((g_on[k] - g_off[k]).norm() / (g_off[k].norm() + 1e-12))
        assert err < 2e-5, (k, err)
