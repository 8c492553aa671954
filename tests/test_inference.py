"""Eval-mode (reconstruction / transfer style) generation through mnk.engine.Reconstructor: batched frames, running
BatchNorm statistics, optional hipGraph replay (BASELINE config 5)."""
import pytest
import torch

from oracle import cases
from test_modules import build, load


def _models(gold, be):
    gen, disc, kpd = build(gold["cfg"])
    for i, m in enumerate((gen, disc, kpd)):
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 7 + i)
        m.load_state_dict(sd)
    return gen.to(be.device), kpd.to(be.device)


def test_reconstructor_matches_reference_eval_golden(be):
    """bair.yaml (norm_const 'sum') in eval mode: separate kp-detector calls for source and driving frames (as
    reconstruction.py:57-59 does) give the same frames as the joint call the golden was recorded with."""
    from mnk import engine
    name = "bair" if be.kind == "hip" else "tiny"
    gold = load(name)
    if name == "tiny":
        gen, disc, kpd = build(gold["cfg"])
        gen.load_state_dict(gold["state"]["generator"]), kpd.load_state_dict(gold["state"]["kp_detector"])
        gen.to(be.device), kpd.to(be.device)
    else:
        gen, kpd = _models(gold, be)
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
    out = engine.Reconstructor(kpd, gen)(be.t(src), be.t(drv))
    be.sync()
    ref = gold["eval64"]
    assert float((out["video_prediction"].cpu().double() - ref["video_prediction"].double()).abs().max()) < 2e-5
    assert float((out["video_deformed"].cpu().double() - ref["video_deformed"].double()).abs().max()) < 1e-4
    assert float((out["kp_driving_mean"].cpu().double() - ref["kp_mean"][:, 1:].double()).abs().max()) < 2e-6
    # reconstruction L1 criterion (reconstruction.py:74): |L1_hip - L1_ref| <= 1e-4
    l1_hip = float((out["video_prediction"].cpu().double() - drv.double()).abs().mean())
    l1_ref = float((ref["video_prediction"].double() - drv.double()).abs().mean())
    assert abs(l1_hip - l1_ref) < 1e-4


@pytest.mark.gpu
def test_hipgraph_replay_equals_eager_launches():
    from conftest import Backend
    from mnk import engine
    be = Backend("hip")
    gold = load("bair")
    gen, kpd = _models(gold, be)
    g = torch.Generator().manual_seed(3)
    eager = engine.Reconstructor(kpd, gen, use_graph=False)
    graphed = engine.Reconstructor(kpd, gen, use_graph=True)
    for it in range(3):
        src = torch.rand(16, 3, 1, 64, 64, generator=g).to(be.device)
        drv = torch.rand(16, 3, 1, 64, 64, generator=g).to(be.device)
        a = eager(src, drv)
        b = graphed(src, drv)
        torch.cuda.synchronize()
        assert torch.equal(a["video_prediction"], b["video_prediction"]), it   # same kernels, same order: bit-equal
        assert torch.equal(a["kp_driving_mean"], b["kp_driving_mean"])


@pytest.mark.gpu
def test_graphed_reconstructor_follows_a_second_checkpoint():
    """Evaluating another checkpoint with the SAME captured Reconstructor (load_state_dict between two replays): the
    replay must run on the new convolution weights together with the new normalisation parameters -- the captured
    forward re-packs from the live parameters (round-1 advisory: it replayed stale packed weights)."""
    from conftest import Backend
    from mnk import engine
    be = Backend("hip")
    gold = load("bair")
    gen, kpd = _models(gold, be)
    g = torch.Generator().manual_seed(4)
    src = torch.rand(8, 3, 1, 64, 64, generator=g).to(be.device)
    drv = torch.rand(8, 3, 1, 64, 64, generator=g).to(be.device)
    graphed = engine.Reconstructor(kpd, gen, use_graph=True)
    first = {k: v.clone() for k, v in graphed(src, drv).items()}
    for seed, m in ((31, gen), (32, kpd)):                   # "checkpoint 2": every tensor changes
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        cases.perturb_state_dict(sd, seed, scale=0.05)
        m.load_state_dict(sd)
    second = {k: v.clone() for k, v in graphed(src, drv).items()}
    fresh = engine.Reconstructor(kpd, gen, use_graph=False)(src, drv)
    torch.cuda.synchronize()
    assert float((second["video_prediction"] - first["video_prediction"]).abs().max()) > 1e-3
    assert torch.equal(second["video_prediction"], fresh["video_prediction"])
    assert torch.equal(second["kp_driving_mean"], fresh["kp_driving_mean"])


def test_no_grad_pack_cache_sees_replaced_storage(be):
    """`p.data = other` keeps the Parameter object and its version counter: the inference-side packed-weight cache is
    also keyed by the storage address (round-1 advisory), and holds the parameter only weakly."""
    import gc
    import torch.nn.functional as F
    from mnk import ops
    torch.manual_seed(7)
    w = torch.nn.Parameter(be.t(torch.randn(5, 4, 1, 3, 3) * 0.3))
    x = torch.rand(1, 4, 1, 6, 6)
    xa = ops.to_act(be.t(x))

    def run():
        with torch.no_grad():
            y, _ = ops.conv3x3(xa, 4, w)
        return ops.from_act(y, 5, 1).cpu()

    def ref():
        return F.conv2d(x[:, :, 0].double(), w.detach().cpu()[:, :, 0].double(), padding=1).float().unsqueeze(2)

    assert float((run() - ref()).abs().max()) < 1e-5
    w.data = be.t(torch.randn(5, 4, 1, 3, 3) * 0.3)
    assert float((run() - ref()).abs().max()) < 1e-5, "stale packed weights after p.data = ..."
    gc.collect()                    # garbage of earlier tests in this process (whole models in reference cycles) goes first
    n0 = len(ops._PACK_CACHE)
    del w
    gc.collect()
    assert len(ops._PACK_CACHE) == n0 - 1


def test_batched_transfer_equals_the_frame_loop_and_normalize_kp_its_formulas(be):
    """mnk.engine.Transfer folds transfer.py:65-79's per-frame loops into one detector and one generator call; here against
    the same modules driven frame by frame, with the key-point normalisation (transfer.py:31-62) restated in torch + scipy
    (convex-hull area) + numpy (symmetric eigen-repair).  (Against the reference's own transfer_one: test_dropin_reference.)"""
    import numpy as np
    from scipy.spatial import ConvexHull
    from mnk import engine
    from modules.util import matrix_inverse
    gold = load("tiny")
    gen, disc, kpd = build(gold["cfg"])
    gen.load_state_dict(gold["state"]["generator"]), kpd.load_state_dict(gold["state"]["kp_detector"])
    gen.to(be.device).eval(), kpd.to(be.device).eval()
    src, _ = cases.smooth_pair(2, gold["size"], gold["size"], seed=11)
    driving = torch.cat([cases.smooth_pair(2, gold["size"], gold["size"], seed=20 + i)[1] for i in range(3)], dim=2)
    src, driving = be.t(src), be.t(driving)
    params = dict(movement_mult=True, move_location=True, adapt_variance=True, clip_mean=True)
    got = engine.Transfer(kpd, gen, params)(src, driving)
    be.sync()
    with torch.no_grad():
        kp_d = {k: torch.cat([kpd(driving[:, :, i:i + 1])[k] for i in range(3)], dim=1) for k in ("mean", "var")}
        kp_s = kpd(src)
        mv, vv, ma, va = (t.cpu().double() for t in (kp_d["mean"], kp_d["var"], kp_s["mean"], kp_s["var"]))
        mult = np.sqrt(ConvexHull(ma[0, 0].numpy()).volume) / np.sqrt(ConvexHull(mv[0, 0].numpy()).volume)
        mean = ((mv - mv[:, 0:1]) * mult + ma).clamp(-1, 1)
        var = torch.matmul(torch.matmul(vv, matrix_inverse(vv[:, 0:1])), va)
        sym = (var + var.transpose(-1, -2)) / 2
        ev, eu = np.linalg.eigh(sym.numpy())
        ev[ev <= 0] = 1e-6
        var = torch.from_numpy(np.einsum("...ij,...j,...kj->...ik", eu, ev, eu))
        assert float((got["kp_norm"]["mean"].cpu().double() - mean).abs().max()) < 2e-6
        assert float((got["kp_norm"]["var"].cpu().double() - var).abs().max()) < 2e-6
        frames = []
        for i in range(3):
            kp_i = {k: v[:, i:i + 1] for k, v in got["kp_norm"].items()}
            frames.append(gen(src, kp_driving=kp_i, kp_source=kp_s)["video_prediction"])
        loop = torch.cat(frames, dim=2)
    be.sync()
    assert got["video_prediction"].shape == loop.shape == (2, 3, 3, gold["size"], gold["size"])
    assert float((got["video_prediction"] - loop).abs().max()) < 2e-6


@pytest.mark.gpu
def test_eval_wrappers_replay_a_frozen_weight_graph_per_frame(monkeypatch):
    """reconstruction.py:45-62's loop: kp_detector and generator behind DataParallelWithCallback in evaluation mode under no_grad,
    one frame at a time.  The wrappers replay a hipGraph captured per input signature with frozen weights (mnk.dropin.EvalRunner):
    same bits as eager launches; `kp_source` survives the later calls (outputs are fresh tensors); a load_state_dict between two
    videos is seen (re-capture)."""
    from conftest import Backend
    from sync_batchnorm import DataParallelWithCallback
    from mnk import dropin
    be = Backend("hip")
    gold = load("bair")
    gen, kpd = _models(gold, be)
    generator, kp_detector = DataParallelWithCallback(gen), DataParallelWithCallback(kpd)
    generator.eval(), kp_detector.eval()
    g = torch.Generator().manual_seed(5)
    video = torch.rand(1, 3, 6, 64, 64, generator=g)

    def loop():
        outs = []
        with torch.no_grad():
            kp_source = kp_detector(video[:, :, :1])
            keep = {k: v.clone() for k, v in kp_source.items()}
            for i in range(video.shape[2]):
                kp_driving = kp_detector(video[:, :, i:i + 1])
                out = generator(source_image=video[:, :, :1], kp_driving=kp_driving, kp_source=kp_source)
                outs.append(out["video_prediction"].clone())
            torch.cuda.synchronize()
            assert all(torch.equal(kp_source[k], keep[k]) for k in keep)          # not overwritten by the later detector calls
        return torch.cat(outs, dim=2)

    monkeypatch.setenv("MNK_EVAL_GRAPH", "0")
    want = loop()
    monkeypatch.delenv("MNK_EVAL_GRAPH")
    got = loop()
    rk, rg = dropin.eval_runner_for_wrapper(kp_detector), dropin.eval_runner_for_wrapper(generator)
    assert rk is not None and rg is not None and rk.stats["captures"] == 1 and rg.stats["captures"] == 1
    assert rk.stats["replays"] == 7 and rg.stats["replays"] == 6
    assert torch.equal(got, want)
    # other weights: the replay must follow them
    sd = {k: v.detach().cpu().clone() for k, v in gen.state_dict().items()}
    cases.perturb_state_dict(sd, 99)
    gen.load_state_dict(sd)
    monkeypatch.setenv("MNK_EVAL_GRAPH", "0")
    want2 = loop()
    monkeypatch.delenv("MNK_EVAL_GRAPH")
    got2 = loop()
    assert rg.stats["captures"] == 2 and torch.equal(got2, want2) and not torch.equal(got2, got)


@pytest.mark.parametrize("config,batch", [("moving-gif", 1), ("moving-gif", 3), ("taichi", 1)])
def test_eval_norm_layers_sum_the_split_partials_themselves_bit_for_bit(be, monkeypatch, config, batch):
    if be.kind == "emu" and (config, batch) != ("moving-gif", 1):
        pytest.skip("the other shapes run on the device only (CPU suite budget)")
    _eval_split_case(be, monkeypatch, config, batch, 64, with_generator=be.kind == "hip")


def kpd_mean_of(a, b, src, kpd):
    with torch.no_grad():
        return kpd(src)["mean"]


def _eval_split_case(be, monkeypatch, config, batch, size, with_generator):
    """(the CPU emulator runs the key-point detector only: down blocks with the pool, sub-pixel up blocks -- CPU suite budget)"""
    """reconstruction.py:45-62 runs the networks frame by frame at batch 1: every convolution is split along K there, and in
    evaluation mode under no_grad the norm layer behind it takes the partials (mnk_bn_eval_split_fwd: reduction + bias + affine +
    ReLU + pool in one launch, y never written).  Same bits as the two-launch form (split reduction, then mnk_bn_act_fwd), and
    the fused form is what ran."""
    from mnk import configs, knobs, ops
    cfg = configs.get(config)
    gen, disc, kpd = build(cfg)
    for i, m in enumerate((gen, disc, kpd)):
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 11 + i)
        m.load_state_dict(sd)
    gen.to(be.device).eval(), kpd.to(be.device).eval()
    src, drv = cases.synthetic_pair(batch, size, size)
    src, drv = be.t(src), be.t(drv)
    calls = {}
    real = ops._call

    def counting(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return real(name, *a)

    monkeypatch.setattr(ops, "_call", counting)

    def run(fused):
        monkeypatch.setitem(knobs.FORMS, "EVAL_SPLIT_FUSED", fused)
        calls.clear()
        with torch.no_grad():
            kp_s, kp_d = kpd(src), kpd(drv)
            out = gen(source_image=src, kp_driving=kp_d, kp_source=kp_s) if with_generator else {}
        be.sync()
        return {"mean": kp_d["mean"], "var": kp_d["var"], **out}, dict(calls)

    a, ca = run(True)
    assert ops.handover_state() == {}
    b, cb = run(False)
    assert ca.get("mnk_bn_eval_split_fwd", 0) > 0 and cb.get("mnk_bn_eval_split_fwd", 0) == 0
    assert ca.get("mnk_bn_act_fwd", 0) + ca["mnk_bn_eval_split_fwd"] == cb["mnk_bn_act_fwd"]
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # with gradients enabled nothing is deferred (a backward pass needs y)
    monkeypatch.setitem(knobs.FORMS, "EVAL_SPLIT_FUSED", True)
    calls.clear()
    out = kpd(src)
    assert calls.get("mnk_bn_eval_split_fwd", 0) == 0 and calls.get("mnk_bn_act_fwd", 0) > 0
    assert torch.equal(out["mean"].detach(), kpd_mean_of(a, b, src, kpd))


def test_an_exception_inside_a_convolution_call_leaves_no_hand_over_behind(be, monkeypatch):
    """conv3x3(eval_bn=True) raises one-shot flags for the launch it is about to make; if the call dies first (here: the weight
    pack fails) the flags must be gone -- the next launch through _conv_launch may be a data-gradient launch of an unrelated
    backward pass, which must write its output."""
    from mnk import ops
    x = be.t(torch.randn(1, 8, 8, 8))
    w = be.t(torch.randn(8, 8, 1, 3, 3) * 0.1)

    def boom(*a, **k):
        raise RuntimeError("pack failed")

    monkeypatch.setattr(ops, "_packed_fwd_weight", boom)
    with torch.no_grad():
        with pytest.raises(RuntimeError, match="pack failed"):
            ops.conv3x3(x, 8, w, None, eval_bn=True)
    assert ops.handover_state() == {}
