"""The library's stream executor (csrc/replay.hip, mnk_replay_*) against hipGraphLaunch: the SAME captured iteration issued
as plain stream launches -- on one stream, and with the branches of its dependency graph (background weight-gradient GEMMs)
on streams of their own -- must give the results of the hipGraph replay it replaces (mnk.engine.TrainStep, the captured
train.py:110-136 iteration)."""
import pytest
import torch

from oracle import cases

pytestmark = pytest.mark.gpu


def _models(name):
    from mnk import configs
    from test_modules import build
    gen, disc, kpd = build(configs.get(name) if isinstance(name, str) else name)
    for i, m in enumerate((gen, disc, kpd)):
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 7 + i)
        m.load_state_dict(sd)
    dev = torch.device("cuda:0")
    return gen.to(dev), disc.to(dev), kpd.to(dev)


def _run(monkeypatch, cfg, name, x, streams, n=4, bg=None):
    from mnk import engine
    monkeypatch.setenv("MNK_REPLAY_STREAMS", str(streams))
    if bg is not None:
        monkeypatch.setenv("MNK_WGRAD_BG", str(bg))
    gen, disc, kpd = _models(name)
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], use_graph=True)
    out = []
    for _ in range(n):
        g_l, d_l, gen_out = step.step(x)
        torch.cuda.synchronize()
        out.append([float(v) for v in g_l] + [float(v) for v in d_l])
    sample = torch.cat([p.detach().reshape(-1)[:64] for p in list(gen.parameters())[:40]]).cpu()
    pred = gen_out["video_prediction"].detach().cpu().clone()
    return out, sample, pred, step.replay_info


@pytest.mark.parametrize("size", ["tiny", "moving-gif"])
def test_stream_executor_replays_the_captured_iteration(monkeypatch, size):
    from mnk import configs
    if size == "tiny":
        cfg, name = cases.TINY, cases.TINY
        src, drv = cases.smooth_pair(4, 32, 32)
    else:
        cfg = configs.get("moving-gif")
        name = "moving-gif"
        src, drv = cases.synthetic_pair(32, 64, 64)
    x = {"source": src.cuda(), "video": drv.cuda()}
    ref, ref_p, ref_pred, info0 = _run(monkeypatch, cfg, name, x, 0)
    ref2, _, _, _ = _run(monkeypatch, cfg, name, x, 0)          # two hipGraph runs: the yard-stick (fp32 atomics of the warp backward)
    assert info0 is None

    def dev(p, q):
        return max(abs(u - v) / max(1.0, abs(v)) for u, v in zip(p, q))

    for streams, bg in ((1, None), (2, None), (3, 2)):
        got, got_p, got_pred, info = _run(monkeypatch, cfg, name, x, streams, bg=bg)
        assert info is not None and "refused" not in info, info
        assert info["memcpys"] == 0 or info["nodes"] > 0
        assert info["streams"] <= streams
        if streams == 1:
            assert info["side_stream_nodes"] == 0 and info["cross_stream_waits"] == 0
        elif size != "tiny":
            assert info["side_stream_nodes"] > 0, "the background weight-gradient launches should form a branch: %s" % info
        for k in range(len(ref)):
            assert all(v == v and abs(v) < 1e6 for v in got[k]), ("non-finite loss", streams, k, got[k])
            noise = dev(ref[k], ref2[k])
            assert dev(got[k], ref[k]) <= 4 * noise + 5e-3, (streams, k, dev(got[k], ref[k]), noise)
        assert torch.isfinite(got_p).all()
        assert float((got_p - ref_p).abs().max()) <= 5e-3 * max(1.0, float(ref_p.abs().max()))
        assert float((got_pred - ref_pred).abs().max()) < 0.05
