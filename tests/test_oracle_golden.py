"""Pins oracle/restate.py (the CPU restatement) to the committed outputs of the REAL reference
(tests/golden/*.pt, written by oracle/make_golden.py in the authoring container where /root/reference exists).
Runs anywhere -- the GPU box has no reference tree."""
import os

import pytest
import torch

from oracle import restate, cases, ref_shim

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)


def test_functions_match_reference():
    g = load("functions")
    assert torch.equal(restate.make_coordinate_grid(5, 7), g["grid_5x7"])
    assert torch.equal(restate.smallest_singular(g["mat"]), g["smallest_singular"])
    assert torch.allclose(restate.matrix_inverse(g["mat"]), g["matrix_inverse"], atol=1e-6)
    logits = g["g2k_logits"]
    heat = torch.softmax(logits.view(2, 3, 2, -1) / 0.1, dim=3).view_as(logits)
    for tag, kw in (("matrix", dict(kp_variance="matrix")), ("clip", dict(kp_variance="matrix", clip_variance=0.001)),
                    ("single", dict(kp_variance="single")), ("const", dict(kp_variance=0.01))):
        out = restate.gaussian2kp(heat, **kw)
        for k, v in out.items():
            assert torch.allclose(v, g["g2k_%s_%s" % (tag, k)], atol=1e-6), (tag, k)
    for tag, kv in (("matrix", "matrix"), ("const", 0.01)):
        assert torch.allclose(restate.kp2gaussian(g["k2g_kp"], (9, 6), kv), g["k2g_" + tag], atol=1e-6)
    for tag, kw in g["emb_variants"].items():
        p = dict(kw, num_kp=4, kp_variance="matrix", num_channels=3)
        out = restate.movement_embedding(p, g["emb_src"], g["emb_kpd"], g["emb_kps"])
        assert torch.allclose(out, g["emb_" + tag], atol=2e-6), tag
    for tag, mode in (("same", "nearest"), ("down", "nearest"), ("up", "nearest"), ("one", "nearest"),
                      ("tri_down", "trilinear"), ("tri_up", "trilinear")):
        out = restate.deform_input(g["deform_%s_in" % tag], g["deform_field"], mode)
        assert torch.allclose(out, g["deform_%s_out" % tag], atol=2e-6), tag


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
def test_modules_match_reference_in_fp64(name):
    """fp64 restatement == fp64 reference to ~1e-9: implementation noise is gone, only the algorithm is compared."""
    gold = load(name)
    cfg = gold["cfg"]
    mp = cfg["model_params"]
    common = mp["common_params"]
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
    sds = restate.to_dtype(gold["state"], torch.float64)
    for mode in ("train", "eval"):
        kp = restate.kp_detector_forward(sds["kp_detector"], dict(mp["kp_detector_params"], **common),
                                         torch.cat([src, drv], 2).double(), training=(mode == "train"))
        res = restate.generator_forward(sds["generator"], mp["generator_params"], common, src.double(),
                                        {k: v[:, 1:] for k, v in kp.items()}, {k: v[:, :1] for k, v in kp.items()},
                                        training=(mode == "train"))
        ref = gold[mode + "64"]
        assert float((kp["mean"] - ref["kp_mean"]).abs().max()) < 1e-9
        assert float((kp["var"] - ref["kp_var"]).abs().max()) < 1e-9
        assert float((res["video_prediction"] - ref["video_prediction"]).abs().max()) < 1e-9
        assert float((res["video_deformed"] - ref["video_deformed"]).abs().max()) < 1e-9


def test_step_losses_match_reference():
    gold = load("step_tiny")
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
    losses, _, _, _, _ = restate.generator_full_forward(gold["state"], gold["cfg"], src, drv)
    for a, b in zip(losses, gold["history"][0]["generator"]):
        assert abs(float(a.mean()) - b) < 2e-4 * max(1.0, abs(b))


def test_flop_model_matches_survey():
    """forward conv GFLOP per training pair (measured with hooks on the reference)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), "..", "monkey-net_amd"))
    from mnk import configs
    for name, size, total in (("taichi", 64, 9.470), ("moving-gif", 64, 5.116), ("moving-gif", 128, 20.465),
                              ("shapes", 64, 1.892), ("bair", 64, 8.867), ("vox", 256, 64.404)):
        f = restate.conv_flops_hot_path(configs.get(name), size, size)
        assert abs(f["total"] / 1e9 - total) < 2e-3, (name, f["total"])
        from mnk import workload          # the product-side accounting bench.py quotes is the same count
        assert workload.conv_flops_hot_path(configs.get(name), size, size)["layers"] == f["layers"]


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree only exists in the authoring container")
def test_live_reference_agrees_with_golden():
    """Where /root/reference is present, re-run the real reference on one case and compare with the fixture."""
    import subprocess
    import sys
    code = ("import sys, torch; sys.path.insert(0, %r); from oracle import ref_shim, cases; ref = ref_shim.load();"
            "g = torch.load(%r, weights_only=False); heat = torch.softmax(g['g2k_logits'].view(2,3,2,-1)/0.1, 3)"
            ".view_as(g['g2k_logits']); out = ref.gaussian2kp(heat, 'matrix', 0.001);"
            "assert torch.equal(out['var'], g['g2k_clip_var']); print('ok')") % (
        os.path.dirname(os.path.dirname(GOLD)), os.path.join(GOLD, "functions.pt"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
