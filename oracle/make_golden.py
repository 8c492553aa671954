"""Generate tests/golden/*.pt from the REAL reference (imported unmodified from /root/reference through
oracle/ref_shim.py) and, in the same run, check oracle/restate.py against it.  TEST INFRASTRUCTURE ONLY.

Run in the authoring container only (the GPU box has no /root/reference):

    python -m oracle.make_golden            # writes tests/golden/, prints the restatement errors

The committed fixtures are what pins the oracle (the reference ships no tests/goldens of its own,
SURVEY.md section 4).
"""
import copy
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim, restate, cases  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REPORT = []


def maxdiff(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max())


def check(name, ref, mine, tol):
    d = maxdiff(ref, mine)
    REPORT.append((name, d, tol))
    assert d <= tol, "%s: restatement differs from reference by %g (> %g)" % (name, d, tol)


def save(name, obj):
    os.makedirs(GOLD, exist_ok=True)
    torch.save(obj, os.path.join(GOLD, name + ".pt"))


def load_cfg(name):
    with open(os.path.join(ref_shim.REFERENCE_ROOT, "config", name + ".yaml")) as f:
        return yaml.safe_load(f)


# ----------------------------------------------------------------------------------------------
def functions(ref):
    out = {}
    g = torch.Generator().manual_seed(11)
    # make_coordinate_grid (modules/util.py:26-42)
    grid = ref.util.make_coordinate_grid((5, 7), torch.FloatTensor().type())
    check("grid", grid, restate.make_coordinate_grid(5, 7), 0)
    out["grid_5x7"] = grid
    # smallest_singular / matrix_inverse
    m = torch.randn(6, 2, 2, generator=g)
    m = torch.matmul(m, m.transpose(-1, -2)) + 0.05 * torch.eye(2)
    ss = ref.util.smallest_singular(m)
    check("smallest_singular", ss, restate.smallest_singular(m), 0)
    inv = ref.util.matrix_inverse(m)
    check("matrix_inverse", inv, restate.matrix_inverse(m), 1e-6)
    out.update(mat=m, smallest_singular=ss, matrix_inverse=inv)
    # gaussian2kp (modules/keypoint_detector.py:43-78)
    logits = torch.randn(2, 3, 2, 8, 12, generator=g) * 3
    heat = torch.softmax(logits.view(2, 3, 2, -1) / 0.1, dim=3).view_as(logits)
    out["g2k_logits"] = logits
    for tag, kw in (("matrix", dict(kp_variance="matrix", clip_variance=None)),
                    ("clip", dict(kp_variance="matrix", clip_variance=0.001)),
                    ("single", dict(kp_variance="single")),
                    ("const", dict(kp_variance=0.01))):
        r = ref.gaussian2kp(heat, **kw)
        mne = restate.gaussian2kp(heat, **kw)
        for k in r:
            check("gaussian2kp.%s.%s" % (tag, k), r[k], mne[k], 1e-6)
            out["g2k_%s_%s" % (tag, k)] = r[k]
    # kp2gaussian (modules/keypoint_detector.py:7-40)
    kp = cases.random_kp(2, 2, 3, seed=5)
    out["k2g_kp"] = kp
    for tag, kv in (("matrix", "matrix"), ("const", 0.01)):
        r = ref.kp2gaussian(kp, (9, 6), kv)
        check("kp2gaussian." + tag, r, restate.kp2gaussian(kp, (9, 6), kv), 1e-6)
        out["k2g_" + tag] = r
    # MovementEmbeddingModule (modules/movement_embedding.py:42-92)
    src = torch.rand(2, 3, 1, 16, 16, generator=g)
    kpd, kps = cases.random_kp(2, 1, 4, seed=6), cases.random_kp(2, 1, 4, seed=7)
    out.update(emb_src=src, emb_kpd=kpd, emb_kps=kps)
    variants = {
        "mask": dict(use_heatmap=True, use_deformed_source_image=True, heatmap_type="difference", norm_const=100,
                     add_bg_feature_map=True),
        "mask_diff": dict(use_heatmap=True, use_deformed_source_image=True, use_difference=True,
                          heatmap_type="difference", norm_const=100, add_bg_feature_map=True),
        "kpemb": dict(use_heatmap=True, norm_const=100, heatmap_type="difference"),
        "sum": dict(use_heatmap=True, use_deformed_source_image=True, heatmap_type="gaussian", norm_const="sum",
                    add_bg_feature_map=True),
        "half": dict(use_heatmap=True, norm_const=10, heatmap_type="gaussian", scale_factor=0.5),
        "diffonly": dict(use_heatmap=False, use_difference=True, add_bg_feature_map=True),
    }
    out["emb_variants"] = variants
    for tag, kw in variants.items():
        mod = ref.MovementEmbeddingModule(num_kp=4, kp_variance="matrix", num_channels=3, **kw)
        r = mod(src, kpd, kps)
        mne = restate.movement_embedding(dict(kw, num_kp=4, kp_variance="matrix", num_channels=3), src, kpd, kps)
        check("movement_embedding." + tag, r, mne, 2e-6)
        out["emb_" + tag] = r
    # deform_input (modules/generator.py:51-58)
    cfg = load_cfg("shapes")
    gen = ref.MotionTransferGenerator(**cfg["model_params"]["generator_params"],
                                      **cfg["model_params"]["common_params"])
    field = torch.cat([restate.make_coordinate_grid(16, 16).view(1, 1, 16, 16, 2).repeat(2, 1, 1, 1, 1) +
                       0.3 * torch.randn(2, 1, 16, 16, 2, generator=g), torch.zeros(2, 1, 16, 16, 1)], -1)
    out["deform_field"] = field
    for tag, shp, mode in (("same", (2, 5, 1, 16, 16), "nearest"), ("down", (2, 6, 1, 4, 4), "nearest"),
                           ("up", (2, 3, 1, 32, 32), "nearest"), ("one", (2, 7, 1, 1, 1), "nearest"),
                           ("tri_down", (2, 6, 1, 8, 8), "trilinear"), ("tri_up", (2, 3, 1, 32, 32), "trilinear")):
        inp = torch.rand(*shp, generator=g)
        gen.interpolation_mode = mode
        r = gen.deform_input(inp, field)
        check("deform_input." + tag, r, restate.deform_input(inp, field, mode), 2e-6)
        out["deform_%s_in" % tag] = inp
        out["deform_%s_out" % tag] = r
    save("functions", out)


# ----------------------------------------------------------------------------------------------
def build_reference(ref, cfg, seed=0, perturb_seed=7):
    """run.py:50-62 construction order: generator, discriminator, kp_detector."""
    mp = cfg["model_params"]
    torch.manual_seed(seed)
    gen = ref.MotionTransferGenerator(**mp["generator_params"], **mp["common_params"])
    disc = ref.Discriminator(**mp["discriminator_params"], **mp["common_params"])
    kpd = ref.KPDetector(**mp["kp_detector_params"], **mp["common_params"])
    init_sums = {n: float(sum(v.double().abs().sum() for v in m.state_dict().values()))
                 for n, m in (("generator", gen), ("discriminator", disc), ("kp_detector", kpd))}
    if perturb_seed is not None:
        for i, m in enumerate((gen, disc, kpd)):
            sd = m.state_dict()
            cases.perturb_state_dict(sd, perturb_seed + i)
            m.load_state_dict(sd)
    return gen, disc, kpd, init_sums


def grads_of(module):
    return {k: p.grad.clone() for k, p in module.named_parameters() if p.grad is not None}


def _run_reference(gen, kpd, src, drv, r1, r2, train, backward):
    gen.train(train), kpd.train(train)
    gen.zero_grad(), kpd.zero_grad()
    kp_joined = kpd(torch.cat([src, drv], dim=2))
    res = gen(src, kp_driving={k: v[:, 1:] for k, v in kp_joined.items()},
              kp_source={k: v[:, :1] for k, v in kp_joined.items()})
    out = {"kp_mean": kp_joined["mean"].detach(), "kp_var": kp_joined["var"].detach(),
           "video_prediction": res["video_prediction"].detach(), "video_deformed": res["video_deformed"].detach()}
    grads = None
    if backward:
        loss = (res["video_prediction"] * r1).sum() + (res["video_deformed"] * r2).sum()
        loss.backward()
        out["loss"] = loss.detach()
        grads = {"generator": grads_of(gen), "kp_detector": grads_of(kpd)}
    return out, grads


def _run_restate(sds, cfg, src, drv, r1, r2, train, backward):
    mp = cfg["model_params"]
    common = mp["common_params"]
    sds = {k: {n: t.clone().requires_grad_(backward and t.is_floating_point()) for n, t in v.items()}
           for k, v in sds.items()}
    kp = restate.kp_detector_forward(sds["kp_detector"], dict(mp["kp_detector_params"], **common),
                                     torch.cat([src, drv], dim=2), training=train)
    res = restate.generator_forward(sds["generator"], mp["generator_params"], common, src,
                                    {k: v[:, 1:] for k, v in kp.items()}, {k: v[:, :1] for k, v in kp.items()},
                                    training=train)
    out = {"kp_mean": kp["mean"].detach(), "kp_var": kp["var"].detach(),
           "video_prediction": res["video_prediction"].detach(), "video_deformed": res["video_deformed"].detach()}
    grads = None
    if backward:
        loss = (res["video_prediction"] * r1).sum() + (res["video_deformed"] * r2).sum()
        loss.backward()
        out["loss"] = loss.detach()
        grads = {m: {k: t.grad for k, t in sds[m].items() if t.grad is not None} for m in ("generator", "kp_detector")}
    return out, grads


def relerr(a, b):
    # conv biases in front of a BatchNorm have an analytically zero gradient (pure rounding noise in the
    # reference too): the absolute floor keeps them from dominating a relative error
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-6))


def module_case(ref, name, cfg, batch, size, store_weights, smooth=True, grad_keys=None, compact=False):
    """Goldens = the reference run in fp32 (what a user of the reference gets) AND in fp64 (the arbiter).
    The restatement is pinned against the reference in fp64, where implementation noise vanishes."""
    gen, disc, kpd, init_sums = build_reference(ref, cfg)
    src, drv = (cases.smooth_pair if smooth else cases.synthetic_pair)(batch, size, size)
    out = {"cfg": cfg, "batch": batch, "size": size, "smooth": smooth, "init_sums": init_sums}
    sds0 = {"generator": copy.deepcopy(gen.state_dict()), "kp_detector": copy.deepcopy(kpd.state_dict()),
            "discriminator": copy.deepcopy(disc.state_dict())}
    if store_weights:
        out["state"] = sds0
    g = torch.Generator().manual_seed(99)
    r1 = torch.randn(batch, 3, 1, size, size, generator=g)
    r2 = torch.randn(batch, 3, 1, size, size, generator=g)
    out["loss_weights"] = (r1, r2)

    def keep(gr):
        if grad_keys is None:
            return gr
        return {m: {k: v for k, v in d.items() if any(s in k for s in grad_keys)} for m, d in gr.items()}

    for mode in ("train", "eval"):
        train = mode == "train"
        gen.load_state_dict(sds0["generator"]), kpd.load_state_dict(sds0["kp_detector"])
        o32, g32 = _run_reference(gen, kpd, src, drv, r1, r2, train, backward=train)
        if train and store_weights:
            out["running_after_train"] = {"generator": {k: v.clone() for k, v in gen.state_dict().items() if "running" in k},
                                          "kp_detector": {k: v.clone() for k, v in kpd.state_dict().items() if "running" in k}}
        gen.load_state_dict(sds0["generator"]), kpd.load_state_dict(sds0["kp_detector"])
        gen.double(), kpd.double()
        o64, g64 = _run_reference(gen, kpd, src.double(), drv.double(), r1.double(), r2.double(), train, backward=train)
        gen.float(), kpd.float()
        sds64 = restate.to_dtype(sds0, torch.float64)
        m64, mg64 = _run_restate(sds64, cfg, src.double(), drv.double(), r1.double(), r2.double(), train, backward=train)
        m32, mg32 = _run_restate(sds0, cfg, src, drv, r1, r2, train, backward=train)
        for k in ("kp_mean", "kp_var", "video_prediction", "video_deformed"):
            check("%s.%s.%s restate64-vs-ref64" % (name, mode, k), o64[k], m64[k], 1e-7)
            REPORT.append(("%s.%s.%s ref32-vs-ref64 (info)" % (name, mode, k), maxdiff(o32[k], o64[k]), float("inf")))
            REPORT.append(("%s.%s.%s restate32-vs-ref64 (info)" % (name, mode, k), maxdiff(m32[k], o64[k]), float("inf")))
        if compact:   # large frames: keep the fp64 run (rounded to fp32) + the fp32-vs-fp64 spreads only
            out[mode + "64"] = {k: v.float() for k, v in o64.items()}
            out[mode + "_spread"] = {k: maxdiff(o32[k], o64[k]) for k in ("kp_mean", "kp_var", "video_prediction",
                                                                            "video_deformed")}
        else:
            out[mode] = o32
            out[mode + "64"] = o64
        if train:
            worst = 0.0
            for m in ("generator", "kp_detector"):
                for k, v in g64[m].items():
                    if cases.is_noise_bias(k):
                        continue
                    d = relerr(mg64[m][k], v)
                    worst = max(worst, d)
                    assert d < 1e-6, (m, k, d)
            REPORT.append(("%s.train.grads restate64-vs-ref64 (worst rel)" % name, worst, 1e-6))
            spread32 = {m: {k: relerr(g32[m][k], g64[m][k]) for k in g64[m] if not cases.is_noise_bias(k)} for m in g64}
            spread_m32 = {m: {k: relerr(mg32[m][k], g64[m][k]) for k in g64[m] if not cases.is_noise_bias(k)} for m in g64}
            REPORT.append(("%s.train.grads ref32-vs-ref64 (median rel, info)" % name,
                           float(torch.tensor([v for d in spread32.values() for v in d.values()]).median()), float("inf")))
            REPORT.append(("%s.train.grads restate32-vs-ref64 (median rel, info)" % name,
                           float(torch.tensor([v for d in spread_m32.values() for v in d.values()]).median()), float("inf")))
            out["grad32"] = keep(g32)
            # fp64 gradients are stored rounded to fp32 (6e-8 relative): enough for an arbiter, half the bytes
            out["grad64"] = {m: {k: v.float() for k, v in d.items()} for m, d in keep(g64).items()}
            out["grad64_norms"] = {m: {k: float(v.norm()) for k, v in d.items()} for m, d in g64.items()}
            out["grad_ref32_vs_ref64_rel"] = spread32
    save(name, out)


def _run_steps(ref, cfg, batch, size, steps, dtype):
    gen, disc, kpd, _ = build_reference(ref, cfg)
    tp = cfg["train_params"]
    state0 = {"generator": copy.deepcopy(gen.state_dict()), "discriminator": copy.deepcopy(disc.state_dict()),
              "kp_detector": copy.deepcopy(kpd.state_dict())}
    for m in (gen, disc, kpd):
        m.to(dtype)
    gfull = ref.GeneratorFullModel(kpd, gen, disc, tp)
    dfull = ref.DiscriminatorFullModel(kpd, gen, disc, tp)
    og = torch.optim.Adam(gen.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    od = torch.optim.Adam(disc.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    ok = torch.optim.Adam(kpd.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    src, drv = cases.smooth_pair(batch, size, size)
    x = {"source": src.to(dtype), "video": drv.to(dtype)}
    hist = []
    for it in range(steps):
        outs = gfull(x)
        lv = [v.mean() for v in outs[:-2]]
        generated, kp_joined = outs[-2], outs[-1]
        sum(lv).backward(retain_graph=not tp["detach_kp_discriminator"])
        og.step(), og.zero_grad(), od.zero_grad()
        if tp["detach_kp_discriminator"]:
            ok.step(), ok.zero_grad()
        gl = [float(v.detach()) for v in lv]
        dl = [v.mean() for v in dfull(x, kp_joined, generated)]
        sum(dl).backward()
        od.step(), od.zero_grad()
        if not tp["detach_kp_discriminator"]:
            ok.step(), ok.zero_grad()
        hist.append({"generator": gl, "discriminator": [float(v.detach()) for v in dl],
                     "prediction_mean": float(generated["video_prediction"].mean())})
    final = {k: float(sum(v.double().abs().sum() for v in m.state_dict().values()))
             for k, m in (("generator", gen), ("discriminator", disc), ("kp_detector", kpd))}
    return state0, hist, final


def step_case(ref, name, cfg, batch, size, steps=3):
    """train.py:110-136, `steps` iterations with 3x Adam(lr, betas=(0.5,0.999)), in fp32 and in fp64.
    Adam's first updates are sign-like (m/sqrt(v) = +-1), so rounding noise on near-zero gradient elements moves
    parameters by +-lr: the fp32-vs-fp64 spread of the reference itself is what bounds a meaningful tolerance."""
    state0, hist, final = _run_steps(ref, cfg, batch, size, steps, torch.float32)
    _, hist64, final64 = _run_steps(ref, cfg, batch, size, steps, torch.float64)
    src, drv = cases.smooth_pair(batch, size, size)
    lm, _, _, _, _ = restate.generator_full_forward(copy.deepcopy(state0), cfg, src, drv)
    for i, (a, b) in enumerate(zip(hist[0]["generator"], lm)):
        check("%s.step0.gen_loss%d restate32-vs-ref32" % (name, i), torch.tensor(a), b.mean().detach(), 2e-4)
    for it in range(steps):
        d = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(hist[it]["generator"] + hist[it]["discriminator"],
                                                              hist64[it]["generator"] + hist64[it]["discriminator"]))
        REPORT.append(("%s.step%d losses ref32-vs-ref64 (max rel, info)" % (name, it), d, float("inf")))
    save(name, {"cfg": cfg, "batch": batch, "size": size, "state": state0, "history": hist, "history64": hist64,
                "final_checksum": final, "final_checksum64": final64})


def main():
    assert ref_shim.available(), "run this in the authoring container (needs /root/reference)"
    torch.set_num_threads(8)
    ref = ref_shim.load()
    functions(ref)
    module_case(ref, "tiny", cases.TINY, batch=2, size=32, store_weights=True)
    module_case(ref, "tiny2", cases.TINY2, batch=3, size=16, store_weights=True)
    keys = ("down_blocks.0.conv.weight", "decoder.conv.weight", "conv-last", "r0.conv1.weight",
            "up_blocks.4.conv.weight", "group_blocks.0.conv.weight")
    module_case(ref, "shapes", load_cfg("shapes"), batch=2, size=64, store_weights=False, grad_keys=keys)
    module_case(ref, "taichi", load_cfg("taichi"), batch=2, size=64, store_weights=False, grad_keys=keys[:1])
    module_case(ref, "moving-gif", load_cfg("moving-gif"), batch=2, size=64, store_weights=False, grad_keys=keys[:1])
    module_case(ref, "bair", load_cfg("bair"), batch=2, size=64, store_weights=False, grad_keys=keys[:1], compact=True)
    module_case(ref, "vox", load_cfg("vox"), batch=2, size=128, store_weights=False, grad_keys=keys[:1], compact=True)
    step_case(ref, "step_tiny", cases.TINY, batch=2, size=32)
    width = max(len(r[0]) for r in REPORT)
    with open(os.path.join(GOLD, "RESTATEMENT_REPORT.txt"), "w") as f:
        f.write("# max |reference - oracle/restate.py| (or relative grad error) per check; made by oracle/make_golden.py\n")
        for n, d, t in REPORT:
            f.write("%-*s %.3e (tol %.1e)\n" % (width, n, d, t))
    print("wrote goldens; %d checks; worst ratio %.3f" % (len(REPORT), max(d / t if (t and t != float("inf")) else 0 for _, d, t in REPORT)))


if __name__ == "__main__":
    main()
