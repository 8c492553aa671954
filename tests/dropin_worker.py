"""Worker of tests/test_dropin_reference.py (its own process: the package names `modules` / `sync_batchnorm` must
resolve freshly).  sys.path = [monkey-net_amd, /root/reference, ...]: the REFERENCE's train.py / reconstruction.py /
transfer.py / logger.py / run.py run unmodified on top of the drop-in packages, kernels on the CPU emulator.
TEST INFRASTRUCTURE ONLY (uses oracle/ and the reference tree; authoring container only)."""
import copy
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "monkey-net_amd")
REF = os.environ.get("MNK_REFERENCE_ROOT", "/root/reference")
sys.path[:] = [PKG, REF, ROOT, os.path.join(ROOT, "tests")] + [p for p in sys.path if p not in ("", ROOT)]

import torch  # noqa: E402

from oracle import ref_shim, cases, restate  # noqa: E402

ref_shim.install_stubs()           # imageio / skimage / torchvision ... (absent third-party packages of the callers)
ref_shim.install_torch_pins()      # torch.gesv (transfer.py:51 calls it directly), grid_sample align_corners
from conftest import emu_library_path  # noqa: E402
from mnk import _lib  # noqa: E402

__import__('_util').set_library(emu_library_path(), strict=False)


def load_gold(name):
    return torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)


def build(cfg):
    from modules.generator import MotionTransferGenerator
    from modules.discriminator import Discriminator
    from modules.keypoint_detector import KPDetector
    mp = cfg["model_params"]
    torch.manual_seed(0)
    gen = MotionTransferGenerator(**mp["generator_params"], **mp["common_params"])
    disc = Discriminator(**mp["discriminator_params"], **mp["common_params"])
    kpd = KPDetector(**mp["kp_detector_params"], **mp["common_params"])
    return gen, disc, kpd


def scenario_imports():
    """`import run` = every import statement of the reference's entry script (run.py:1-18)."""
    import run as ref_run
    import modules.generator, modules.prediction_module, modules.losses, sync_batchnorm, train, prediction
    ours = lambda m: os.path.abspath(m.__file__).startswith(PKG)
    assert ours(modules.generator) and ours(modules.losses) and ours(sync_batchnorm)
    assert os.path.abspath(modules.prediction_module.__file__).startswith(REF)      # falls through to the reference
    assert os.path.abspath(train.__file__).startswith(REF) and os.path.abspath(ref_run.__file__).startswith(REF)
    assert ours(sys.modules[train.DataParallelWithCallback.__module__])
    assert ours(sys.modules[ref_run.MotionTransferGenerator.__module__])
    assert ours(sys.modules[ref_run.KPDetector.__module__]) and ours(sys.modules[ref_run.Discriminator.__module__])
    return {"ok": True}


def scenario_train():
    """The reference's own train() (train.py:78-153: its optimisers, schedulers, DataLoader, DataParallelWithCallback with
    device_ids, Logger.log_iter / log_epoch / save_cpk) for three one-iteration epochs on the drop-in modules, against the
    loss history the reference recorded with ITS modules (tests/golden/step_tiny.pt); then Logger.load_cpk round trip."""
    import logger as ref_logger
    import train as ref_train
    gold = load_gold("step_tiny")
    cfg = copy.deepcopy(gold["cfg"])
    cfg["train_params"].update(num_epochs=3, epoch_milestones=[], batch_size=gold["batch"],
                               log_params={"log_freq_iter": 10 ** 9, "cpk_freq_epoch": 1})
    cfg["visualizer_params"] = {}
    gen, disc, kpd = build(cfg)
    gen.load_state_dict(gold["state"]["generator"]), disc.load_state_dict(gold["state"]["discriminator"])
    kpd.load_state_dict(gold["state"]["kp_detector"])
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])

    class Pairs(torch.utils.data.Dataset):
        def __len__(self):
            return src.shape[0]

        def __getitem__(self, i):
            return {"source": src[i], "video": drv[i]}

    seen = []

    class QuietVisualizer:                      # the drawing code needs skimage; not on the path under test
        def __init__(self, **kw):
            pass

    class RecordingLogger(ref_logger.Logger):   # the reference's Logger with the GIF writer switched off
        def log_iter(self, it, names, values, inp, out):
            seen.append({"names": list(names), "values": [float(v) for v in values]})
            self.it = it

    ref_logger.Visualizer = QuietVisualizer
    ref_train.Logger = RecordingLogger
    real_loader = ref_train.DataLoader
    ref_train.DataLoader = lambda ds, **kw: real_loader(ds, **dict(kw, num_workers=0, shuffle=False))
    log_dir = tempfile.mkdtemp(prefix="mnk_dropin_")
    ref_train.train(cfg, gen, disc, kpd, None, log_dir, Pairs(), device_ids=[0])
    report = []
    for it, (rec, r32, r64) in enumerate(zip(seen, gold["history"], gold["history64"])):
        a32, a64 = r32["generator"] + r32["discriminator"], r64["generator"] + r64["discriminator"]
        spread = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(a32, a64))
        err = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(rec["values"], a64))
        assert len(rec["values"]) == len(a64)
        assert err <= 16.0 * spread + 2e-5, (it, err, spread)        # the bound of tests/test_step.py
        report.append((it, err, spread))
    assert len(seen) == 3
    from modules.losses import generator_loss_names, discriminator_loss_names
    assert seen[0]["names"] == generator_loss_names(cfg["train_params"]["loss_weights"]) + discriminator_loss_names()
    # checkpoint written by the reference's Logger.save_cpk (logger.py:43-47), read back by Logger.load_cpk (:49-66)
    cpks = sorted(f for f in os.listdir(log_dir) if f.endswith("checkpoint.pth.tar"))
    assert cpks, os.listdir(log_dir)
    real_load = torch.load
    torch.load = lambda f, **kw: real_load(f, **dict(kw, weights_only=False))   # torch 0.4.1 semantics of logger.py:52
    try:
        gen2, disc2, kpd2 = build(cfg)
        opts = [torch.optim.Adam(m.parameters(), lr=1e-4, betas=(0.5, 0.999)) for m in (gen2, disc2, kpd2)]
        epoch, it = ref_logger.Logger.load_cpk(os.path.join(log_dir, cpks[-1]), gen2, disc2, kpd2, *opts)
    finally:
        torch.load = real_load
    assert (epoch, it) == (2, 2)
    for a, b in ((gen, gen2), (disc, disc2), (kpd, kpd2)):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    assert opts[0].state_dict()["state"][0]["exp_avg"].shape == next(gen2.parameters()).shape
    return {"report": report, "checkpoint": cpks[-1]}


def _eval_models(name="tiny"):
    gold = load_gold(name)
    gen, disc, kpd = build(gold["cfg"])
    gen.load_state_dict(gold["state"]["generator"]), kpd.load_state_dict(gold["state"]["kp_detector"])
    return gold, gen, kpd


def scenario_reconstruction():
    """reconstruction.py:45-61 (DataParallelWithCallback without device_ids, .eval(), per-frame kp detector calls,
    generate()) against the reference's eval run recorded in tests/golden/tiny.pt, incl. the L1 criterion of :74."""
    import reconstruction as ref_rec
    from sync_batchnorm import DataParallelWithCallback
    gold, gen, kpd = _eval_models()
    generator, kp_detector = DataParallelWithCallback(gen), DataParallelWithCallback(kpd)
    generator.eval(), kp_detector.eval()
    src, drv = cases.smooth_pair(gold["batch"], gold["size"], gold["size"])
    video = torch.cat([src, drv], dim=2)
    cat_dict = lambda l, dim: {k: torch.cat([v[k] for v in l], dim=dim) for k in l[0]}
    with torch.no_grad():
        kp_appearance = kp_detector(video[:, :, :1])
        kp_video = cat_dict([kp_detector(video[:, :, i:(i + 1)]) for i in range(video.shape[2])], dim=1)
        out = ref_rec.generate(generator, appearance_image=video[:, :, :1], kp_appearance=kp_appearance,
                               kp_video=kp_video)
        ref = gold["eval64"]
        pred = out["video_prediction"][:, :, 1:2]                       # frame 1 = the golden's driving frame
        err = float((pred.double() - ref["video_prediction"].double()).abs().max())
        assert err < 2e-5, err
        assert float((kp_video["mean"].double() - ref["kp_mean"].double()).abs().max()) < 2e-6
        l1 = float(ref_rec.reconstruction_loss(pred.cpu(), drv.cpu(), 1).mean())
        l1_ref = float((ref["video_prediction"].double() - drv.double()).abs().mean())
        assert abs(l1 - l1_ref) < 1e-4, (l1, l1_ref)
        # frame 0 drives the source onto itself: identity key-point motion
        assert out["video_prediction"].shape == (gold["batch"], 3, 2, gold["size"], gold["size"])
    return {"max_err": err, "l1": l1, "l1_ref": l1_ref}


def scenario_transfer():
    """transfer.py:65-79 transfer_one incl. normalize_kp (:31-62: move_location, adapt_variance through the public
    modules.util.matrix_inverse, make_symetric_matrix) on the drop-in modules, against the same reference functions fed
    with the oracle's key-points / generator (oracle/restate.py, eval mode)."""
    import transfer as ref_transfer
    gold, gen, kpd = _eval_models()
    gen.eval(), kpd.eval()
    cfg = gold["cfg"]
    mp = cfg["model_params"]
    g = torch.Generator().manual_seed(5)
    src, _ = cases.smooth_pair(2, gold["size"], gold["size"], seed=11)
    frames = [cases.smooth_pair(2, gold["size"], gold["size"], seed=20 + i)[1] for i in range(3)]
    driving = torch.cat(frames, dim=2)
    params = {"normalization_params": {"movement_mult": False, "move_location": True, "adapt_variance": True,
                                       "clip_mean": True}}
    with torch.no_grad():
        out = ref_transfer.transfer_one(gen, kpd, src, driving, params)
    sds = {"generator": gold["state"]["generator"], "kp_detector": gold["state"]["kp_detector"]}

    def oracle_kp(x):
        return restate.kp_detector_forward(sds["kp_detector"], dict(mp["kp_detector_params"], **mp["common_params"]), x,
                                           training=False)

    def oracle_gen(source_image, kp_driving, kp_source):
        return restate.generator_forward(sds["generator"], mp["generator_params"], mp["common_params"], source_image,
                                         kp_driving, kp_source, training=False)

    with torch.no_grad():
        exp = ref_transfer.transfer_one(oracle_gen, oracle_kp, src, driving, params)
    errs = {}
    for k in ("video_prediction", "video_deformed"):
        errs[k] = float((out[k].double() - exp[k].double()).abs().max())
    for k in ("mean", "var"):
        errs["kp_norm_" + k] = float((out["kp_norm"][k].double() - exp["kp_norm"][k].double()).abs().max())
    assert out["video_prediction"].shape == (2, 3, 3, gold["size"], gold["size"])
    assert errs["video_prediction"] < 5e-5 and errs["video_deformed"] < 2e-4, errs
    assert errs["kp_norm_mean"] < 5e-6 and errs["kp_norm_var"] < 5e-6, errs
    return errs


def scenario_transfer_batched():
    """mnk.engine.Transfer (one detector call, device-side normalize_kp, one generator call) and mnk.engine.normalize_kp
    against the REFERENCE's transfer_one / normalize_kp (transfer.py:31-79, scipy convex hulls, numpy eigen-decomposition)
    running frame by frame on the same drop-in modules -- for every normalisation variant of the shipped configs."""
    import transfer as ref_transfer
    from mnk import engine
    gold, gen, kpd = _eval_models()
    gen.eval(), kpd.eval()
    src, _ = cases.smooth_pair(2, gold["size"], gold["size"], seed=11)
    driving = torch.cat([cases.smooth_pair(2, gold["size"], gold["size"], seed=20 + i)[1] for i in range(3)], dim=2)
    worst = {}
    variants = [dict(movement_mult=False, move_location=True, adapt_variance=True, clip_mean=True),
                dict(movement_mult=True, move_location=True, adapt_variance=True, clip_mean=False),
                dict(movement_mult=False, move_location=False, adapt_variance=False, clip_mean=False),
                dict(movement_mult=True, move_location=True, adapt_variance=False, clip_mean=True)]
    for params in variants:
        with torch.no_grad():
            exp = ref_transfer.transfer_one(gen, kpd, src, driving, {"normalization_params": params})
        got = engine.Transfer(kpd, gen, params)(src, driving)
        for k in ("video_prediction", "video_deformed"):
            e = float((got[k].double() - exp[k].double()).abs().max())
            worst[k] = max(worst.get(k, 0.0), e)
        for k in ("mean", "var"):
            e = float((got["kp_norm"][k].double() - exp["kp_norm"][k].double()).abs().max())
            worst["kp_norm_" + k] = max(worst.get("kp_norm_" + k, 0.0), e)
    # the covariance repair: a covariance product with a non-positive eigenvalue (make_symetric_matrix, transfer.py:17-28)
    g = torch.Generator().manual_seed(8)
    kp_v = cases.random_kp(2, 3, 4, seed=31)
    kp_a = cases.random_kp(2, 1, 4, seed=32)
    kp_a["var"][0, 0, 1] = torch.tensor([[0.02, 0.05], [0.05, 0.01]])        # indefinite after the transfer
    kp_a["var"][1, 0, 2] = torch.tensor([[-0.03, 0.0], [0.0, 0.02]])
    params = dict(movement_mult=False, move_location=True, adapt_variance=True, clip_mean=False)
    exp = ref_transfer.normalize_kp(kp_v, kp_a, **params)
    got = engine.normalize_kp(kp_v, kp_a, **params)
    worst["repair_var"] = float((got["var"].double() - exp["var"].double()).abs().max())
    worst["repair_mean"] = float((got["mean"].double() - exp["mean"].double()).abs().max())
    assert worst["video_prediction"] < 2e-5 and worst["video_deformed"] < 2e-4, worst
    assert worst["kp_norm_mean"] < 2e-6 and worst["kp_norm_var"] < 2e-6, worst
    assert worst["repair_var"] < 2e-6 and worst["repair_mean"] < 1e-6, worst
    return worst


if __name__ == "__main__":
    res = globals()["scenario_" + sys.argv[1]]()
    print("DROPIN_RESULT " + json.dumps(res))
