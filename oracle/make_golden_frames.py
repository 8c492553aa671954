"""Fixtures of the INPUT path from the REAL reference:

    python -m oracle.make_golden_frames        -> tests/golden/frames_shapes.npz, tests/golden/step_shapes_frames.pt

* frames_shapes.npz -- the first eight videos of the reference's own data/shapes/train (BASELINE configs[0]; stacked-frame
  PNGs, 2048 x 64 RGBA = 32 frames of 64 x 64) as decoded uint8 strips, and what the UNMODIFIED reference input pipeline --
  frames_dataset.FramesDataset.__getitem__ with augmentation.AllAugmentationTransform (frames_dataset.py:14-88,
  augmentation.py:91-171,324-389) -- returns for them under fixed seeds of `random` / `numpy.random`, for
    "cfg"    config/shapes.yaml's augmentation (time + horizontal flip, crop 64x64),
    "crop48" a random 48x48 crop (random.randint draws), "pad80" a crop larger than the frame (pad_clip, mode='edge'),
    "eval"   is_train=False (VideoToTensor: every frame).
  The third-party packages the reference imports are absent from this image; the few functions the exact transforms touch
  are pinned from outside (as oracle/ref_shim.py does for torch): skimage.io.imread = PIL decode, skimage.img_as_float32 =
  uint8 * float32(1 / 255) (scikit-image 0.14 `convert`: np.multiply(image, 1. / 255, dtype=float32)), gray2rgb =
  replicate, skimage.util.pad = numpy.pad (0.14 re-exports it).  resize / rotate / ColorJitter are not touched.
* step_shapes_frames.pt -- three full training iterations of config/shapes.yaml on a batch of REAL frames (the eight samples
  of the "cfg" record at seed 0) by the reference's own GeneratorFullModel / DiscriminatorFullModel + torch.optim.Adam in
  fp32 and fp64 (the step_tiny record of make_golden.py, on real data), and a checkpoint in the reference's
  Logger.save_cpk layout (logger.py:43-47) written after those steps.
TEST INFRASTRUCTURE ONLY."""
import copy
import os
import random
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shim, cases  # noqa: E402
from oracle.make_golden import build_reference, load_cfg, save, GOLD  # noqa: E402

SEEDS = (0, 1, 2)


def pin_third_party():
    """after ref_shim.install(): the functions of the absent packages that the exact transforms call"""
    from PIL import Image

    def imread(name):
        with Image.open(name) as im:
            return np.array(im)

    def img_as_float32(image):
        assert image.dtype == np.uint8
        return np.multiply(image, 1. / 255, dtype=np.float32)

    sys.modules["skimage.io"].imread = imread
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    sys.modules["skimage"].img_as_float32 = img_as_float32
    sys.modules["skimage.color"].gray2rgb = lambda a: np.stack([a, a, a], axis=-1)
    sys.modules["skimage.util"].pad = np.pad


def run_steps(ref, cfg, x32, steps, dtype):
    gen, disc, kpd, _ = build_reference(ref, cfg)
    tp = cfg["train_params"]
    for m in (gen, disc, kpd):
        m.to(dtype)
    gfull = ref.GeneratorFullModel(kpd, gen, disc, tp)
    dfull = ref.DiscriminatorFullModel(kpd, gen, disc, tp)
    og = torch.optim.Adam(gen.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    od = torch.optim.Adam(disc.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    ok = torch.optim.Adam(kpd.parameters(), lr=tp["lr"], betas=(0.5, 0.999))
    x = {k: v.to(dtype) for k, v in x32.items()}
    hist = []
    for it in range(steps):                              # train.py:110-136
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
        hist.append({"generator": gl, "discriminator": [float(v.detach()) for v in dl]})
    return hist, (gen, disc, kpd, og, od, ok)


def main():
    assert ref_shim.available(), "run this in the authoring container (needs /root/reference)"
    torch.set_num_threads(8)
    ref = ref_shim.load()
    pin_third_party()
    import frames_dataset                                  # the reference's own module
    src_dir = os.path.join(ref_shim.REFERENCE_ROOT, "data", "shapes")
    names = sorted(os.listdir(os.path.join(src_dir, "train")))[:8]
    cfg = load_cfg("shapes")
    aug = cfg["dataset_params"]["augmentation_params"]
    variants = {"cfg": (aug, True), "crop48": (dict(aug, crop_param={"size": [48, 48]}), True),
                "pad80": (dict(aug, crop_param={"size": [80, 80]}), True), "eval": (aug, False)}
    out = {"names": np.array(names)}
    with tempfile.TemporaryDirectory() as tmp:
        for sub in ("train", "test"):
            os.makedirs(os.path.join(tmp, sub))
            for n in names:
                os.symlink(os.path.join(src_dir, "train", n), os.path.join(tmp, sub, n))
        from PIL import Image
        for i, n in enumerate(names):
            with Image.open(os.path.join(src_dir, "train", n)) as im:
                out["strip%d" % i] = np.array(im)
        for tag, (params, is_train) in variants.items():
            ds = frames_dataset.FramesDataset(root_dir=tmp, augmentation_params=params, image_shape=(64, 64, 3),
                                              is_train=is_train)
            order = [names.index(n) for n in ds.images]      # os.listdir order of the temporary directory
            out["order_" + tag] = np.array(order)
            for seed in SEEDS if is_train else SEEDS[:1]:
                random.seed(seed)
                np.random.seed(seed)
                for idx in range(len(ds) if is_train else 2):
                    item = ds[idx]
                    assert item["name"] == ds.images[idx]
                    for k in ("source", "video"):
                        if k in item:
                            out["%s_s%d_i%d_%s" % (tag, seed, idx, k)] = np.ascontiguousarray(item[k])
            print(tag, "done", flush=True)
    np.savez_compressed(os.path.join(GOLD, "frames_shapes.npz"), **out)
    # ---- three training iterations of config/shapes.yaml on the real frames of the "cfg" record, seed 0 ----------------------
    order = list(out["order_cfg"])
    src = torch.from_numpy(np.stack([out["cfg_s0_i%d_source" % i] for i in range(8)]))
    drv = torch.from_numpy(np.stack([out["cfg_s0_i%d_video" % i] for i in range(8)]))
    x = {"source": src, "video": drv}
    hist32, _ = run_steps(ref, cfg, x, 3, torch.float32)
    hist64, _ = run_steps(ref, cfg, x, 3, torch.float64)
    # the same three iterations with the small TINY networks (oracle/cases.py): their checkpoint -- written in the reference's
    # Logger.save_cpk layout (logger.py:43-47): {name: state_dict of every model and optimiser, 'epoch', 'it'} -- is small
    # enough to commit (config/shapes.yaml's is 59 MB)
    tiny = copy.deepcopy(cases.TINY)
    thist32, models = run_steps(ref, tiny, x, 3, torch.float32)
    gen, disc, kpd, og, od, ok = models
    cpk = {"generator": gen.state_dict(), "discriminator": disc.state_dict(), "kp_detector": kpd.state_dict(),
           "optimizer_generator": og.state_dict(), "optimizer_discriminator": od.state_dict(),
           "optimizer_kp_detector": ok.state_dict(), "epoch": 0, "it": 3}
    cpk = copy.deepcopy(cpk)
    # one more iteration from that state: what a run resumed from the checkpoint must reproduce
    next_losses = None
    tp = tiny["train_params"]
    gfull = ref.GeneratorFullModel(kpd, gen, disc, tp)
    outs = gfull(x)
    next_losses = [float(v.mean()) for v in outs[:-2]]
    with torch.no_grad():                                   # and the restored models' evaluation forward (fp32)
        gen.eval(), kpd.eval()
        gen.load_state_dict(cpk["generator"]), kpd.load_state_dict(cpk["kp_detector"])
        kp_s, kp_d = kpd(src), kpd(drv)
        pred = gen(src, kp_driving=kp_d, kp_source=kp_s)["video_prediction"]
    save("step_shapes_frames", {"cfg": cfg, "batch": 8, "size": 64, "history": hist32, "history64": hist64, "order": order,
                                "tiny_cfg": tiny, "tiny_history": thist32, "tiny_checkpoint": cpk,
                                "tiny_next_generator_losses": next_losses, "tiny_eval_prediction_after": pred,
                                "tiny_eval_kp_mean_after": kp_d["mean"]})
    print("losses fp32", hist32[0]["generator"], "fp64", hist64[0]["generator"])
    for f in ("frames_shapes.npz", "step_shapes_frames.pt"):
        print(f, os.path.getsize(os.path.join(GOLD, f)))


if __name__ == "__main__":
    main()
