"""Device-side input path: mnk.frames.DeviceFramesDataset / mnk_frames_gather against what the
UNMODIFIED reference input pipeline (frames_dataset.FramesDataset + augmentation.AllAugmentationTransform) returned for the
first eight videos of its own data/shapes/train (tests/golden/frames_shapes.npz, made by oracle/make_golden_frames.py) --
bit for bit: frame selection, time / horizontal flip, random crop, edge padding, RGBA -> RGB, uint8 -> float32, layout.
Plus: three training iterations of config/shapes.yaml on those REAL frames (BASELINE configs[0]) against the reference's
loss history, and a checkpoint written in the reference's Logger.save_cpk layout (logger.py:43-66) restored into the
drop-in modules and optimisers."""
import copy
import math
import os
import random
import struct
import zlib

import numpy as np
import pytest
import torch

from test_modules import build, load

GOLD = os.path.join(os.path.dirname(__file__), "golden")
VARIANTS = {"cfg": ({"flip_param": {"time_flip": True, "horizontal_flip": True}, "crop_param": {"size": [64, 64]}}, True),
            "crop48": ({"flip_param": {"time_flip": True, "horizontal_flip": True}, "crop_param": {"size": [48, 48]}}, True),
            "pad80": ({"flip_param": {"time_flip": True, "horizontal_flip": True}, "crop_param": {"size": [80, 80]}}, True),
            "eval": ({"flip_param": {"time_flip": True, "horizontal_flip": True}, "crop_param": {"size": [64, 64]}}, False)}


def _write_png(path, arr):
    """8-bit PNG, filter 0, colour type by channel count (test helper: the GPU box has no reference data set)"""
    h, w = arr.shape[:2]
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    raw = b"".join(b"\x00" + arr[r].tobytes() for r in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


@pytest.fixture(scope="module")
def shapes(tmp_path_factory):
    fx = np.load(os.path.join(GOLD, "frames_shapes.npz"))
    root = tmp_path_factory.mktemp("shapes")
    names = [str(n) for n in fx["names"]]
    for sub in ("train", "test"):
        os.makedirs(os.path.join(root, sub))
        for i, n in enumerate(names):
            _write_png(os.path.join(root, sub, n), fx["strip%d" % i])
    return fx, str(root), names


def test_png_decoder(tmp_path):
    from mnk import frames
    g = np.random.RandomState(0)
    for ch in (1, 2, 3, 4):
        arr = g.randint(0, 256, size=(7, 13) + ((ch,) if ch > 1 else ()), dtype=np.uint8)
        p = os.path.join(tmp_path, "f0_%d.png" % ch)
        _write_png(p, arr)
        assert np.array_equal(frames.decode_png(p), arr)
    try:
        from PIL import Image
    except ImportError:
        return
    # PIL picks the five row filters adaptively: smooth gradients + noise exercise Sub / Up / Average / Paeth
    yy, xx = np.mgrid[0:33, 0:47]
    for ch, mode in ((1, "L"), (3, "RGB"), (4, "RGBA")):
        arr = np.stack([(xx * (k + 2) + yy * 3 + g.randint(0, 4, size=xx.shape)) % 256 for k in range(ch)], -1).astype(np.uint8)
        arr = arr[..., 0] if ch == 1 else arr
        p = os.path.join(tmp_path, "pil_%d.png" % ch)
        Image.fromarray(arr, mode).save(p, optimize=True)
        assert np.array_equal(frames.decode_png(p), arr)
    ref_png = "/root/reference/data/shapes/train/00000000.png"
    if os.path.exists(ref_png):
        with Image.open(ref_png) as im:
            assert np.array_equal(frames.decode_png(ref_png), np.array(im))


@pytest.mark.parametrize("tag", ["cfg", "crop48", "pad80", "eval"])
def test_samples_equal_the_reference_pipeline_bit_for_bit(be, shapes, tag):
    from mnk import frames
    fx, root, names = shapes
    params, is_train = VARIANTS[tag]
    order = [names[i] for i in fx["order_" + tag]]
    ds = frames.DeviceFramesDataset(root, params, image_shape=(64, 64, 3), is_train=is_train, device=be.device, files=order)
    assert len(ds) == 8
    checked = 0
    for seed in (0, 1, 2) if is_train else (0,):
        random.seed(seed)
        np.random.seed(seed)
        for idx in range(8 if is_train else 2):
            item = ds[idx]
            assert item["name"] == order[idx]
            for k in ("source", "video"):
                key = "%s_s%d_i%d_%s" % (tag, seed, idx, k)
                if key not in fx:
                    assert k not in item
                    continue
                ref = torch.from_numpy(fx[key])
                got = item[k].cpu()
                assert got.shape == ref.shape and got.dtype == torch.float32, (key, got.shape, ref.shape)
                assert torch.equal(got, ref), (key, float((got - ref).abs().max()))
                checked += 1
    assert checked >= (48 if is_train else 2)


def test_a_batch_is_one_launch_of_the_same_samples(be, shapes):
    from mnk import frames
    fx, root, names = shapes
    params, _ = VARIANTS["crop48"]
    order = [names[i] for i in fx["order_crop48"]]
    ds = frames.DeviceFramesDataset(root, params, image_shape=(64, 64, 3), is_train=True, device=be.device, files=order)
    random.seed(5), np.random.seed(5)
    singles = [ds[i] for i in (3, 0, 7, 7)]
    random.seed(5), np.random.seed(5)
    b = ds.batch([3, 0, 7, 7])
    assert b["source"].shape == (4, 3, 1, 48, 48) and b["video"].shape == (4, 3, 1, 48, 48)
    assert b["name"] == [s["name"] for s in singles]
    for k in ("source", "video"):
        assert torch.equal(b[k].cpu(), torch.stack([s[k].cpu() for s in singles]))
    loader = frames.DeviceLoader(ds, batch_size=3, shuffle=True, drop_last=True, generator=torch.Generator().manual_seed(1))
    seen = [x["name"] for x in loader]
    assert len(loader) == 2 and len(seen) == 2 and all(len(n) == 3 for n in seen) and len(set(sum(seen, []))) == 6
    ev = frames.DeviceFramesDataset(root, params, image_shape=(64, 64, 3), is_train=False, device=be.device, files=order)
    vb = ev.batch([1, 2])
    assert vb["video"].shape == (2, 3, 32, 64, 64) and "source" not in vb


def test_gif_videos_give_the_samples_of_the_same_frames_stacked_as_a_png_strip(be, shapes, tmp_path):
    """frames_dataset.py:30-36: a .gif video (the moving-gif data set's format) is its frames, composited and converted from the
    palette to RGB.  The fixture's strips (few colours: lossless in a GIF palette) are written as animated GIFs -- once with a
    global palette and full frames, once with a transparent index and partial-frame optimisation, which a decoder has to
    composite -- and every sample must equal, bit for bit, the sample of the PNG strip of the same frames."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from mnk import frames
    fx, root, names = shapes
    params, _ = VARIANTS["crop48"]
    for variant in ("plain", "optimised"):
        gdir = os.path.join(tmp_path, variant, "train")
        os.makedirs(gdir), os.makedirs(os.path.join(tmp_path, variant, "test"))
        pdir = os.path.join(tmp_path, variant + "_png", "train")
        os.makedirs(pdir), os.makedirs(os.path.join(tmp_path, variant + "_png", "test"))
        gnames = []
        for i, n in enumerate(names[:4]):
            strip = fx["strip%d" % i][:, :, :3]
            fr = [np.ascontiguousarray(strip[:, k * 64:(k + 1) * 64]) for k in range(strip.shape[1] // 64)]
            # (Pillow's GIF writer folds a frame that equals its predecessor into the predecessor's duration: compare on
            # frame lists without such repeats)
            fr = [f for k, f in enumerate(fr) if k == 0 or not np.array_equal(f, fr[k - 1])]
            strip = np.concatenate(fr, axis=1)
            _write_png(os.path.join(pdir, n), strip)
            assert len(np.unique(strip.reshape(-1, 3), axis=0)) <= 255
            ims = [Image.fromarray(f, "RGB") for f in fr]
            g = os.path.join(gdir, n.replace(".png", ".gif"))
            if variant == "plain":
                ims[0].save(g, save_all=True, append_images=ims[1:], loop=0, duration=40, optimize=False, disposal=1)
            else:
                ims[0].save(g, save_all=True, append_images=ims[1:], loop=0, duration=40, optimize=True, disposal=1)
            gnames.append(os.path.basename(g))
            assert np.array_equal(frames.read_strip(g), strip), (variant, n)
        dg = frames.DeviceFramesDataset(os.path.join(tmp_path, variant), params, image_shape=(64, 64, 3), is_train=True,
                                        device=be.device, files=gnames)
        dp = frames.DeviceFramesDataset(os.path.join(tmp_path, variant + "_png"), params, image_shape=(64, 64, 3), is_train=True,
                                        device=be.device, files=names[:4])
        random.seed(9), np.random.seed(9)
        a = dg.batch([0, 1, 2, 3, 1])
        random.seed(9), np.random.seed(9)
        b = dp.batch([0, 1, 2, 3, 1])
        for k in ("source", "video"):
            assert torch.equal(a[k].cpu(), b[k].cpu()), (variant, k)
    with pytest.raises(NotImplementedError):
        frames.read_strip(os.path.join(tmp_path, "clip.mp4"))


def test_random_train_test_split_is_scikit_learns(be, shapes, tmp_path):
    """frames_dataset.py:61-63: a data directory without train/ and test/ is split by sklearn's train_test_split(images,
    random_state=random_seed, test_size=0.2) -- restated in mnk.frames.split_train_test and compared with scikit-learn itself."""
    sk = pytest.importorskip("sklearn.model_selection")
    from mnk import frames
    for n in (1, 2, 5, 8, 10, 33, 101):
        items = ["v%03d.png" % i for i in range(n)]
        for seed in (0, 1, 7):
            if n == 1:
                continue                                   # (sklearn refuses an empty train set)
            tr, te = sk.train_test_split(items, random_state=seed, test_size=0.2)
            assert frames.split_train_test(items, seed) == (tr, te), (n, seed)
    fx, root, names = shapes
    flat = os.path.join(tmp_path, "flat")
    os.makedirs(flat)
    for i, n in enumerate(names):
        _write_png(os.path.join(flat, n), fx["strip%d" % i])
    listing = os.listdir(flat)
    tr, te = sk.train_test_split(listing, random_state=3, test_size=0.2)
    assert frames.DeviceFramesDataset(flat, None, is_train=True, random_seed=3, device=be.device).images == tr
    assert frames.DeviceFramesDataset(flat, None, is_train=False, random_seed=3, device=be.device).images == te


def test_transforms_without_a_device_form_raise(shapes):
    from mnk import frames
    fx, root, names = shapes
    for bad in ({"resize_param": {"ratio": [0.2, 1.1]}},):           # (round 5: every term of ColorJitter and the anti-aliased
        # resize down to ratio 0.32 have device forms; below that the Gaussian needs more than nine taps)
        with pytest.raises(NotImplementedError):
            frames.DeviceFramesDataset(root, bad, device="cpu", files=names)
    # the per-axis scale is size / int(size * f): a 30-pixel axis at f = 0.32 is 9 pixels, scale 3.33, radius 5 -- refused by the
    # constructor, not by a training batch; the same ratio on the fixture's 64-pixel frames is fine
    with pytest.raises(NotImplementedError, match="radius 5"):
        frames.DeviceFramesDataset(root, {"resize_param": {"ratio": [0.32, 1.0]}}, image_shape=(30, 30, 3), device="cpu", files=names)


AUG = {   # config/moving-gif.yaml:5-14 (at the 64x64 frames of the fixture), config/actions.yaml:5-16, and the order-1 resize
    "moving-gif": {"flip_param": {"horizontal_flip": True, "time_flip": True}, "crop_param": {"size": [64, 64]},
                   "resize_param": {"ratio": [0.9, 1.1]}, "jitter_param": {"hue": 0.5}},
    "actions": {"flip_param": {"time_flip": True, "horizontal_flip": True}, "crop_param": {"size": [64, 64]},
                "resize_param": {"ratio": [0.9, 1.1]}, "jitter_param": {"hue": 0.5}, "rotation_param": {"degrees": [-10, 10]}},
    "bilinear": {"crop_param": {"size": [56, 72]}, "resize_param": {"ratio": [0.85, 1.2], "interpolation": "bilinear"},
                 "rotation_param": {"degrees": 25}},
    "hue-only": {"jitter_param": {"hue": 0.3}},
    "full-jitter": {"crop_param": {"size": [64, 64]}, "resize_param": {"ratio": [0.9, 1.1]},
                    "jitter_param": {"brightness": 0.4, "contrast": 0.5, "saturation": 0.6, "hue": 0.2}},
    "contrast-rotated": {"rotation_param": {"degrees": 15}, "jitter_param": {"contrast": 0.8, "brightness": 0.9}},
    # RandomResize's DEFAULT ratio (3/4, 4/3) and stronger down-scalings: skimage's multi-tap anti-aliasing filter (round 5)
    "aa-default-ratio": {"crop_param": {"size": [64, 64]}, "resize_param": {}},
    "aa-nearest": {"crop_param": {"size": [48, 40]}, "resize_param": {"ratio": [0.4, 0.78]}, "jitter_param": {"hue": 0.2}},
    "aa-bilinear-rotated": {"crop_param": {"size": [40, 56]}, "resize_param": {"ratio": [0.35, 0.75], "interpolation": "bilinear"},
                            "rotation_param": {"degrees": 20}},
}


@pytest.mark.parametrize("tag", sorted(AUG))
def test_resize_rotation_and_hue_jitter_equal_the_restated_library_arithmetic(be, shapes, tag):
    """the augmentation_params of config/moving-gif.yaml / actions.yaml through DeviceFramesDataset (mnk_frames_augment) against
    oracle/augment_restate.py -- skimage 0.14.0's resize / rotate, Pillow 5.2.0's HSV conversions, torchvision 0.2.1's
    adjust_hue restated in numpy; those packages are absent here, so the restatement itself is unpinned -- with the SAME random
    draws.  Tolerance: the warps run in float64 on both sides and differ only by libm (cos / sin / fmod): |diff| <= 1e-6 without
    the jitter; with it the values are uint8 levels / 255 and a level can flip where x * 255 lands within ~1e-9 of a rounding
    boundary: at most 0.01 % of the values may differ, by one hue step's worth of colour at most."""
    from mnk import frames
    from oracle import augment_restate as ar
    fx, root, names = shapes
    params = AUG[tag]
    ds = frames.DeviceFramesDataset(root, params, image_shape=(64, 64, 3), is_train=True, device=be.device, files=names)
    strips = [fx["strip%d" % i] for i in range(len(names))]
    total = differ = 0
    worst = 0.0
    for seed in (0, 1, 2):
        for idx in range(len(names)):
            random.seed(100 * seed + idx), np.random.seed(100 * seed + idx)
            item = ds[idx]
            random.seed(100 * seed + idx), np.random.seed(100 * seed + idx)
            sel, hflip, x1, y1, pt, pl, oh, ow, angle, new_hw, hue, jit = ds._draw(ds.meta[idx][3])
            strip = strips[idx][:, :, :3]
            frames_u8 = np.moveaxis(strip.reshape(64, -1, 64, 3), 1, 0)          # read_video: (F, H, W, 3)
            ref = ar.pipeline(frames_u8, sel, hflip, angle, new_hw, ds.crop, x1, y1, hue, resize_order=ds.resize_order, jitter=jit)
            got = torch.cat([item["source"], item["video"]], dim=1).cpu().numpy()
            assert got.shape == ref.shape, (got.shape, ref.shape)
            d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
            total += d.size
            differ += int((d > 1e-6).sum())
            worst = max(worst, float(d.max()))
    if "jitter_param" in params:
        assert differ <= 1e-4 * total, (tag, differ, total, worst)
    else:
        assert worst <= 1e-6, (tag, worst)
    print("%s: %d values, %d differ by more than 1e-6, largest difference %.3e" % (tag, total, differ, worst))


def test_anti_aliasing_restatement_equals_the_installed_scipy():
    """skimage 0.14's resize calls scipy.ndimage.gaussian_filter before it samples; oracle/augment_restate.py::gaussian_aa restates
    that filter (kernel construction, the symmetric correlate1d's order of additions, rows then columns, zero padding) and must
    equal the REAL scipy bit for bit -- the filter half of the anti-aliased resize is pinned to a real library (the sampling
    half, skimage's warp, is not installable here)"""
    ndi = pytest.importorskip("scipy.ndimage")
    from oracle import augment_restate as ar
    rng = np.random.RandomState(3)
    for trial in range(24):
        h, w = rng.randint(12, 60, size=2)
        img = rng.rand(h, w, 3)
        fr, fc = rng.uniform(0.8, 3.2, size=2)
        want = ndi.gaussian_filter(img, (max(0.0, (fr - 1) / 2), max(0.0, (fc - 1) / 2), 0), cval=0, mode="constant")
        assert np.array_equal(ar.gaussian_aa(img, fr, fc), want), (trial, fr, fc)


def _ndi_rotate(ndi, img, angle_deg, order=1):
    """skimage.transform.rotate's map (centre (cols/2 - 0.5, rows/2 - 0.5), augmentation.py:175-214) evaluated by
    scipy.ndimage.affine_transform -- a third-party implementation of the same interior arithmetic -- plus the mask of the
    output pixels whose taps all lie inside the source frame (the border rule, skimage's per-tap cval, is NOT scipy's)"""
    rows, cols = img.shape[:2]
    cx, cy = cols / 2.0 - 0.5, rows / 2.0 - 0.5
    a = math.radians(angle_deg)
    co, si = math.cos(a), math.sin(a)
    tx, ty = cx - co * cx + si * cy, cy - si * cx - co * cy
    m = np.array([[co, si], [-si, co]])                  # (row, col) order: r_in = si * c + co * r + ty, c_in = co * c - si * r + tx
    out = np.stack([ndi.affine_transform(img[..., ch], m, offset=(ty, tx), output_shape=(rows, cols), order=order,
                                         mode="constant", cval=0.0, prefilter=False) for ch in range(img.shape[2])], axis=-1)
    rr, cc = np.meshgrid(np.arange(rows, dtype=np.float64), np.arange(cols, dtype=np.float64), indexing="ij")
    r_in, c_in = si * cc + co * rr + ty, co * cc - si * rr + tx
    inside = (r_in >= 0) & (r_in <= rows - 1) & (c_in >= 0) & (c_in <= cols - 1)
    return out, inside


def _ndi_resize(ndi, img, new_rows, new_cols, order):
    """skimage.transform.resize (augmentation.py:42-57,105-133) from scipy.ndimage alone: gaussian_filter (the anti-aliasing step)
    + affine_transform for the sampling map  in = scale * (out + 0.5) - 0.5, and the interior mask as above"""
    rows, cols = img.shape[:2]
    rs, cs = float(rows) / new_rows, float(cols) / new_cols
    # the anti-aliasing filter of skimage 0.14's resize by the real scipy (one tap -- the identity -- for ratios > 0.8)
    img = ndi.gaussian_filter(img, (max(0.0, (rs - 1) / 2), max(0.0, (cs - 1) / 2), 0), cval=0, mode="constant")
    out = np.stack([ndi.affine_transform(img[..., ch], np.diag([rs, cs]), offset=(rs / 2.0 - 0.5, cs / 2.0 - 0.5),
                                         output_shape=(new_rows, new_cols), order=order, mode="constant", cval=0.0,
                                         prefilter=False) for ch in range(img.shape[2])], axis=-1)
    r_in = rs * np.arange(new_rows, dtype=np.float64) + (rs / 2.0 - 0.5)
    c_in = cs * np.arange(new_cols, dtype=np.float64) + (cs / 2.0 - 0.5)
    # scipy's 'constant' mode treats a COORDINATE outside [0, n - 1] as outside, even one that rounds to an edge pixel (order 0):
    # those are border pixels here
    ok_r, ok_c = (r_in >= 0) & (r_in <= rows - 1), (c_in >= 0) & (c_in <= cols - 1)
    return out, ok_r[:, None] & ok_c[None, :]


def test_rotation_and_resize_restatement_equals_scipy_ndimage_on_interior_pixels():
    """The pin of the sampling half of RandomRotation / RandomResize (augmentation.py:42-57,105-133,175-214): skimage 0.14 is not
    installable here, but scipy.ndimage.affine_transform(order = 0 / 1, mode='constant') is an INDEPENDENT implementation of the
    same arithmetic wherever every tap lies inside the source frame.  oracle/augment_restate.py's rotate / resize must equal it
    there: order 1 to 1e-13 (float64; scipy forms the blend from spline weights, skimage as two nested lerps), order 0 exactly.
    What stays a restatement: the border rule (skimage reads cval per out-of-image tap; scipy's 'constant' mode does not) and
    the clip to the input's value range -- a no-op for interior blends."""
    ndi = pytest.importorskip("scipy.ndimage")
    from oracle import augment_restate as ar
    rng = np.random.RandomState(11)
    checked = {"rotate": 0, "resize1": 0, "resize0": 0}
    for trial in range(16):
        h, w = rng.randint(20, 70, size=2)
        img = rng.rand(h, w, 3)
        angle = rng.uniform(-40, 40)
        want, inside = _ndi_rotate(ndi, img, angle)
        got = ar.rotate_bilinear(img, angle)
        assert inside.mean() > 0.5
        assert np.abs(got - want)[inside].max() <= 1e-13, (trial, angle)
        checked["rotate"] += int(inside.sum())
        ratio = rng.uniform(0.4, 1.3)                   # below 0.8: the multi-tap anti-aliasing filter in front of the sampling
        nh, nw = int(h * ratio), int(w * ratio)
        for order in (1, 0):
            want, inside = _ndi_resize(ndi, img, nh, nw, order)
            got = ar.resize(img, nh, nw, order)
            assert inside.mean() > 0.8
            if order == 1:
                assert np.abs(got - want)[inside].max() <= 1e-13, (trial, ratio)
            else:
                assert np.array_equal(got[inside], want[inside]), (trial, ratio)
            checked["resize%d" % order] += int(inside.sum())
    print("interior pixels pinned to scipy.ndimage:", checked)


ROT_ONLY = {"rotation_param": {"degrees": 25}}
RES_ONLY = {"nearest": {"crop_param": {"size": [64, 64]}, "resize_param": {"ratio": [0.85, 1.2]}},
            "bilinear": {"crop_param": {"size": [64, 64]}, "resize_param": {"ratio": [0.85, 1.2], "interpolation": "bilinear"}}}


@pytest.mark.parametrize("tag", ["rotation", "resize-nearest", "resize-bilinear"])
def test_rotation_and_resize_kernel_equals_scipy_ndimage_on_interior_pixels(be, shapes, tag):
    """the DEVICE kernel (mnk_frames_augment through DeviceFramesDataset) against scipy.ndimage directly -- no
    oracle/augment_restate.py in between -- on the pixels whose taps lie inside the source frame; float32 outputs of float64
    warps: two float32 ulps (a nearest-neighbour resize: exact)."""
    ndi = pytest.importorskip("scipy.ndimage")
    from mnk import frames
    fx, root, names = shapes
    params = ROT_ONLY if tag == "rotation" else RES_ONLY[tag.split("-")[1]]
    ds = frames.DeviceFramesDataset(root, params, image_shape=(64, 64, 3), is_train=True, device=be.device, files=names)
    strips = [fx["strip%d" % i] for i in range(len(names))]
    pinned = 0
    for seed in (0, 1):
        for idx in range(len(names)):
            random.seed(50 * seed + idx), np.random.seed(50 * seed + idx)
            item = ds[idx]
            random.seed(50 * seed + idx), np.random.seed(50 * seed + idx)
            sel, hflip, x1, y1, pt, pl, oh, ow, angle, new_hw, hue, jit = ds._draw(ds.meta[idx][3])
            frames_u8 = np.moveaxis(strips[idx][:, :, :3].reshape(64, -1, 64, 3), 1, 0)
            got = torch.cat([item["source"], item["video"]], dim=1).cpu().numpy()            # (C, D, h, w)
            for d, f in enumerate(sel):
                img = np.multiply(frames_u8[f], 1.0 / 255, dtype=np.float32).astype(np.float64)
                if hflip:
                    img = np.fliplr(img)
                if tag == "rotation":
                    want, inside = _ndi_rotate(ndi, img, angle)
                else:
                    want, inside = _ndi_resize(ndi, img, new_hw[0], new_hw[1], ds.resize_order)
                    h, w = ds.crop                      # pad_clip(mode='edge') + crop (augmentation.py:33-39,138-171): padded pixels
                    ih, iw = want.shape[:2]             # are copies of border pixels -- not pinned
                    ph = (0, 0) if h < ih else ((h - ih) // 2, (h - ih + 1) // 2)
                    pw = (0, 0) if w < iw else ((w - iw) // 2, (w - iw + 1) // 2)
                    want = np.pad(want, (ph, pw, (0, 0)), mode="edge")[y1:y1 + h, x1:x1 + w]
                    inside = np.pad(inside, (ph, pw), mode="constant", constant_values=False)[y1:y1 + h, x1:x1 + w]
                mine = got[:, d].transpose(1, 2, 0).astype(np.float64)
                assert mine.shape == want.shape, (mine.shape, want.shape)
                diff = np.abs(mine - want.astype(np.float32).astype(np.float64))[inside]
                assert inside.mean() > 0.4, (tag, inside.mean())
                assert diff.max() <= (0.0 if tag == "resize-nearest" else 1.2e-7), (tag, idx, float(diff.max()))
                pinned += int(inside.sum())
    print("%s: %d interior pixels of the kernel's output equal scipy.ndimage" % (tag, pinned))


def test_hue_restatement_equals_the_golden_of_the_real_pillow():
    """oracle/make_golden_hue.py ran the REAL Pillow (Image.convert RGB <-> HSV; every one of the 2^24 triples of both directions
    equals the restatement: tests/golden/HUE_PILLOW_REPORT.txt) under torchvision 0.2.1's five adjust_hue statements and
    recorded the uint8 output for 24 images x 10 hue factors: the numpy restatement that checks the device kernel must
    reproduce it exactly -- the colour-conversion half of the hue jitter is pinned to a real library, not only to its
    description."""
    from oracle import augment_restate as ar
    g = np.load(os.path.join(GOLD, "hue_pillow.npz"))
    bad = 0
    for k, f in enumerate(g["factors"]):
        for i, im in enumerate(g["images"]):
            got = np.rint(ar.adjust_hue(ar.img_as_float(im).astype(np.float32), float(f)).astype(np.float64) * 255.0).astype(np.uint8)
            bad += int((got != g["out"][k, i]).sum())
    assert bad == 0
    try:
        from PIL import Image
    except ImportError:
        return
    # live, where a Pillow is installed (whatever its version): a slice of the exhaustive comparison
    r, gg, b = np.meshgrid(np.arange(0, 256, 1, dtype=np.uint8), np.arange(0, 256, 3, dtype=np.uint8),
                           np.arange(0, 256, 5, dtype=np.uint8), indexing="ij")
    cube = np.ascontiguousarray(np.stack([r, gg, b], -1).reshape(256, -1, 3))
    assert np.array_equal(np.asarray(Image.fromarray(cube, "RGB").convert("HSV")), ar.rgb2hsv_u8(cube))
    assert np.array_equal(np.asarray(Image.fromarray(cube, "HSV").convert("RGB")), ar.hsv2rgb_u8(cube))


def test_hue_jitter_kernel_equals_the_golden_of_the_real_pillow(be, tmp_path, monkeypatch):
    """the device pipeline (mnk_frames_augment, hue only) on the golden's images with the golden's hue factors: every uint8
    level the real Pillow produced, times 1 / 255 (skimage's img_as_float, float64 -> float32)"""
    from mnk import frames
    g = np.load(os.path.join(GOLD, "hue_pillow.npz"))
    imgs, factors, out = g["images"], g["factors"], g["out"]
    os.makedirs(os.path.join(tmp_path, "train"))
    os.makedirs(os.path.join(tmp_path, "test"))
    names = []
    for v in range(len(imgs) // 2):                           # videos of two frames: strips (H, 2 W, 3)
        names.append("%03d.png" % v)
        for sub in ("train", "test"):
            _write_png(os.path.join(tmp_path, sub, names[-1]), np.concatenate([imgs[2 * v], imgs[2 * v + 1]], axis=1))
    ds = frames.DeviceFramesDataset(str(tmp_path), {"jitter_param": {"hue": 0.5}}, image_shape=(32, 32, 3), is_train=True,
                                    device=be.device, files=names)
    bad = total = 0
    for k, f in enumerate(factors):
        monkeypatch.setattr(ds, "_draw", lambda frame_count, f=float(f): ([0, 1], 0, 0, 0, 0, 0, 32, 32, None, None, f, [(3, f)]))
        b = ds.batch(list(range(len(names))))
        be.sync()
        got = torch.cat([b["source"], b["video"]], dim=2).cpu().numpy()           # (V, C, 2, H, W)
        for v in range(len(names)):
            for d in range(2):
                want = np.multiply(out[k, 2 * v + d], 1.0 / 255, dtype=np.float64).astype(np.float32)     # (H, W, 3)
                bad += int((got[v, :, d].transpose(1, 2, 0) != want).sum())
                total += want.size
    assert bad == 0, (bad, total)


def test_colour_jitter_restatement_and_kernel_equal_the_golden_of_the_real_pillow(be, tmp_path, monkeypatch):
    """all four terms of ColorJitter (augmentation.py:217-320) in shuffled sequences: oracle/make_golden_hue.py ran the REAL Pillow's
    ImageEnhance.Brightness / Color / Contrast (what torchvision 0.2.1's adjust_* call) and the hue path for 40 sequences x 8
    images; the numpy restatement and the device pipeline (mnk_frames_augment: the enhancers per pixel, the contrast term's
    frame mean by a pre-pass) must both reproduce every uint8 level"""
    from mnk import frames
    from oracle import augment_restate as ar
    g = np.load(os.path.join(GOLD, "hue_pillow.npz"))
    imgs, codes, factors, out = g["images"][:8], g["jit_codes"], g["jit_factors"], g["jit_out"]
    seqs = [[(int(c), float(f)) for c, f in zip(cs, fs) if c] for cs, fs in zip(codes, factors)]
    assert sum(int((ar.jitter_u8(im.copy(), ops) != out[k, i]).sum()) for k, ops in enumerate(seqs) for i, im in enumerate(imgs)) == 0
    for sub in ("train", "test"):
        os.makedirs(os.path.join(tmp_path, sub))
    names = []
    for v in range(len(imgs) // 2):
        names.append("%03d.png" % v)
        for sub in ("train", "test"):
            _write_png(os.path.join(tmp_path, sub, names[-1]), np.concatenate([imgs[2 * v], imgs[2 * v + 1]], axis=1))
    ds = frames.DeviceFramesDataset(str(tmp_path), {"jitter_param": {"brightness": 0.5, "contrast": 0.5, "saturation": 0.5, "hue": 0.5}},
                                    image_shape=(32, 32, 3), is_train=True, device=be.device, files=names)
    bad = total = 0
    for k, ops in enumerate(seqs):
        monkeypatch.setattr(ds, "_draw", lambda frame_count, ops=ops: ([0, 1], 0, 0, 0, 0, 0, 32, 32, None, None, None, ops))
        b = ds.batch(list(range(len(names))))
        be.sync()
        got = torch.cat([b["source"], b["video"]], dim=2).cpu().numpy()
        for v in range(len(names)):
            for d in range(2):
                want = np.multiply(out[k, 2 * v + d], 1.0 / 255, dtype=np.float64).astype(np.float32)
                bad += int((got[v, :, d].transpose(1, 2, 0) != want).sum())
                total += want.size
    assert bad == 0, (bad, total)


def test_the_draws_of_the_full_colour_jitter_follow_the_reference_statements():
    """ColorJitter.get_params draws brightness, contrast, saturation, hue in that order; __call__ appends brightness, saturation,
    hue, contrast and random.shuffle()s the list (augmentation.py:236-282)"""
    from mnk import frames
    ds = frames.DeviceFramesDataset.__new__(frames.DeviceFramesDataset)
    ds.image_shape, ds.is_train = (64, 64, 3), True
    ds.flip = ds.crop = ds.rotation = ds.resize = None
    ds.hue, ds.jitter, ds.resize_order = 0.3, {"brightness": 0.4, "contrast": 0.0, "saturation": 0.6}, 0
    for seed in range(20):
        random.seed(seed), np.random.seed(seed)
        got = ds._draw(16)
        after = (random.random(), np.random.rand())
        random.seed(seed), np.random.seed(seed)
        sel = list(np.sort(np.random.choice(range(16), replace=True, size=2)))
        fb = random.uniform(max(0, 1 - 0.4), 1 + 0.4)
        fs = random.uniform(max(0, 1 - 0.6), 1 + 0.6)
        hue = random.uniform(-0.3, 0.3)
        terms = [(frames.JIT_BRIGHTNESS, fb), (frames.JIT_SATURATION, fs), (frames.JIT_HUE, hue)]       # contrast = 0: no term
        random.shuffle(terms)
        assert (random.random(), np.random.rand()) == after
        assert got[0] == sel and got[10] == hue and got[11] == terms


def test_the_draw_order_of_the_full_augmentation_matches_the_reference_statement_order():
    """AllAugmentationTransform (augmentation.py:369-389) runs select -> flip -> rotation -> resize -> crop -> jitter; a replay of
    exactly those `random` / `numpy.random` calls must leave both generators where DeviceFramesDataset._draw leaves them"""
    from mnk import frames
    ds = frames.DeviceFramesDataset.__new__(frames.DeviceFramesDataset)
    ds.image_shape, ds.is_train = (64, 64, 3), True
    ds.flip, ds.crop = {"time_flip": True, "horizontal_flip": True}, (64, 64)
    ds.rotation, ds.resize, ds.hue, ds.resize_order, ds.jitter = (-10.0, 10.0), (0.9, 1.1), 0.5, 0, None
    for seed in range(20):
        random.seed(seed), np.random.seed(seed)
        got = ds._draw(32)
        after = (random.random(), np.random.rand())
        random.seed(seed), np.random.seed(seed)
        sel = list(np.sort(np.random.choice(range(32), replace=True, size=2)))         # SelectRandomFrames
        hflip = 0
        if random.random() < 0.5:                                                       # RandomFlip (both flags set)
            sel = sel[::-1]
        elif random.random() < 0.5:
            hflip = 1
        angle = random.uniform(-10.0, 10.0)                                             # RandomRotation
        sf = random.uniform(0.9, 1.1)                                                   # RandomResize
        nw, nh = int(64 * sf), int(64 * sf)
        im_h = nh if 64 < nh else nh + (64 - nh) // 2 + (64 - nh + 1) // 2              # pad_clip
        im_w = nw if 64 < nw else nw + (64 - nw) // 2 + (64 - nw + 1) // 2
        x1 = 0 if 64 == im_h else random.randint(0, im_w - 64)                          # RandomCrop
        y1 = 0 if 64 == im_w else random.randint(0, im_h - 64)
        hue = random.uniform(-0.5, 0.5)                                                 # ColorJitter.get_params (hue only)
        random.shuffle([None])                                                          # one transform: no draw
        assert (random.random(), np.random.rand()) == after
        assert got[0] == sel and got[1] == hflip and got[2:4] == (x1, y1) and got[8] == angle and got[9] == (nh, nw) and got[10] == hue


def test_paired_dataset_pairs_and_items(be, shapes):
    """DevicePairedDataset = frames_dataset.py:91-131: the same pairs as the reference's statements draw, items with the
    reference's keys"""
    from mnk import frames
    fx, root, names = shapes
    ds = frames.DeviceFramesDataset(root, None, image_shape=(64, 64, 3), is_train=False, device=be.device, files=names)
    pd_ = frames.DevicePairedDataset(ds, number_of_pairs=5, seed=3)
    np.random.seed(3)                                     # frames_dataset.py:100-108, restated
    xy = np.mgrid[:5, :5].reshape(2, -1).T
    want = xy.take(np.random.choice(xy.shape[0], 5, replace=False), axis=0)
    assert np.array_equal(np.asarray(pd_.pairs), want) and len(pd_) == 5
    item = pd_[1]
    assert set(item) == {"driving_video", "driving_name", "source_video", "source_name"}
    assert item["driving_name"] == names[want[1][0]] and item["source_name"] == names[want[1][1]]
    assert torch.equal(item["driving_video"].cpu(), ds[int(want[1][0])]["video"].cpu())
    # pairs_list csv (frames_dataset.py:109-121)
    csv = os.path.join(root, "pairs.csv")
    with open(csv, "w") as f:
        f.write("source,driving\n%s,%s\n%s,%s\nmissing.png,%s\n" % (names[0], names[1], names[2], names[0], names[1]))
    ds2 = frames.DeviceFramesDataset(root, None, image_shape=(64, 64, 3), is_train=False, device=be.device, files=names,
                                     pairs_list=csv)
    pd2 = frames.DevicePairedDataset(ds2, number_of_pairs=10)
    assert pd2.pairs == [(1, 0), (0, 2)]                  # (driving, source); the row with a missing file is dropped


def _history_check(step, x, history, history64, be):
    report = []
    for it, (ref, ref64) in enumerate(zip(history, history64)):
        g_losses, d_losses, _ = step.step(x)
        be.sync()
        mine = [float(v) for v in g_losses] + [float(v) for v in d_losses]
        r32 = ref["generator"] + ref["discriminator"]
        r64 = ref64["generator"] + ref64["discriminator"]
        spread = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(r32, r64))
        err = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(mine, r64))
        report.append((it, err, spread))
        # the yard-stick of tests/test_step.py: the reference's own fp32-vs-fp64 separation at the same iteration
        assert err <= 16.0 * spread + 2e-5, "iteration %d: |hip - ref64| = %.3e vs reference fp32 noise %.3e" % (it, err, spread)
    return report


def _real_batch(be, shapes, seed=0):
    from mnk import frames
    fx, root, names = shapes
    params, _ = VARIANTS["cfg"]
    order = [names[i] for i in fx["order_cfg"]]
    ds = frames.DeviceFramesDataset(root, params, image_shape=(64, 64, 3), is_train=True, device=be.device, files=order)
    random.seed(seed), np.random.seed(seed)
    return ds.batch(range(8))


@pytest.mark.gpu
def test_three_training_steps_of_shapes_yaml_on_real_frames():
    """BASELINE configs[0]: config/shapes.yaml on frames of data/shapes/train -- the frames come out of the device-side input
    path, the three iterations run through mnk.engine.TrainStep, the yard-stick is the reference's own loss history."""
    from conftest import Backend
    from mnk import engine
    from oracle import cases
    be = Backend("hip")
    gold = load("step_shapes_frames")
    fx = np.load(os.path.join(GOLD, "frames_shapes.npz"))
    cfg = gold["cfg"]
    gen, disc, kpd = build(cfg)
    for i, m in enumerate((gen, disc, kpd)):          # oracle/make_golden.py::build_reference
        sd = m.state_dict()
        cases.perturb_state_dict(sd, 7 + i)
        m.load_state_dict(sd)
    gen.to(be.device), disc.to(be.device), kpd.to(be.device)
    step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=False)
    import tempfile
    with tempfile.TemporaryDirectory() as root:
        names = [str(n) for n in fx["names"]]
        for sub in ("train", "test"):
            os.makedirs(os.path.join(root, sub))
            for i, n in enumerate(names):
                _write_png(os.path.join(root, sub, n), fx["strip%d" % i])
        x = _real_batch(be, (fx, root, names))
    assert torch.equal(x["source"][3].cpu(), torch.from_numpy(fx["cfg_s0_i3_source"]))
    report = _history_check(step, {"source": x["source"], "video": x["video"]}, gold["history"], gold["history64"], be)
    print("shapes.yaml on real frames (iteration, |hip-ref64|, |ref32-ref64|):", report)


def test_reference_checkpoint_layout_restores_models_and_optimisers(be, shapes):
    """A file in the layout of Logger.save_cpk (logger.py:43-47), written by the REFERENCE's models and torch.optim.Adam
    after three iterations on real frames, restored the way Logger.load_cpk does (logger.py:49-66) into the drop-in modules
    and into both optimiser pipelines: the evaluation forward of the restored networks equals the reference's, and the
    iteration after the restore starts from the reference's losses."""
    from mnk import engine
    gold = load("step_shapes_frames")
    cfg, cpk = gold["tiny_cfg"], gold["tiny_checkpoint"]
    assert set(cpk) == {"generator", "discriminator", "kp_detector", "optimizer_generator", "optimizer_discriminator",
                        "optimizer_kp_detector", "epoch", "it"}
    x = _real_batch(be, shapes)
    for mnk_adam in (False, True):
        gen, disc, kpd = build(cfg)
        gen.to(be.device), disc.to(be.device), kpd.to(be.device)
        step = engine.TrainStep(gen, disc, kpd, cfg["train_params"], fused_adam=mnk_adam)
        # Logger.load_cpk
        gen.load_state_dict(cpk["generator"])
        kpd.load_state_dict(cpk["kp_detector"])
        disc.load_state_dict(cpk["discriminator"])
        step.opt_g.load_state_dict(copy.deepcopy(cpk["optimizer_generator"]))
        step.opt_d.load_state_dict(copy.deepcopy(cpk["optimizer_discriminator"]))
        step.opt_k.load_state_dict(copy.deepcopy(cpk["optimizer_kp_detector"]))
        assert (cpk["epoch"], cpk["it"]) == (0, 3)
        # the optimiser state is the reference's: exp_avg / exp_avg_sq / step of every parameter that had received a gradient
        sd = step.opt_g.state_dict()
        ref_state = cpk["optimizer_generator"]["state"]
        for k, st in ref_state.items():
            assert float((sd["state"][k]["exp_avg"].cpu() - st["exp_avg"]).abs().max()) == 0.0
            assert float((sd["state"][k]["exp_avg_sq"].cpu() - st["exp_avg_sq"]).abs().max()) == 0.0
            assert float(sd["state"][k]["step"]) == float(st["step"]) == 3.0
        if not mnk_adam:
            continue          # (the networks are restored by the same calls in both cases: their forward and the iteration after the
                              # restore run once, on the default pipeline -- the expensive part on the emulator)
        gen.eval(), kpd.eval()
        with torch.no_grad():
            kp_s, kp_d = kpd(x["source"]), kpd(x["video"])
            pred = gen(x["source"], kp_driving=kp_d, kp_source=kp_s)["video_prediction"]
        be.sync()
        assert float((kp_d["mean"].cpu() - gold["tiny_eval_kp_mean_after"]).abs().max()) < 2e-5
        assert float((pred.cpu() - gold["tiny_eval_prediction_after"]).abs().max()) < 2e-4
        gen.train(), kpd.train()
        g_losses, _, _ = step.step({"source": x["source"], "video": x["video"]})
        be.sync()
        for a, b in zip(g_losses, gold["tiny_next_generator_losses"]):
            assert abs(float(a) - b) <= 2e-3 * max(1.0, abs(b)), (float(a), b)
