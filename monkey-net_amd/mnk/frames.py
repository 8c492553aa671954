"""Device-side input path: `FramesDataset` / the training `DataLoader` of the reference
(frames_dataset.py:43-88, train.py:99) with the dataset resident in HBM.

The reference decodes a stacked-frame PNG per sample on the host, augments it with numpy / skimage inside 4 DataLoader
workers and copies fp32 batches over PCIe (at ~3 k frames/s per GPU, ~24 k/s on 8, that pipeline becomes the bottleneck of
real-data training).  Here every strip is decoded ONCE, stays in device memory as uint8 (a 64x64 data set of 10^5 videos x 32
frames is 39 GB of the 288 GB), and a batch is ONE kernel launch (mnk_frames_gather) driven by a small job table: the random
choices of the augmentation are drawn on the host in the reference's own order from the same `random` / `numpy.random`
generators, so with equal seeds a sample is bit-identical to `FramesDataset.__getitem__`.

Supported, integer-exact and parity-tested against the unmodified reference transforms: frame selection, time flip,
horizontal flip, edge padding + random crop, gray / RGBA handling, uint8 -> float32, (C, D, H, W) layout.  Round 4: the
non-integer augmentations of config/moving-gif.yaml and actions.yaml -- `rotation_param` (skimage.transform.rotate),
`resize_param` (skimage.transform.resize, order 0 / 1; round 5: with the multi-tap anti-aliasing filter of ratios < 0.8 -- RandomResize's
DEFAULT ratio (3/4, 4/3) included -- down to ratio 0.32; the filter is scipy's gaussian_filter, pinned to the installed scipy) and `jitter_param` with `hue` (img_as_ubyte -> PIL HSV ->
torchvision adjust_hue -> img_as_float) -- in one launch per batch (mnk_frames_augment), in the arithmetic of the package
versions the reference pins; scikit-image and torchvision are not in this image (and not installable offline), so rotation and
resize are checked against a numpy restatement of skimage 0.14's published algorithm (oracle/augment_restate.py) AND, since
round 6, directly against scipy.ndimage.affine_transform / gaussian_filter on every pixel whose taps lie inside the source frame
(an independent third-party implementation of the same interior arithmetic; the border rule stays a restatement); the hue
jitter's colour conversions are pinned to the INSTALLED Pillow (12.2; the reference pins 5.2.0) since round 5: the restatement equals
its Image.convert RGB <-> HSV on all 2^24 triples of both directions, and kernel and restatement reproduce a
golden made by that Pillow (oracle/make_golden_hue.py, tests/golden/hue_pillow.npz).  Round 5 also: brightness / contrast /
saturation -- all four terms of ColorJitter in the reference's shuffled order (Pillow's ImageEnhance arithmetic per pixel, the
contrast term's frame mean by a pre-pass), pinned the same way (40 shuffled sequences x 8 images from the real library).  `.gif` files (the moving-gif data set) are decoded with
Pillow (read_gif); `.mp4` / `.mov` need a decoder this image does not have; PNG strips are read by the small decoder below
(zlib + the five PNG filters), or by PIL when it is importable.
`DevicePairedDataset` is frames_dataset.py:91-131's PairedDataset over a DeviceFramesDataset."""
import math
import os
import random
import struct
import zlib

import numpy as np
import torch

from . import ops as mops

JOB = np.dtype([("strip_offset", "<u8"), ("out_offset", "<u8"), ("chan_stride", "<u8"), ("strip_w", "<i4"), ("in_h", "<i4"),
                ("in_w", "<i4"), ("channels", "<i4"), ("frame", "<i4"), ("hflip", "<i4"), ("x1", "<i4"), ("y1", "<i4"),
                ("pad_top", "<i4"), ("pad_left", "<i4"), ("reserved0", "<i4"), ("reserved1", "<i4")])


AUGJOB = np.dtype([("strip_offset", "<u8"), ("out_offset", "<u8"), ("chan_stride", "<u8"), ("rot", "<f8", (6,)),
                   ("strip_w", "<i4"), ("in_h", "<i4"), ("in_w", "<i4"), ("channels", "<i4"), ("frame", "<i4"), ("hflip", "<i4"),
                   ("x1", "<i4"), ("y1", "<i4"), ("pad_top", "<i4"), ("pad_left", "<i4"), ("new_h", "<i4"), ("new_w", "<i4"),
                   ("flags", "<i4"), ("hue_shift", "<i4"), ("vmin", "<f4"), ("vmax", "<f4"), ("jit_n", "<i4"),
                   ("jit_op", "<i4", (4,)), ("jit_f", "<f4", (4,)), ("reserved", "<i4"), ("aa_rr", "<i4"), ("aa_rc", "<i4"),
                   ("aa_wr", "<f8", (5,)), ("aa_wc", "<f8", (5,))])
JIT_BRIGHTNESS, JIT_SATURATION, JIT_HUE, JIT_CONTRAST = 1, 2, 3, 4      # MnkAugJob.jit_op (the order ColorJitter appends them in)


# ---- PNG (8 bit, non-interlaced; gray, gray + alpha, RGB, RGBA): what `skimage.io.imread` returns for those files -------
def decode_png(path):
    """-> uint8 array (H, W) or (H, W, C)."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, head = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            head = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = head
    chans = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if depth != 8 or chans is None or interlace:
        raise NotImplementedError("%s: only 8-bit non-interlaced gray / RGB (+ alpha) PNGs (got depth %d, colour type %d, "
                                  "interlace %d)" % (path, depth, ctype, interlace))
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * chans)
    out = np.zeros((h, w * chans), dtype=np.uint8)
    prev = np.zeros(w * chans, dtype=np.int32)
    for r in range(h):
        f, line = int(raw[r, 0]), raw[r, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:                                   # Up
            cur = (line + prev) & 255
        elif f == 1:                                   # Sub: a running sum per byte lane
            cur = line.reshape(w, chans).cumsum(axis=0).reshape(-1) & 255
        else:                                          # Average / Paeth depend on the reconstructed left byte: sequential
            cur = np.zeros(w * chans, dtype=np.int32)
            for i in range(w * chans):
                a = cur[i - chans] if i >= chans else 0
                b = prev[i]
                if f == 3:
                    cur[i] = (line[i] + ((a + b) >> 1)) & 255
                elif f == 4:
                    c = prev[i - chans] if i >= chans else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    cur[i] = (line[i] + (a if pa <= pb and pa <= pc else (b if pb <= pc else c))) & 255
                else:
                    raise ValueError("%s: bad PNG filter %d" % (path, f))
        out[r] = cur
        prev = cur
    return out.reshape(h, w) if chans == 1 else out.reshape(h, w, chans)


def read_gif(path):
    """frames_dataset.py:30-36 for a .gif: `np.array(imageio.mimread(name))` -- every frame composited on the canvas (disposal
    methods, transparency) and converted from its palette to RGB (RGBA when the file declares a transparent index; the
    reference then drops the alpha channel, :34-35; gray frames become three equal channels, :32-33) -- laid out as the strip
    of frames (H, W * F, channels) the rest of this module works on.  Decoded with Pillow, the library imageio's GIF reader
    itself sits on (imageio 2.3.0 `GIF-PIL`); imageio is not part of this image, so this branch is checked on GIFs written
    from known frames (tests/test_frames.py), not against imageio: "parity unpinned" for files whose frames depend on how a
    decoder composites partial frames."""
    try:
        from PIL import Image, ImageSequence
    except ImportError:
        raise NotImplementedError("%s: reading .gif files needs Pillow" % path)
    frames = []
    with Image.open(path) as im:
        alpha = "transparency" in im.info
        for fr in ImageSequence.Iterator(im):
            a = np.array(fr.convert("RGBA" if alpha else "RGB"))
            frames.append(a[:, :, :3])
    if not frames:
        raise ValueError("%s: no frames" % path)
    if any(f.shape != frames[0].shape for f in frames):
        raise ValueError("%s: frames of different sizes" % path)
    return np.concatenate(frames, axis=1)


def read_strip(path):
    """the decoded frames of a file as uint8 (H, W * F, channels): a stacked-frame .png (frames_dataset.py:15-29) as it is, the
    frames of a .gif side by side"""
    if path.lower().endswith(".gif"):
        return read_gif(path)
    if path.lower().endswith(".jpg"):
        try:
            from PIL import Image
        except ImportError:
            raise NotImplementedError("%s: reading .jpg files needs Pillow" % path)
        with Image.open(path) as im:          # (what skimage.io.imread's default plugin does for a .jpg: PIL's decoder)
            arr = np.array(im if im.mode in ("L", "RGB") else im.convert("RGB"))
        return arr[:, :, None] if arr.ndim == 2 else arr
    if not path.lower().endswith(".png"):
        raise NotImplementedError("%s: stacked-frame .png / .jpg and .gif files only (frames_dataset.py:15-36); .mp4 / .mov need a "
                                  "decoder that is not part of this image" % path)
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode not in ("L", "LA", "RGB", "RGBA"):
                im = im.convert("RGBA" if "A" in im.mode or "transparency" in im.info else "RGB")
            arr = np.array(im)
    except ImportError:
        arr = decode_png(path)
    return arr[:, :, None] if arr.ndim == 2 else arr


def split_train_test(images, random_seed=0, test_size=0.2):
    """scikit-learn 0.19.2's train_test_split(images, random_state=random_seed, test_size=test_size) for a list"""
    n = len(images)
    n_test = int(math.ceil(test_size * n))
    perm = np.random.RandomState(random_seed).permutation(n)
    return [images[i] for i in perm[n_test:]], [images[i] for i in perm[:n_test]]


class DeviceFramesDataset:
    """`FramesDataset(root_dir, augmentation_params, image_shape, is_train, random_seed, pairs_list)` with the decoded
    strips resident on `device`.  `dataset[i]` returns what the reference's returns (device tensors instead of numpy
    arrays); `batch(indices)` makes a whole batch with one launch."""

    def __init__(self, root_dir, augmentation_params=None, image_shape=(64, 64, 3), is_train=True, random_seed=0,
                 pairs_list=None, device=None, files=None):
        self.image_shape = tuple(image_shape)
        self.pairs_list = pairs_list
        self.is_train = bool(is_train)
        if os.path.exists(os.path.join(root_dir, "train")):       # frames_dataset.py:55-60: predefined train-test split
            assert os.path.exists(os.path.join(root_dir, "test"))
            root_dir = os.path.join(root_dir, "train" if is_train else "test")
        elif files is None:
            # frames_dataset.py:61-63: sklearn.model_selection.train_test_split(images, random_state=random_seed, test_size=0.2),
            # restated: ceil(0.2 n) test items from the front of RandomState(seed).permutation(n), the rest train
            # (ShuffleSplit._iter_indices; tests/test_frames.py compares with scikit-learn itself where it is installed)
            train, test = split_train_test(os.listdir(root_dir), random_seed)
            files = train if is_train else test
        if files is None:
            files = os.listdir(root_dir)                          # the reference's order: the directory listing, not sorted
        self.root_dir = root_dir
        self.images = list(files)
        p = dict(augmentation_params or {})
        self.flip = dict(p["flip_param"]) if is_train and p.get("flip_param") is not None else None
        crop = p.get("crop_param") if is_train else None
        if crop is not None:
            size = crop["size"]
            self.crop = (int(size), int(size)) if isinstance(size, (int, float)) else (int(size[0]), int(size[1]))
        else:
            self.crop = None
        # the non-integer augmentations (augmentation.py:105-133,175-214,217-320)
        self.rotation = self.resize = self.hue = self.jitter = None
        self.resize_order = 0
        if is_train and p.get("rotation_param") is not None:
            deg = p["rotation_param"]["degrees"]
            if isinstance(deg, (int, float)):
                if deg < 0:
                    raise ValueError("If degrees is a single number,must be positive")      # augmentation.py:186-188
                deg = (-deg, deg)
            if len(deg) != 2:
                raise ValueError("If degrees is a sequence,it must be of len 2.")
            self.rotation = (float(deg[0]), float(deg[1]))
        if is_train and p.get("resize_param") is not None:
            rp = dict(p["resize_param"])
            ratio = tuple(rp.get("ratio", (3. / 4., 4. / 3.)))
            # resize_clip (augmentation.py:55) runs skimage's resize with order=1 ONLY for interpolation == 'bilinear'; the
            # default 'nearest' -- every shipped config -- is order 0
            self.resize_order = 1 if rp.get("interpolation", "nearest") == "bilinear" else 0
            # (round 5) ratios below 0.8 run skimage's multi-tap anti-aliasing filter on the device (radius <= 4: ratio >= ~0.31)
            if min(ratio) < 0.32:
                raise NotImplementedError("resize ratios below 0.32 need an anti-aliasing kernel wider than 9 taps (ratio %s)" % (ratio,))
            # the per-axis scale is size / int(size * f), not 1 / f: for small frames it can exceed 1 / 0.32 although f >= 0.32
            # (H = 30, f = 0.32: int(9.6) = 9 rows, scale 3.33, radius 5).  Refuse here, with the launch's own arithmetic, not
            # from inside a training batch
            for size in self.image_shape[:2]:
                new = int(size * min(ratio))
                worst = int(4.0 * max(0.0, (float(size) / max(new, 1) - 1.0) / 2.0) + 0.5)
                if new < 1 or worst > 4:
                    raise NotImplementedError("resize ratio %s of a %d-pixel axis gives %d pixels: an anti-aliasing kernel of radius %d "
                                              "(the device kernel holds 4)" % (ratio, size, new, worst))
            self.resize = (float(ratio[0]), float(ratio[1]))
        if is_train and p.get("jitter_param") is not None:
            jp = dict(p["jitter_param"])
            if jp.get("hue", 0) > 0:
                self.hue = float(jp["hue"])
            # (round 5) the other three terms: ColorJitter(brightness, contrast, saturation, hue), augmentation.py:217-235
            self.jitter = {k: float(jp.get(k, 0)) for k in ("brightness", "contrast", "saturation")}
            if not any(v > 0 for v in self.jitter.values()):
                self.jitter = None
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # ---- decode once, keep resident -----------------------------------------------------------------------------
        H, W, C = self.image_shape
        strips, self.meta, off = [], [], 0
        self.ranges = []
        for name in self.images:
            arr = np.ascontiguousarray(read_strip(os.path.join(root_dir, name)))
            h, wf, ch = arr.shape
            if h != H or wf % W != 0:
                raise ValueError("%s: a %dx%d strip is not a row of %dx%d frames" % (name, h, wf, H, W))
            self.meta.append((off, wf, ch, wf // W))
            # value range of every frame (over the colour channels the pipeline keeps): skimage clips a warp's output to it
            fr = arr[:, :, :3] if ch >= 3 else arr[:, :, :1]
            fr = fr.reshape(h, wf // W, W, -1)
            self.ranges.append((fr.min(axis=(0, 2, 3)), fr.max(axis=(0, 2, 3))))
            strips.append(arr.reshape(-1))
            off += (arr.size + 15) // 16 * 16
        pool = np.zeros(max(off, 16), dtype=np.uint8)
        for (o, _, _, _), flat in zip(self.meta, strips):
            pool[o:o + flat.size] = flat
        self.pool = torch.from_numpy(pool).to(self.device)
        self._tables = []

    def __len__(self):
        return len(self.images)

    # ---- the random choices of one sample, in the reference's draw order ------------------------------------------------
    def _draw(self, frame_count):
        """-> (frames [source, driving...], hflip, x1, y1, pad_top, pad_left, out_h, out_w, angle, new_hw, hue_factor, jitter terms
        [(code, factor), ...] in applied order or None): the
        random choices of AllAugmentationTransform (augmentation.py:369-389) in ITS draw order -- select, flip, rotation,
        resize, crop, jitter"""
        H, W, _ = self.image_shape
        if not self.is_train:                                   # VideoToTensor: every frame, no augmentation
            return list(range(frame_count)), 0, 0, 0, 0, 0, H, W, None, None, None, None
        # SelectRandomFrames (augmentation.py:324-345): two indices with replacement, sorted
        sel = list(np.sort(np.random.choice(range(frame_count), replace=True, size=2)))
        hflip = 0
        if self.flip is not None:                               # RandomFlip (:91-104): a time flip returns BEFORE the second draw
            if random.random() < 0.5 and self.flip.get("time_flip", False):
                sel = sel[::-1]
            elif random.random() < 0.5 and self.flip.get("horizontal_flip", False):
                hflip = 1
        angle = random.uniform(self.rotation[0], self.rotation[1]) if self.rotation is not None else None     # :206
        new_hw = None
        rh, rw = H, W                                           # the frame size the crop sees
        if self.resize is not None:                             # RandomResize (:120-133)
            scaling_factor = random.uniform(self.resize[0], self.resize[1])
            rw, rh = int(W * scaling_factor), int(H * scaling_factor)
            new_hw = (rh, rw)
        x1 = y1 = pt = pl = 0
        oh, ow = rh, rw
        if self.crop is not None:                               # RandomCrop (:135-171) incl. its pad_clip and its quirks
            oh, ow = self.crop
            pt = 0 if oh < rh else (oh - rh) // 2
            pl = 0 if ow < rw else (ow - rw) // 2
            im_h = rh if oh < rh else rh + (oh - rh) // 2 + (oh - rh + 1) // 2
            im_w = rw if ow < rw else rw + (ow - rw) // 2 + (ow - rw + 1) // 2
            x1 = 0 if oh == im_h else random.randint(0, im_w - ow)      # (sic: the height decides whether x is drawn)
            y1 = 0 if ow == im_w else random.randint(0, im_h - oh)
        # ColorJitter.get_params (:238-262): brightness, contrast, saturation, hue are drawn in THAT order; __call__ (:271-282) appends
        # the terms as brightness, saturation, hue, contrast and random.shuffle()s the list (a list of one element draws nothing)
        jit = None
        if self.jitter is None:
            hue = random.uniform(-self.hue, self.hue) if self.hue is not None else None
            if hue is not None:
                jit = [(JIT_HUE, hue)]
        else:
            jb, jc, js = self.jitter["brightness"], self.jitter["contrast"], self.jitter["saturation"]
            fb = random.uniform(max(0, 1 - jb), 1 + jb) if jb > 0 else None
            fc = random.uniform(max(0, 1 - jc), 1 + jc) if jc > 0 else None
            fs = random.uniform(max(0, 1 - js), 1 + js) if js > 0 else None
            hue = random.uniform(-self.hue, self.hue) if self.hue is not None else None
            jit = [t for t in ((JIT_BRIGHTNESS, fb), (JIT_SATURATION, fs), (JIT_HUE, hue), (JIT_CONTRAST, fc)) if t[1] is not None]
            random.shuffle(jit)
        return sel, hflip, x1, y1, pt, pl, oh, ow, angle, new_hw, hue, jit

    def _jobs(self, indices):
        """draw every sample of a batch -> (job rows, per-tensor frame counts, output size)"""
        drawn = [(i,) + tuple(self._draw(self.meta[i][3])) for i in indices]
        sizes = {(d[7], d[8]) for d in drawn}
        assert len(sizes) == 1, "samples of one batch must have one output size (a resize needs a crop_param behind it)"
        return drawn, sizes.pop()

    def _launch(self, rows, total_floats, oh, ow):
        """rows: (frames_gather fields ..., video index, angle, new_hw, jitter terms) per output frame"""
        C = self.image_shape[2]
        out = torch.empty(total_floats, dtype=torch.float32, device=self.device)
        augment = self.rotation is not None or self.resize is not None or self.hue is not None or self.jitter is not None
        if not augment:
            rec = np.zeros(len(rows), dtype=JOB)
            for k, r in enumerate(rows):
                rec[k] = r[:15]
            table = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self.device, non_blocking=True)
            for k0 in range(0, len(rows), 65535):
                n = min(65535, len(rows) - k0)
                mops._call("mnk_frames_gather", out, mops._p(self.pool), table.data_ptr() + k0 * JOB.itemsize, n, oh, ow, C,
                           mops._p(out))
            return out
        rec = np.zeros(len(rows), dtype=AUGJOB)
        k32 = np.float32(1.0 / 255)
        any_contrast = any_aa = False
        for k, r in enumerate(rows):
            (off, out_off, cstride, wf, H, W, ch, frame, hflip, x1, y1, pt, pl, _, _, vid, angle, new_hw, jit) = r
            j = rec[k]
            j["strip_offset"], j["out_offset"], j["chan_stride"] = off, out_off, cstride
            j["strip_w"], j["in_h"], j["in_w"], j["channels"] = wf, H, W, ch
            j["frame"], j["hflip"], j["x1"], j["y1"], j["pad_top"], j["pad_left"] = frame, hflip, x1, y1, pt, pl
            flags = 0
            if angle is not None:                     # skimage.transform.rotate: T(centre) R(angle) T(-centre) as the inverse map
                flags |= 1
                cx, cy = W / 2.0 - 0.5, H / 2.0 - 0.5
                a = math.radians(angle)
                co, si = math.cos(a), math.sin(a)
                j["rot"] = (co, -si, cx - co * cx + si * cy, si, co, cy - si * cx - co * cy)
            if new_hw is not None:
                flags |= 2 if self.resize_order == 1 else 8
                j["new_h"], j["new_w"] = new_hw
                # skimage's resize: ndi.gaussian_filter(sigma = max(0, (in / out - 1) / 2)) in front of the sampling; scipy's kernel:
                # radius int(4 sigma + 0.5), exp(-0.5 / sigma^2 * x^2) normalised -- made here as scipy makes it, bit for bit
                for axis, scale in (("r", float(H) / new_hw[0]), ("c", float(W) / new_hw[1])):
                    sigma = max(0.0, (scale - 1.0) / 2.0)
                    radius = int(4.0 * sigma + 0.5) if sigma > 1e-15 else 0
                    if radius > 4:
                        raise NotImplementedError("anti-aliasing kernel of radius %d (a resize to %s)" % (radius, new_hw))
                    w = np.ones(1)
                    if radius > 0:
                        xk = np.arange(-radius, radius + 1)
                        w = np.exp(-0.5 / (sigma * sigma) * xk ** 2)
                        w = w / w.sum()
                        flags |= 16
                        any_aa = True
                    j["aa_r" + axis] = radius
                    j["aa_w" + axis][:radius + 1] = w[:radius + 1]
            else:
                j["new_h"], j["new_w"] = H, W
            if jit:
                flags |= 4
                j["jit_n"] = len(jit)
                for t, (code, f) in enumerate(jit):
                    j["jit_op"][t] = code
                    j["jit_f"][t] = np.float32(f)              # Image.blend takes its factor as a C float
                    if code == JIT_HUE:
                        j["hue_shift"] = int(math.trunc(f * 255)) % 256      # np.uint8(hue_factor * 255): truncation, wrap-around
                any_contrast = any_contrast or any(code == JIT_CONTRAST for code, _ in jit)
            j["flags"] = flags
            lo, hi = self.ranges[vid]
            j["vmin"], j["vmax"] = np.float32(lo[frame]) * k32, np.float32(hi[frame]) * k32
        table = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self.device, non_blocking=True)
        any_rot = int(self.rotation is not None or any_aa)          # the range pre-pass of the frame the resize samples
        for k0 in range(0, len(rows), 65535):
            n = min(65535, len(rows) - k0)
            rng = torch.empty(2 * n, dtype=torch.float64, device=self.device) if any_rot else None
            cmean = torch.empty(n, dtype=torch.int32, device=self.device) if any_contrast else None
            mops._call("mnk_frames_augment", out, mops._p(self.pool), table.data_ptr() + k0 * AUGJOB.itemsize, n, any_rot,
                       mops._p(rng), int(any_contrast), mops._p(cmean), oh, ow, C, mops._p(out))
        return out

    def batch(self, indices):
        """{'source': (B,C,1,H,W), 'video': (B,C,D-1,H,W), 'name': [...]} in training mode (SplitSourceDriving), {'video':
        (B,C,F,H,W), 'name'} otherwise (VideoToTensor; the videos of a batch must have equal frame counts): ONE launch."""
        indices = [int(i) for i in indices]
        drawn, (oh, ow) = self._jobs(indices)
        H, W, C = self.image_shape
        B = len(indices)
        nf = {len(d[1]) for d in drawn}
        assert len(nf) == 1, "videos of one batch must have the same number of frames"
        nf = nf.pop()
        plane = oh * ow
        rows = []
        if self.is_train:
            d_drv = nf - 1
            src_floats = B * C * plane
            for b, (i, sel, hflip, x1, y1, pt, pl, _, _, angle, new_hw, _hue, jit) in enumerate(drawn):
                off, wf, ch, _ = self.meta[i]
                rows.append((off, b * C * plane, plane, wf, H, W, ch, int(sel[0]), hflip, x1, y1, pt, pl, 0, 0, i, angle, new_hw,
                             jit))
                for d, f in enumerate(sel[1:]):
                    rows.append((off, src_floats + (b * C * d_drv + d) * plane, d_drv * plane, wf, H, W, ch, int(f), hflip, x1,
                                 y1, pt, pl, 0, 0, i, angle, new_hw, jit))
            out = self._launch(rows, src_floats + B * C * d_drv * plane, oh, ow)
            return {"source": out[:src_floats].view(B, C, 1, oh, ow), "video": out[src_floats:].view(B, C, d_drv, oh, ow),
                    "name": [self.images[i] for i in indices]}
        for b, (i, sel, hflip, x1, y1, pt, pl, _, _, angle, new_hw, _hue, jit) in enumerate(drawn):
            off, wf, ch, _ = self.meta[i]
            for d, f in enumerate(sel):
                rows.append((off, (b * C * nf + d) * plane, nf * plane, wf, H, W, ch, int(f), hflip, x1, y1, pt, pl, 0, 0, i,
                             angle, new_hw, jit))
        out = self._launch(rows, B * C * nf * plane, oh, ow)
        return {"video": out.view(B, C, nf, oh, ow), "name": [self.images[i] for i in indices]}

    def __getitem__(self, idx):
        b = self.batch([idx])
        out = {k: v[0] for k, v in b.items() if k != "name"}
        out["name"] = b["name"][0]
        return out


class DeviceLoader:
    """`DataLoader(dataset, batch_size, shuffle=True, drop_last=True)` of train.py:99 over a DeviceFramesDataset: one
    permutation per epoch (torch's generator, as RandomSampler draws it), one launch per batch, nothing crosses PCIe but
    the job table (60 bytes per frame)."""

    def __init__(self, dataset, batch_size, shuffle=True, drop_last=True, generator=None):
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.generator = generator

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for k in range(len(self)):
            yield self.dataset.batch(order[k * self.batch_size:(k + 1) * self.batch_size])



class DevicePairedDataset:
    """PairedDataset (frames_dataset.py:91-131): the (driving, source) video pairs that transfer.py iterates, over a
    DeviceFramesDataset.  Same constructor, same random choice of the pairs (numpy's global generator seeded with `seed`, the
    mgrid / choice statements of :103-108) or the pairs of the data set's `pairs_list` csv (:109-121); an item is the reference's
    dict {'driving_video', 'driving_name', 'source_video', 'source_name'} with device tensors."""

    def __init__(self, initial_dataset, number_of_pairs, seed=0):
        self.initial_dataset = initial_dataset
        pairs_list = self.initial_dataset.pairs_list
        np.random.seed(seed)
        if pairs_list is None:
            max_idx = min(number_of_pairs, len(initial_dataset))
            nx, ny = max_idx, max_idx
            xy = np.mgrid[:nx, :ny].reshape(2, -1).T
            number_of_pairs = min(xy.shape[0], number_of_pairs)
            self.pairs = xy.take(np.random.choice(xy.shape[0], number_of_pairs, replace=False), axis=0)
        else:
            import pandas as pd
            images = self.initial_dataset.images
            name_to_index = {name: index for index, name in enumerate(images)}
            pairs = pd.read_csv(pairs_list)
            pairs = pairs[np.logical_and(pairs['source'].isin(images), pairs['driving'].isin(images))]
            number_of_pairs = min(pairs.shape[0], number_of_pairs)
            self.pairs = []
            self.start_frames = []
            for ind in range(number_of_pairs):
                self.pairs.append((name_to_index[pairs['driving'].iloc[ind]], name_to_index[pairs['source'].iloc[ind]]))

    def __len__(self):
        return len(self.pairs)

    def __getitem__(self, idx):
        pair = self.pairs[idx]
        first = self.initial_dataset[int(pair[0])]
        second = self.initial_dataset[int(pair[1])]
        first = {'driving_' + key: value for key, value in first.items()}
        second = {'source_' + key: value for key, value in second.items()}
        return {**first, **second}
