"""Device-side input path: `FramesDataset` / the training `DataLoader` of the reference
(frames_dataset.py:43-88, train.py:99) with the dataset resident in HBM.

The reference decodes a stacked-frame PNG per sample on the host, augments it with numpy / skimage inside 4 DataLoader
workers and copies fp32 batches over PCIe (at ~3 k frames/s per GPU, ~24 k/s on 8, that pipeline becomes the bottleneck of
real-data training).  Here every strip is decoded ONCE, stays in device memory as uint8 (a 64x64 data set of 10^5 videos x 32
frames is 39 GB of the 288 GB), and a batch is ONE kernel launch (mnk_frames_gather) driven by a small job table: the random
choices of the augmentation are drawn on the host in the reference's own order from the same `random` / `numpy.random`
generators, so with equal seeds a sample is bit-identical to `FramesDataset.__getitem__`.

Supported (integer-exact, parity-tested against the unmodified reference transforms): frame selection, time flip,
horizontal flip, edge padding + random crop, gray / RGBA handling, uint8 -> float32, (C, D, H, W) layout.  `resize_param`,
`rotation_param` and `jitter_param` (skimage / PIL arithmetic: config/actions.yaml, moving-gif.yaml) are NOT implemented and
raise -- there is no host fallback.  `.gif` / `.mp4` inputs need a decoder this image does not have; PNG strips are read by
the small decoder below (zlib + the five PNG filters), or by PIL when it is importable."""
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


def read_strip(path):
    """the decoded image of a stacked-frame file as uint8 (H, W * F, channels)"""
    if not path.lower().endswith(".png"):
        raise NotImplementedError("%s: only stacked-frame .png files (frames_dataset.py:15-29); .jpg / .gif / .mp4 need a "
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
            raise NotImplementedError("random train-test split (sklearn's train_test_split, frames_dataset.py:63) -- give a "
                                      "directory with train/ and test/, or pass `files`")
        if files is None:
            files = os.listdir(root_dir)                          # the reference's order: the directory listing, not sorted
        self.root_dir = root_dir
        self.images = list(files)
        p = dict(augmentation_params or {})
        for k in ("resize_param", "rotation_param", "jitter_param"):
            if is_train and p.get(k) is not None:
                raise NotImplementedError("augmentation %s is skimage / PIL arithmetic on the host in the reference and is not "
                                          "part of the device-side input path" % k)
        self.flip = dict(p["flip_param"]) if is_train and p.get("flip_param") is not None else None
        crop = p.get("crop_param") if is_train else None
        if crop is not None:
            size = crop["size"]
            self.crop = (int(size), int(size)) if isinstance(size, (int, float)) else (int(size[0]), int(size[1]))
        else:
            self.crop = None
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # ---- decode once, keep resident -----------------------------------------------------------------------------
        H, W, C = self.image_shape
        strips, self.meta, off = [], [], 0
        for name in self.images:
            arr = np.ascontiguousarray(read_strip(os.path.join(root_dir, name)))
            h, wf, ch = arr.shape
            if h != H or wf % W != 0:
                raise ValueError("%s: a %dx%d strip is not a row of %dx%d frames" % (name, h, wf, H, W))
            self.meta.append((off, wf, ch, wf // W))
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
        """-> (frames [source, driving...], hflip, x1, y1, pad_top, pad_left, out_h, out_w)"""
        H, W, _ = self.image_shape
        if not self.is_train:                                   # VideoToTensor: every frame, no augmentation
            return list(range(frame_count)), 0, 0, 0, 0, 0, H, W
        # SelectRandomFrames (augmentation.py:324-345): two indices with replacement, sorted
        sel = list(np.sort(np.random.choice(range(frame_count), replace=True, size=2)))
        hflip = 0
        if self.flip is not None:                               # RandomFlip (:91-104): a time flip returns BEFORE the second draw
            if random.random() < 0.5 and self.flip.get("time_flip", False):
                sel = sel[::-1]
            elif random.random() < 0.5 and self.flip.get("horizontal_flip", False):
                hflip = 1
        x1 = y1 = pt = pl = 0
        oh, ow = H, W
        if self.crop is not None:                               # RandomCrop (:135-171) incl. its pad_clip and its quirks
            oh, ow = self.crop
            pt = 0 if oh < H else (oh - H) // 2
            pl = 0 if ow < W else (ow - W) // 2
            im_h = H if oh < H else H + (oh - H) // 2 + (oh - H + 1) // 2
            im_w = W if ow < W else W + (ow - W) // 2 + (ow - W + 1) // 2
            x1 = 0 if oh == im_h else random.randint(0, im_w - ow)      # (sic: the height decides whether x is drawn)
            y1 = 0 if ow == im_w else random.randint(0, im_h - oh)
        return sel, hflip, x1, y1, pt, pl, oh, ow

    def _jobs(self, indices):
        """draw every sample of a batch -> (job rows, per-tensor frame counts, output size)"""
        drawn = [(i,) + tuple(self._draw(self.meta[i][3])) for i in indices]
        sizes = {(d[7], d[8]) for d in drawn}
        assert len(sizes) == 1
        return drawn, sizes.pop()

    def _launch(self, rows, total_floats, oh, ow):
        rec = np.zeros(len(rows), dtype=JOB)
        for k, r in enumerate(rows):
            rec[k] = r
        table = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self.device, non_blocking=True)
        out = torch.empty(total_floats, dtype=torch.float32, device=self.device)
        C = self.image_shape[2]
        for k0 in range(0, len(rows), 65535):
            n = min(65535, len(rows) - k0)
            mops._call("mnk_frames_gather", out, mops._p(self.pool), table.data_ptr() + k0 * JOB.itemsize, n, oh, ow, C, mops._p(out))
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
            for b, (i, sel, hflip, x1, y1, pt, pl, _, _) in enumerate(drawn):
                off, wf, ch, _ = self.meta[i]
                rows.append((off, b * C * plane, plane, wf, H, W, ch, int(sel[0]), hflip, x1, y1, pt, pl, 0, 0))
                for d, f in enumerate(sel[1:]):
                    rows.append((off, src_floats + (b * C * d_drv + d) * plane, d_drv * plane, wf, H, W, ch, int(f), hflip, x1,
                                 y1, pt, pl, 0, 0))
            out = self._launch(rows, src_floats + B * C * d_drv * plane, oh, ow)
            return {"source": out[:src_floats].view(B, C, 1, oh, ow), "video": out[src_floats:].view(B, C, d_drv, oh, ow),
                    "name": [self.images[i] for i in indices]}
        for b, (i, sel, hflip, x1, y1, pt, pl, _, _) in enumerate(drawn):
            off, wf, ch, _ = self.meta[i]
            for d, f in enumerate(sel):
                rows.append((off, (b * C * nf + d) * plane, nf * plane, wf, H, W, ch, int(f), hflip, x1, y1, pt, pl, 0, 0))
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
