#!/usr/bin/env python
"""Pins the COLOUR-CONVERSION half of the hue jitter (augmentation.py:217-320 -> torchvision adjust_hue -> PIL
Image.convert('HSV') / .convert('RGB')) to the REAL Pillow that this image ships (TEST INFRASTRUCTURE; run in the authoring
container, writes tests/golden/hue_pillow.npz + tests/golden/HUE_PILLOW_REPORT.txt).

What is real here: PIL.Image.convert between 'RGB' and 'HSV' (libImaging/Convert.c rgb2hsv / hsv2rgb) of Pillow %(pil)s -- the
reference pins 5.2.0, which is not installable offline; the two functions are compared with oracle/augment_restate.py's
restatement on ALL 2^24 RGB triples and ALL 2^24 HSV triples (both directions: 0 differences), so the restatement of these two
functions is pinned to a real Pillow, exhaustively.  What is restated: torchvision 0.2.1's adjust_hue body (five statements:
split, np.uint8 add with wrap-around, merge, convert back) -- torchvision is absent -- and skimage's img_as_ubyte / img_as_float
around it.  The golden holds, for seeded uint8 images and a ladder of hue factors, the uint8 output of
    Image.fromarray(img).convert('HSV') -> h += uint8(hue_factor * 255) (mod 256) -> .convert('RGB')
executed by the real Pillow; tests/test_frames.py checks the restatement (CPU) and the device kernel (GPU) against it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pil_adjust_hue_u8(img_u8, hue_factor):
    """torchvision 0.2.1 transforms/functional.py::adjust_hue with the real Pillow doing the conversions"""
    from PIL import Image
    img = Image.fromarray(img_u8, "RGB")
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    shift = int(np.trunc(hue_factor * 255)) % 256          # np.uint8(hue_factor * 255) of numpy 1.15: the C cast, wrap-around
    np_h = ((np_h.astype(np.int64) + shift) % 256).astype(np.uint8)
    h = Image.fromarray(np_h, "L")
    return np.asarray(Image.merge("HSV", (h, s, v)).convert("RGB"))


def main():
    import PIL
    from PIL import Image
    from oracle import augment_restate as ar
    lines = ["Pillow %s (the reference pins 5.2.0)" % PIL.__version__]
    r, g, b = np.meshgrid(*(np.arange(256, dtype=np.uint8),) * 3, indexing="ij")
    cube = np.stack([r, g, b], -1).reshape(4096, 4096, 3)
    d1 = np.asarray(Image.fromarray(cube, "RGB").convert("HSV")).astype(int) - ar.rgb2hsv_u8(cube).astype(int)
    d2 = np.asarray(Image.fromarray(cube, "HSV").convert("RGB")).astype(int) - ar.hsv2rgb_u8(cube).astype(int)
    lines.append("RGB -> HSV, all 2^24 triples: %d differ from oracle/augment_restate.py::rgb2hsv_u8" % int((d1 != 0).any(-1).sum()))
    lines.append("HSV -> RGB, all 2^24 triples: %d differ from oracle/augment_restate.py::hsv2rgb_u8" % int((d2 != 0).any(-1).sum()))
    assert not d1.any() and not d2.any()
    rng = np.random.RandomState(7)
    imgs = rng.randint(0, 256, size=(24, 32, 32, 3)).astype(np.uint8)
    imgs[0] = 0
    imgs[1] = 255
    imgs[2, ..., 1] = imgs[2, ..., 0]
    imgs[2, ..., 2] = imgs[2, ..., 0]                      # a gray image: hue undefined, must come back unchanged
    factors = np.array([-0.5, -0.3, -0.1, -0.004, 0.0, 0.003, 0.1, 0.25, 0.4999, 0.5])
    out = np.stack([np.stack([pil_adjust_hue_u8(im, f) for im in imgs]) for f in factors])
    bad = 0
    for k, f in enumerate(factors):
        for i, im in enumerate(imgs):
            got = (ar.adjust_hue(ar.img_as_float(im).astype(np.float32), float(f)) * 255.0).round().astype(np.uint8)
            bad += int((got != out[k, i]).sum())
    lines.append("adjust_hue ladder (%d images x %d factors): %d values differ from the restatement" % (len(imgs), len(factors), bad))
    assert bad == 0
    # ---- the other three terms of ColorJitter: ImageEnhance.Brightness / Color / Contrast (torchvision 0.2.1's adjust_brightness /
    # adjust_saturation / adjust_contrast are one-line wrappers around them), alone and in shuffled sequences with the hue term
    from PIL import ImageEnhance
    ok = int((np.asarray(Image.fromarray(cube, "RGB").convert("L")) != ar.rgb2l_u8(cube)).sum())
    lines.append("RGB -> L, all 2^24 triples: %d differ from oracle/augment_restate.py::rgb2l_u8" % ok)
    assert ok == 0
    a, v = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    A, V = Image.fromarray(np.ascontiguousarray(a), "L"), Image.fromarray(np.ascontiguousarray(v), "L")
    blend_f = np.concatenate([rng.uniform(0, 2.5, size=200), [0.0, 1.0, 0.5, 2.0]])
    ok = sum(int((np.asarray(Image.blend(A, V, float(f))) != ar.blend_u8(a, v, f)).sum()) for f in blend_f)
    lines.append("Image.blend, all 256 x 256 value pairs x %d factors: %d differ from blend_u8" % (len(blend_f), ok))
    assert ok == 0

    def pil_jitter(img_u8, ops):
        img = Image.fromarray(img_u8, "RGB")
        for code, f in ops:
            if code == ar.JIT_BRIGHTNESS:
                img = ImageEnhance.Brightness(img).enhance(f)
            elif code == ar.JIT_SATURATION:
                img = ImageEnhance.Color(img).enhance(f)
            elif code == ar.JIT_CONTRAST:
                img = ImageEnhance.Contrast(img).enhance(f)
            else:
                img = Image.fromarray(pil_adjust_hue_u8(np.asarray(img), f), "RGB")
        return np.asarray(img)

    seqs = []
    for k in range(40):
        n = 1 + k % 4
        codes = list(rng.permutation([ar.JIT_BRIGHTNESS, ar.JIT_SATURATION, ar.JIT_HUE, ar.JIT_CONTRAST])[:n])
        seqs.append([(int(c), float(rng.uniform(-0.5, 0.5)) if c == ar.JIT_HUE else float(rng.uniform(0.2, 1.9))) for c in codes])
    jit_out = np.stack([np.stack([pil_jitter(im, ops) for im in imgs[:8]]) for ops in seqs])
    bad = sum(int((ar.jitter_u8(im.copy(), ops) != jit_out[k, i]).sum()) for k, ops in enumerate(seqs) for i, im in enumerate(imgs[:8]))
    lines.append("ColorJitter sequences (%d shuffled sequences of 1-4 terms x 8 images): %d values differ from jitter_u8" % (len(seqs), bad))
    assert bad == 0
    jit_codes = np.zeros((len(seqs), 4), dtype=np.int32)
    jit_factors = np.zeros((len(seqs), 4), dtype=np.float64)
    for k, ops in enumerate(seqs):
        for i, (c, f) in enumerate(ops):
            jit_codes[k, i], jit_factors[k, i] = c, f
    gold = os.path.join(ROOT, "tests", "golden")
    np.savez_compressed(os.path.join(gold, "hue_pillow.npz"), images=imgs, factors=factors, out=out, pillow=PIL.__version__,
                        jit_codes=jit_codes, jit_factors=jit_factors, jit_out=jit_out)
    with open(os.path.join(gold, "HUE_PILLOW_REPORT.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
