"""CPU restatement (numpy) of the reference's non-integer augmentations -- TEST INFRASTRUCTURE ONLY (the checker of
tests/test_frames.py for csrc/frames.hip::frames_augment_kernel; nothing in monkey-net_amd/ imports it).

The reference's RandomResize / RandomRotation / ColorJitter (augmentation.py:105-133,175-214,217-320) are thin wrappers around
third-party functions that are NOT in this image and not vendored by the reference (requirements.txt pins scikit-image==0.14.0,
Pillow==5.2.0, torchvision==0.2.1, numpy==1.15.0), so this file restates the published algorithms of exactly those versions,
function by function, and says where each statement comes from.  **The skimage half (resize / rotate / warp) is pinned on
INTERIOR pixels to scipy.ndimage since round 6**: scikit-image is neither installed nor installable offline, but
scipy.ndimage.affine_transform(order = 0 / 1, mode='constant') -- with scipy.ndimage.gaussian_filter in front for the
anti-aliased resize -- is an independent third-party implementation of the same arithmetic wherever every tap lies inside the
source frame; rotate_bilinear / resize equal it there (order 1 to 1e-13 in float64, order 0 exactly), and the device kernel
equals scipy directly on the same pixels (tests/test_frames.py::test_rotation_and_resize_*_scipy_ndimage_*).  What stays a
restatement of skimage 0.14 with nothing to check it against: the BORDER rule (out-of-image taps read cval one by one; scipy's
'constant' mode drops the whole sample), the clip of a warp's output to its input's value range, and img_as_ubyte /
img_as_float.  **The Pillow half is pinned (round 5) -- to the INSTALLED Pillow (12.2.0), not to the 5.2.0 the reference
pins**: rgb2hsv_u8 / hsv2rgb_u8 equal the installed Pillow's Image.convert on ALL 2^24 RGB and ALL 2^24 HSV triples, and
adjust_hue reproduces a golden made with that Pillow under torchvision 0.2.1's five adjust_hue statements
(oracle/make_golden_hue.py -> tests/golden/hue_pillow.npz, HUE_PILLOW_REPORT.txt; tests/test_frames.py::test_hue_*).  Two known
differences between the installed and the pinned versions (no shipped config reaches them): Pillow >= 7.1 rounds the
16-bit fixed-point luma of convert('L') (+0x8000) where 5.2.0 truncates -- rgb2l_u8 below and csrc/frames.hip::aug_luma follow
the INSTALLED library, so the saturation / contrast terms can differ from the reference's pinned Pillow by one level --, and
scipy 1.1.0 evaluated the gaussian kernel's exponent as (c * x) * x where the installed scipy and gaussian_kernel1d below use
c * x ** 2 (one ulp in a weight).  What is checked besides (tests/test_frames.py): the device kernel
reproduces this file, and the integer-exact parts of the pipeline around it reproduce the unmodified reference
(tests/golden/frames_shapes.npz).

    skimage.transform.resize(img, (rows, cols), order=1, preserve_range=True, mode='constant', anti_aliasing=True)
        0.14.0 transform/_warps.py::resize: image.astype(double); anti-aliasing = ndi.gaussian_filter with
        sigma = max(0, (in / out - 1) / 2) per axis (scipy truncates the kernel at int(4 sigma + 0.5) taps each side: ONE tap --
        the identity -- for every ratio >= 0.8, which covers ratio: [0.9, 1.1] of config/moving-gif.yaml and actions.yaml; wider
        down-scalings are refused here and by the kernel); then warp() with the affine map  in = scale * (out + 0.5) - 0.5,
        order 1, mode 'constant', cval 0, clip=True.
    skimage.transform.rotate(img, angle, preserve_range=True)
        0.14.0 _warps.py::rotate: centre (cols / 2 - 0.5, rows / 2 - 0.5), inverse map T(c) R(angle) T(-c), warp() as above.
    skimage.transform.warp, order 1, 3-D image
        0.14.0 _warps_cy.pyx::_warp_fast per channel -> interpolation.pxd::bilinear_interpolation: minr = floor(r), maxr = ceil(r)
        (likewise c), out-of-image taps read cval (mode 'C'); then _clip_warp_output: np.clip to [image.min(), image.max()] of
        the whole (H, W, C) input.
    skimage.img_as_ubyte / img_as_float
        0.14.0 util/dtype.py::convert: float -> uint8 = clip(rint(x * 255)) in the input's float type; uint8 -> float64 =
        x * (1 / 255).
    torchvision.transforms.functional.adjust_hue (0.2.1): PIL RGB -> HSV, h += uint8(hue_factor * 255) with uint8 wrap-around,
        HSV -> RGB.
    PIL Image.convert('HSV') / .convert('RGB') (Pillow 5.2.0 libImaging/Convert.c::rgb2hsv, hsv2rgb): the colorsys formulas in
        C float / double arithmetic with (int) truncation on the way to HSV and round() on the way back."""
import math

import numpy as np


# ---- skimage 0.14.0: warp, order 1, mode 'constant', cval 0, clip ------------------------------------------------------------
def _bilinear_constant(img, r, c):
    """interpolation.pxd::bilinear_interpolation on one channel plane `img` (rows, cols) at float64 coordinate arrays r, c"""
    rows, cols = img.shape
    minr, minc = np.floor(r).astype(np.int64), np.floor(c).astype(np.int64)
    maxr, maxc = np.ceil(r).astype(np.int64), np.ceil(c).astype(np.int64)
    dr, dc = r - minr, c - minc

    def px(rr, cc):
        ok = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols)
        return np.where(ok, img[np.clip(rr, 0, rows - 1), np.clip(cc, 0, cols - 1)], 0.0)

    top = (1 - dc) * px(minr, minc) + dc * px(minr, maxc)
    bottom = (1 - dc) * px(maxr, minc) + dc * px(maxr, maxc)
    return (1 - dr) * top + dr * bottom


def _warp_affine(image, matrix, out_rows, out_cols):
    """warp(image (H, W, C) float64, inverse map `matrix` (3x3, acting on (col, row, 1)), order=1, mode='constant', clip=True)"""
    tfr, tfc = np.meshgrid(np.arange(out_rows, dtype=np.float64), np.arange(out_cols, dtype=np.float64), indexing="ij")
    c = matrix[0, 0] * tfc + matrix[0, 1] * tfr + matrix[0, 2]
    r = matrix[1, 0] * tfc + matrix[1, 1] * tfr + matrix[1, 2]
    out = np.stack([_bilinear_constant(image[..., ch], r, c) for ch in range(image.shape[2])], axis=-1)
    lo, hi = image.min(), image.max()                 # _clip_warp_output (cval 0 inside [lo, hi] or not: clip wins for order 1
    preserve = not (lo <= 0.0 <= hi)                  #  unless cval lies outside the range: those pixels keep cval)
    mask = out == 0.0 if preserve else None
    out = np.clip(out, lo, hi)
    if preserve:
        out[mask] = 0.0
    return out


def _c_round_half_away(x):
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5))


def gaussian_kernel1d(sigma):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius) with gaussian_filter1d's radius = int(truncate * sigma + 0.5),
    truncate = 4.0: the weights scipy hands to correlate1d (symmetric: the reversal is a no-op)"""
    radius = int(4.0 * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


def _correlate1d_constant(img, w, radius, axis):
    """ni_filters.c::NI_Correlate1D, its branch for symmetric weights, mode 'constant' with cval 0:
    out[l] = in[l] * w[mid] + sum over jj = -radius .. -1 of (in[l + jj] + in[l - jj]) * w[jj + mid], added in that order"""
    a = np.moveaxis(img, axis, 0)
    n = a.shape[0]
    pad = np.zeros((radius,) + a.shape[1:])
    ext = np.concatenate([pad, a, pad], 0)
    out = ext[radius:radius + n] * w[radius]
    for jj in range(-radius, 0):
        out = out + (ext[radius + jj:radius + jj + n] + ext[radius - jj:radius - jj + n]) * w[jj + radius]
    return np.moveaxis(out, 0, axis)


def gaussian_aa(image, row_scale, col_scale):
    """the anti-aliasing step of skimage 0.14's resize: ndi.gaussian_filter(image, (sigma_r, sigma_c, 0), cval=0, mode='constant')
    with sigma = max(0, (in / out - 1) / 2) per axis -- rows first, then columns, each a correlate1d; an axis with sigma <= 1e-15 is
    skipped.  Pinned (round 5) against the installed scipy's gaussian_filter, bit for bit (tests/test_frames.py)."""
    out = image
    for axis, f in ((0, row_scale), (1, col_scale)):
        sigma = max(0.0, (f - 1.0) / 2.0)
        if sigma > 1e-15:
            w, radius = gaussian_kernel1d(sigma)
            out = _correlate1d_constant(out, w, radius, axis)
    return out


def resize(img, new_rows, new_cols, order):
    """skimage.transform.resize(img, (new_rows, new_cols), order=order, preserve_range=True, mode='constant',
    anti_aliasing=True), order 1 (bilinear) or 0 -- resize_clip (augmentation.py:55) passes order=1 ONLY for
    interpolation == 'bilinear'; RandomResize's default 'nearest', which every shipped config uses, is order 0:
    interpolation.pxd::nearest_neighbour_interpolation = the pixel at (round(r), round(c)) with C's round(), cval 0 outside, and
    no clipping (_clip_warp_output acts on order != 0 only)."""
    if order == 1:
        return resize_bilinear(img, new_rows, new_cols)
    image = img.astype(np.float64)
    rows, cols = image.shape[:2]
    row_scale, col_scale = float(rows) / new_rows, float(cols) / new_cols
    image = gaussian_aa(image, row_scale, col_scale)          # (one tap -- the identity -- for ratios >= 0.8)
    tfr, tfc = np.meshgrid(np.arange(new_rows, dtype=np.float64), np.arange(new_cols, dtype=np.float64), indexing="ij")
    c = col_scale * tfc + 0.0 * tfr + (col_scale / 2.0 - 0.5)
    r = 0.0 * tfc + row_scale * tfr + (row_scale / 2.0 - 0.5)
    rr, cc = _c_round_half_away(r).astype(np.int64), _c_round_half_away(c).astype(np.int64)
    ok = (rr >= 0) & (rr < rows) & (cc >= 0) & (cc < cols)
    out = image[np.clip(rr, 0, rows - 1), np.clip(cc, 0, cols - 1)]
    return np.where(ok[..., None], out, 0.0)


def resize_bilinear(img, new_rows, new_cols):
    """skimage.transform.resize(img, (new_rows, new_cols), order=1, preserve_range=True, mode='constant', anti_aliasing=True)"""
    image = img.astype(np.float64)
    rows, cols = image.shape[:2]
    row_scale, col_scale = float(rows) / new_rows, float(cols) / new_cols
    image = gaussian_aa(image, row_scale, col_scale)   # anti-aliasing: scipy's gaussian_filter1d truncates at int(4 sigma + .5)
    m = np.array([[col_scale, 0.0, col_scale / 2.0 - 0.5], [0.0, row_scale, row_scale / 2.0 - 0.5], [0.0, 0.0, 1.0]])
    return _warp_affine(image, m, new_rows, new_cols)


def rotate_bilinear(img, angle_deg):
    """skimage.transform.rotate(image=img, angle=angle_deg, preserve_range=True)"""
    image = img.astype(np.float64)
    rows, cols = image.shape[:2]
    cx, cy = cols / 2.0 - 0.5, rows / 2.0 - 0.5
    a = math.radians(angle_deg)
    co, si = math.cos(a), math.sin(a)
    # tform3 + tform2 + tform1 = T(center) @ R(a) @ T(-center)   (SimilarityTransform: [[cos, -sin, tx], [sin, cos, ty]])
    m = np.array([[co, -si, cx - co * cx + si * cy], [si, co, cy - si * cx - co * cy], [0.0, 0.0, 1.0]])
    return _warp_affine(image, m, rows, cols)


# ---- skimage 0.14.0 dtype conversions ---------------------------------------------------------------------------------------
def img_as_ubyte(x):
    y = x * x.dtype.type(255)                          # (the input's own float type: float64 after resize / rotate, else float32)
    return np.clip(np.rint(y), 0, 255).astype(np.uint8)


def img_as_float(u8):
    return np.multiply(u8, 1.0 / 255, dtype=np.float64)


# ---- Pillow 5.2.0 libImaging/Convert.c -----------------------------------------------------------------------------------------
def rgb2hsv_u8(rgb):
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    gray = maxc == minc
    cr = np.where(gray, 1, maxc - minc).astype(np.float32)
    mx = np.where(maxc == 0, 1, maxc).astype(np.float32)
    s = cr / mx                                                                   # float
    rc, gc, bc = ((maxc - r).astype(np.float32) / cr, (maxc - g).astype(np.float32) / cr, (maxc - b).astype(np.float32) / cr)
    # 2.0 + rc - bc: the C expression promotes to double
    h = np.where(r == maxc, (bc - gc).astype(np.float32),
                 np.where(g == maxc, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)))
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)         # h = fmod((h/6.0 + 1.0), 1.0), stored as float
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)         # (int) truncation, CLIP8
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], axis=-1).astype(np.uint8)


def _c_round(x):
    """C round(): half away from zero"""
    return np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int64)


def hsv2rgb_u8(hsv):
    h, s, v = (hsv[..., i].astype(np.int64) for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int64)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32).astype(np.float64)     # float f
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32).astype(np.float64)   # float fs
    vf = v.astype(np.float32).astype(np.float64)
    p = np.clip(_c_round(vf * (1.0 - fs)), 0, 255)
    q = np.clip(_c_round(vf * (1.0 - fs * f)), 0, 255)
    t = np.clip(_c_round(vf * (1.0 - fs * (1.0 - f))), 0, 255)
    k = i % 6
    r = np.choose(k, [v, q, p, p, t, v])
    g = np.choose(k, [t, v, v, q, p, p])
    b = np.choose(k, [p, p, t, v, v, q])
    gray = s == 0
    return np.stack([np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)], axis=-1).astype(np.uint8)


def hue_shift_u8(hue_factor):
    """np.uint8(hue_factor * 255): the C cast of a double -- truncation toward zero, then wrap-around modulo 256"""
    return int(math.trunc(hue_factor * 255)) % 256


# ---- Pillow ImageEnhance (what torchvision 0.2.1's adjust_brightness / adjust_saturation / adjust_contrast call) -------------------
# Pinned against the installed Pillow by oracle/make_golden_hue.py: rgb2l_u8 on all 2^24 triples, blend_u8 on all 256 x 256 value
# pairs for a ladder of factors, the three enhancers on random images.
JIT_BRIGHTNESS, JIT_SATURATION, JIT_HUE, JIT_CONTRAST = 1, 2, 3, 4      # (the order ColorJitter.__call__ appends them in)


def rgb2l_u8(rgb):
    """Image.convert('L') of an RGB image: libImaging/Convert.c L24(rgb) >> 16 = ITU-R 601-2 luma in 16-bit fixed point"""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(a, v, factor):
    """Image.blend(a, v, factor) (libImaging/Blend.c): a + factor * (v - a) in C float arithmetic; inside [0, 1] the result is cast
    (truncated), outside it is clipped to [0, 255] first"""
    f = np.float32(factor)
    t = (a.astype(np.int32).astype(np.float32) + f * (v.astype(np.int32) - a.astype(np.int32)).astype(np.float32)).astype(np.float32)
    if 0 <= f <= 1:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def jitter_u8(u8, ops):
    """ops: [(code, factor), ...] in the order they are applied (ColorJitter shuffles them): ImageEnhance.Brightness = blend with
    black, .Color = blend with the image's own luma, .Contrast = blend with the constant int(mean(luma) + 0.5); hue as adjust_hue"""
    for code, f in ops:
        if code == JIT_BRIGHTNESS:
            u8 = blend_u8(np.zeros_like(u8), u8, f)
        elif code == JIT_SATURATION:
            u8 = blend_u8(np.repeat(rgb2l_u8(u8)[..., None], 3, -1), u8, f)
        elif code == JIT_CONTRAST:
            lum = rgb2l_u8(u8)
            mean = int(int(lum.astype(np.int64).sum()) / lum.size + 0.5)          # ImageStat: sum / count as Python floats, then int()
            u8 = blend_u8(np.full_like(u8, mean), u8, f)
        elif code == JIT_HUE:
            hsv = rgb2hsv_u8(u8)
            hsv[..., 0] = (hsv[..., 0].astype(np.int64) + hue_shift_u8(f)) % 256
            u8 = hsv2rgb_u8(hsv)
        else:
            raise ValueError(code)
    return u8


def adjust_jitter(img_float, ops):
    """ColorJitter.__call__ for one image (augmentation.py:269-300): img_as_ubyte -> ToPILImage -> the shuffled enhancers ->
    np.array -> img_as_float -> astype('float32')"""
    return img_as_float(jitter_u8(img_as_ubyte(img_float), ops)).astype(np.float32)


def adjust_hue(img_float, hue_factor):
    """ColorJitter.__call__ for one image with only `hue` set (augmentation.py:269-300): img_as_ubyte -> ToPILImage ->
    torchvision adjust_hue -> np.array -> img_as_float -> astype('float32')"""
    u8 = img_as_ubyte(img_float)
    hsv = rgb2hsv_u8(u8)
    hsv[..., 0] = (hsv[..., 0].astype(np.int64) + hue_shift_u8(hue_factor)) % 256
    return img_as_float(hsv2rgb_u8(hsv)).astype(np.float32)


# ---- the reference's pipeline on one sample (augmentation.py:369-389) with the random choices handed in ----------------------
def pad_clip_edge(clip, h, w):
    """augmentation.py:33-39 (skimage.util.pad = numpy.pad, mode='edge')"""
    im_h, im_w = clip[0].shape[:2]
    pad_h = (0, 0) if h < im_h else ((h - im_h) // 2, (h - im_h + 1) // 2)
    pad_w = (0, 0) if w < im_w else ((w - im_w) // 2, (w - im_w + 1) // 2)
    return np.pad(np.asarray(clip), ((0, 0), pad_h, pad_w, (0, 0)), mode="edge")


def pipeline(frames_u8, sel, hflip, angle, new_hw, crop, x1, y1, hue_factor, resize_order=0, jitter=None):
    """frames_u8 (F, H, W, 3) uint8 of one video; the draws of mnk.frames.DeviceFramesDataset._draw -> (C, D, h, w) float32 in
    SplitSourceDriving's layout (source first).  angle / new_hw / crop / hue_factor: None = that transform is not configured."""
    clip = [np.multiply(frames_u8[f], 1.0 / 255, dtype=np.float32) for f in sel]          # img_as_float32, selection (+ time flip)
    if hflip:
        clip = [np.fliplr(img) for img in clip]
    if angle is not None:
        clip = [rotate_bilinear(img, angle) for img in clip]
    if new_hw is not None:
        clip = [resize(img, new_hw[0], new_hw[1], resize_order) for img in clip]
    if crop is not None:
        h, w = crop
        clip = pad_clip_edge(clip, h, w)
        clip = [img[y1:y1 + h, x1:x1 + w, :] for img in clip]
    if jitter:                                   # [(code, factor), ...] in applied order (ColorJitter with more than the hue term)
        clip = [adjust_jitter(np.asarray(img), jitter) for img in clip]
    elif hue_factor is not None:
        clip = [adjust_hue(np.asarray(img), hue_factor) for img in clip]
    return np.array(clip, dtype="float32").transpose((3, 0, 1, 2))
