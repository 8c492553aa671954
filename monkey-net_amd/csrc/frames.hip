// Device-side input path: from decoded uint8 "stacked frame" strips resident in HBM to the
// fp32 (B, C, D, H, W) batches the hot path consumes, in one launch per batch.
//
// Replaces, for the transforms that are integer-exact (frame selection, time / horizontal flip, edge padding + crop,
// uint8 -> float32, HWC -> CDHW):
//   frames_dataset.py:14-29   read_video: strip (H, W*F, C) -> frames (F, H, W, C), gray -> RGB, RGBA -> RGB, img_as_float32
//   augmentation.py:91-104    RandomFlip           (np.fliplr per frame; the time flip is a choice of frame indices)
//   augmentation.py:135-171   RandomCrop           (pad_clip with mode='edge', then crop at (y1, x1))
//   augmentation.py:324-366   SelectRandomFrames, SplitSourceDriving / VideoToTensor (transpose (3, 0, 1, 2))
// which the reference runs on the host with numpy / skimage inside 4 DataLoader workers (train.py:99) and then copies over
// PCIe.  Here the dataset lives in HBM as it was decoded (uint8, 1 / 4 of the fp32 bytes); the per-sample random choices are
// made on the host in the reference's draw order (mnk/frames.py) and travel as a small job table.
// HBM-bound byte work: one thread per output pixel reads <= 4 source bytes and writes 3 floats into 3 channel planes
// (coalesced along W in every plane); nothing here is shaped like a GEMM.
#include "mnk_common.h"

using namespace mnk;

namespace {

// one thread per (job, output pixel): grid (ceil(H * W / 256), jobs)
__global__ void __launch_bounds__(256) frames_gather_kernel(const unsigned char* __restrict__ pool,
                                                            const MnkFrameJob* __restrict__ jobs, int H, int W, int Cout,
                                                            float* __restrict__ out) {
    const MnkFrameJob j = jobs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int h = p / W, w = p - h * W;
    // crop position inside the edge-padded frame -> position inside the (flipped) frame: pad_clip(mode='edge') repeats the
    // border pixel, i.e. the coordinate is clamped
    int r = h + j.y1 - j.pad_top, c = w + j.x1 - j.pad_left;
    r = r < 0 ? 0 : (r > j.in_h - 1 ? j.in_h - 1 : r);
    c = c < 0 ? 0 : (c > j.in_w - 1 ? j.in_w - 1 : c);
    if (j.hflip) c = j.in_w - 1 - c;                                // np.fliplr happened before padding and cropping
    // frame f of the strip occupies columns [f * in_w, (f + 1) * in_w)  (frames_dataset.py:25-29)
    const unsigned char* px = pool + j.strip_offset + ((size_t)r * j.strip_w + (size_t)j.frame * j.in_w + c) * j.channels;
    const float k = 1.f / 255.f;                                    // img_as_float32: uint8 * float32(1 / 255)
    float v[3];
    if (j.channels >= 3) {                                          // RGB, or RGBA with the alpha channel dropped (:20-21)
        v[0] = (float)px[0] * k;
        v[1] = (float)px[1] * k;
        v[2] = (float)px[2] * k;
    } else {                                                        // gray (+ alpha): gray2rgb replicates (:17-18)
        v[0] = v[1] = v[2] = (float)px[0] * k;
    }
    float* o = out + j.out_offset + p;
    for (int ch = 0; ch < Cout; ++ch) o[(size_t)ch * j.chan_stride] = v[ch < 3 ? ch : 2];
}

// ---- the non-integer augmentations (round 4): RandomRotation -> RandomResize -> RandomCrop -> ColorJitter(hue) --------------------
//   augmentation.py:175-214  RandomRotation = skimage.transform.rotate(img, angle, preserve_range=True)
//   augmentation.py:105-133  RandomResize   = skimage.transform.resize(img, (new_h, new_w), order=1, preserve_range=True,
//                                             mode='constant', anti_aliasing=True)
//   augmentation.py:217-320  ColorJitter    = img_as_ubyte -> PIL -> torchvision adjust_hue -> np.array -> img_as_float -> float32
// in the arithmetic of the versions the reference pins (scikit-image 0.14.0, Pillow 5.2.0, torchvision 0.2.1), restated in
// oracle/augment_restate.py with the citations; this kernel follows that restatement statement by statement (float64 warps as
// skimage computes them, Pillow's float / double mix in the colour conversions), so the two agree to the last bit except where
// a libm call differs.  The two warps are NOT composed into one resampling: a pixel of the resized frame is the bilinear blend
// of four pixels of the ROTATED frame, each of which is itself a bilinear blend of four source bytes, clipped to the source
// frame's value range -- 16 byte reads per output value and no intermediate image.  One thread per output pixel.
#pragma clang fp contract(off)      // (a * b + c must round twice, as numpy does)

struct AugSrc {
    const unsigned char* base;      // the frame's first byte in the strip
    int strip_w, in_h, in_w, channels, hflip;
};

// channel values of the float32 frame img_as_float32(uint8) at (r, c) -- np.fliplr applied -- or cval = 0 outside the image
__device__ __forceinline__ void aug_src(const AugSrc& s, long r, long c, double v[3]) {
    if (r < 0 || r >= s.in_h || c < 0 || c >= s.in_w) {
        v[0] = v[1] = v[2] = 0.0;
        return;
    }
    if (s.hflip) c = s.in_w - 1 - c;
    const unsigned char* px = s.base + ((size_t)r * s.strip_w + c) * s.channels;
    const float k = 1.f / 255.f;
    if (s.channels >= 3) {
        v[0] = (double)((float)px[0] * k);
        v[1] = (double)((float)px[1] * k);
        v[2] = (double)((float)px[2] * k);
    } else {
        v[0] = v[1] = v[2] = (double)((float)px[0] * k);
    }
}

// _clip_warp_output: clip to the input's range; a pixel that is exactly cval (0) keeps it when 0 lies outside the range
__device__ __forceinline__ double aug_clip(double x, double lo, double hi) {
    if (!(lo <= 0.0 && 0.0 <= hi) && x == 0.0) return 0.0;
    return x < lo ? lo : (x > hi ? hi : x);
}

template <class Tap>
__device__ __forceinline__ void aug_bilinear(double r, double c, Tap tap, double out[3]) {
    const double fr = floor(r), fc = floor(c);
    const long minr = (long)fr, minc = (long)fc, maxr = (long)ceil(r), maxc = (long)ceil(c);
    const double dr = r - fr, dc = c - fc;
    double a[3], b[3], cc[3], d[3];
    tap(minr, minc, a);
    tap(minr, maxc, b);
    tap(maxr, minc, cc);
    tap(maxr, maxc, d);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double top = (1 - dc) * a[i] + dc * b[i];
        const double bottom = (1 - dc) * cc[i] + dc * d[i];
        out[i] = (1 - dr) * top + dr * bottom;
    }
}

// pixel (r, c) of the rotated frame (or of the source frame when the job has no rotation); 0 outside the frame
__device__ __forceinline__ void aug_rotated(const MnkAugJob& j, const AugSrc& s, long r, long c, double v[3]) {
    if (!(j.flags & 1)) {
        aug_src(s, r, c, v);
        return;
    }
    if (r < 0 || r >= s.in_h || c < 0 || c >= s.in_w) {
        v[0] = v[1] = v[2] = 0.0;
        return;
    }
    const double cc = j.rot[0] * (double)c + j.rot[1] * (double)r + j.rot[2];
    const double rr = j.rot[3] * (double)c + j.rot[4] * (double)r + j.rot[5];
    aug_bilinear(rr, cc, [&](long y, long x, double* o) { aug_src(s, y, x, o); }, v);
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = aug_clip(v[i], (double)j.vmin, (double)j.vmax);
}

// pixel (r, c) of the frame the resize samples: the (rotated) frame behind skimage's anti-aliasing filter (flags & 16) --
// ndi.gaussian_filter(image, (sigma_r, sigma_c, 0), mode='constant', cval=0), i.e. scipy's correlate1d along the rows, then along
// the columns of that result, each in its symmetric-kernel form  out[l] = in[l] w[mid] + sum_{jj=-R..-1} (in[l+jj] + in[l-jj]) w[jj+mid]
// (the additions in that order; zeros beyond the line's ends).  aa_wr / aa_wc hold w[0 .. R] (w[R] = the centre weight), made on
// the host as scipy makes them.  Pinned against the installed scipy bit for bit (oracle/augment_restate.py::gaussian_aa).
__device__ __forceinline__ void aug_filtered(const MnkAugJob& j, const AugSrc& s, long r, long c, double v[3]) {
    if (!(j.flags & 16)) {
        aug_rotated(j, s, r, c, v);
        return;
    }
    if (r < 0 || r >= s.in_h || c < 0 || c >= s.in_w) {
        v[0] = v[1] = v[2] = 0.0;
        return;
    }
    const int Rr = j.aa_rr, Rc = j.aa_rc;
    // T(r, cc): the row pass at column cc (0 outside the image: the column pass pads ITS input line with zeros)
    auto rowpass = [&](long cc, double t[3]) {
        if (cc < 0 || cc >= s.in_w) {
            t[0] = t[1] = t[2] = 0.0;
            return;
        }
        double x0[3];
        aug_rotated(j, s, r, cc, x0);
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = x0[i] * j.aa_wr[Rr];
        for (int jj = -Rr; jj < 0; ++jj) {
            double a[3], b[3];
            aug_rotated(j, s, r + jj, cc, a);        // (0 outside the frame)
            aug_rotated(j, s, r - jj, cc, b);
#pragma unroll
            for (int i = 0; i < 3; ++i) t[i] = t[i] + (a[i] + b[i]) * j.aa_wr[jj + Rr];
        }
    };
    double t0[3];
    rowpass(c, t0);
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = t0[i] * j.aa_wc[Rc];
    for (int jj = -Rc; jj < 0; ++jj) {
        double a[3], b[3];
        rowpass(c + jj, a);
        rowpass(c - jj, b);
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = v[i] + (a[i] + b[i]) * j.aa_wc[jj + Rc];
    }
}

// per job: min / max over the three channels of the frame the resize samples -- the rotated and / or anti-alias-filtered frame
// (skimage clips a warp's output to the range of its INPUT image)
__global__ void __launch_bounds__(256) frames_rotated_range_kernel(const unsigned char* __restrict__ pool,
                                                                   const MnkAugJob* __restrict__ jobs,
                                                                   double* __restrict__ range) {
    __shared__ double lo_s[256], hi_s[256];
    const MnkAugJob j = jobs[blockIdx.x];
    const AugSrc s{pool + j.strip_offset + (size_t)j.frame * j.in_w * j.channels, j.strip_w, j.in_h, j.in_w, j.channels, j.hflip};
    double lo = 1e300, hi = -1e300;
    for (int p = threadIdx.x; p < j.in_h * j.in_w; p += 256) {
        double v[3];
        aug_filtered(j, s, p / j.in_w, p % j.in_w, v);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lo = v[i] < lo ? v[i] : lo;
            hi = v[i] > hi ? v[i] : hi;
        }
    }
    lo_s[threadIdx.x] = lo;
    hi_s[threadIdx.x] = hi;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            lo_s[threadIdx.x] = lo_s[threadIdx.x + k] < lo_s[threadIdx.x] ? lo_s[threadIdx.x + k] : lo_s[threadIdx.x];
            hi_s[threadIdx.x] = hi_s[threadIdx.x + k] > hi_s[threadIdx.x] ? hi_s[threadIdx.x + k] : hi_s[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        range[2 * blockIdx.x] = lo_s[0];
        range[2 * blockIdx.x + 1] = hi_s[0];
    }
}

// Pillow 5.2.0 libImaging/Convert.c::rgb2hsv / hsv2rgb on one pixel, with torchvision 0.2.1's uint8 hue shift in between
__device__ __forceinline__ void aug_hue(unsigned char rgb[3], int shift) {
    const unsigned char r = rgb[0], g = rgb[1], b = rgb[2];
    const unsigned char maxc = r > g ? (r > b ? r : b) : (g > b ? g : b), minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
    unsigned char uh = 0, us = 0;
    const unsigned char uv = maxc;
    if (minc != maxc) {
        const float cr = (float)(maxc - minc);
        const float s = cr / (float)maxc;
        const float rc = ((float)(maxc - r)) / cr, gc = ((float)(maxc - g)) / cr, bc = ((float)(maxc - b)) / cr;
        float h;
        if (r == maxc)
            h = bc - gc;
        else if (g == maxc)
            h = (float)(2.0 + (double)rc - (double)bc);
        else
            h = (float)(4.0 + (double)gc - (double)rc);
        h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
        int ih = (int)((double)h * 255.0), is = (int)((double)s * 255.0);
        uh = (unsigned char)(ih < 0 ? 0 : (ih > 255 ? 255 : ih));
        us = (unsigned char)(is < 0 ? 0 : (is > 255 ? 255 : is));
    }
    uh = (unsigned char)((uh + shift) & 255);                       // np_h += np.uint8(hue_factor * 255), uint8 wrap-around
    if (us == 0) {
        rgb[0] = rgb[1] = rgb[2] = uv;
        return;
    }
    const double hf = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)us / 255.0);
    auto cround = [](double x) { return x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5); };      // C round(): half away from zero
    auto clip8 = [](double x) { return (unsigned char)(x < 0.0 ? 0 : (x > 255.0 ? 255 : (int)x)); };
    const unsigned char up = clip8(cround((double)(float)uv * (1.0 - (double)fs)));
    const unsigned char uq = clip8(cround((double)(float)uv * (1.0 - (double)fs * (double)f)));
    const unsigned char ut = clip8(cround((double)(float)uv * (1.0 - (double)fs * (1.0 - (double)f))));
    switch (i % 6) {
        case 0: rgb[0] = uv, rgb[1] = ut, rgb[2] = up; break;
        case 1: rgb[0] = uq, rgb[1] = uv, rgb[2] = up; break;
        case 2: rgb[0] = up, rgb[1] = uv, rgb[2] = ut; break;
        case 3: rgb[0] = up, rgb[1] = uq, rgb[2] = uv; break;
        case 4: rgb[0] = ut, rgb[1] = up, rgb[2] = uv; break;
        default: rgb[0] = uv, rgb[1] = up, rgb[2] = uq; break;
    }
}

// ---- the other terms of ColorJitter (round 5): Pillow's ImageEnhance on one pixel -- what torchvision 0.2.1's adjust_brightness /
// adjust_saturation / adjust_contrast call.  Image.convert('L') = (19595 R + 38470 G + 7471 B + 0x8000) >> 16; Image.blend(a, v, f)
// = a + f * (v - a) in C float arithmetic (two roundings: no fused multiply-add), truncated inside [0, 1], clipped outside.
// Pinned to the installed Pillow by oracle/make_golden_hue.py (all 2^24 RGB triples, all 256 x 256 blend pairs x 204 factors).
__device__ __forceinline__ int aug_luma(const unsigned char rgb[3]) {
    return (rgb[0] * 19595 + rgb[1] * 38470 + rgb[2] * 7471 + 0x8000) >> 16;
}
__device__ __forceinline__ unsigned char aug_blend(int a, int v, float f) {
    const float t = __fadd_rn((float)a, __fmul_rn(f, (float)(v - a)));
    if (f >= 0.f && f <= 1.f) return (unsigned char)(int)t;
    return (unsigned char)(t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t));
}
// terms [0, upto) of the job's shuffled sequence; contrast_mean: int(mean(luma of the frame at that point) + 0.5)
__device__ __forceinline__ void aug_jitter(unsigned char rgb[3], const MnkAugJob& j, int upto, int contrast_mean) {
    for (int k = 0; k < upto; ++k) {
        const int op = j.jit_op[k];
        const float f = j.jit_f[k];
        if (op == 3) {
            aug_hue(rgb, j.hue_shift);
            continue;
        }
        const int lum = op == 2 ? aug_luma(rgb) : (op == 4 ? contrast_mean : 0);        // brightness blends with black
#pragma unroll
        for (int i = 0; i < 3; ++i) rgb[i] = aug_blend(lum, rgb[i], f);
    }
}

// the frame value under output pixel (h, w) of job j, before the colour jitter: crop o (resize of (rotation of the source))
__device__ __forceinline__ bool aug_value(const MnkAugJob& j, const AugSrc& s, const double* __restrict__ rot_range, int job, int h,
                                          int w, double v[3]) {
    // RandomCrop: position inside the edge-padded (rotated, resized) frame -> clamped position inside that frame
    int ry = h + j.y1 - j.pad_top, rx = w + j.x1 - j.pad_left;
    ry = ry < 0 ? 0 : (ry > j.new_h - 1 ? j.new_h - 1 : ry);
    rx = rx < 0 ? 0 : (rx > j.new_w - 1 ? j.new_w - 1 : rx);
    const bool warped = (j.flags & 27) != 0;         // float64 values from here on (skimage converts to double), else float32
    if (j.flags & 8) {
        // order 0 (RandomResize's default interpolation 'nearest' -- what every shipped config runs): the pixel at
        // (round(r), round(c)), C round(), cval outside, no clipping
        const double row_scale = (double)j.in_h / (double)j.new_h, col_scale = (double)j.in_w / (double)j.new_w;
        const double c = col_scale * (double)rx + 0.0 * (double)ry + (col_scale / 2.0 - 0.5);
        const double r = 0.0 * (double)rx + row_scale * (double)ry + (row_scale / 2.0 - 0.5);
        const double rr = r >= 0.0 ? floor(r + 0.5) : ceil(r - 0.5), cc = c >= 0.0 ? floor(c + 0.5) : ceil(c - 0.5);
        aug_filtered(j, s, (long)rr, (long)cc, v);
    } else if (j.flags & 2) {
        const double row_scale = (double)j.in_h / (double)j.new_h, col_scale = (double)j.in_w / (double)j.new_w;
        const double c = col_scale * (double)rx + 0.0 * (double)ry + (col_scale / 2.0 - 0.5);
        const double r = 0.0 * (double)rx + row_scale * (double)ry + (row_scale / 2.0 - 0.5);
        aug_bilinear(r, c, [&](long y, long x, double* o) { aug_filtered(j, s, y, x, o); }, v);
        const double lo = (j.flags & 17) ? rot_range[2 * job] : (double)j.vmin;
        const double hi = (j.flags & 17) ? rot_range[2 * job + 1] : (double)j.vmax;
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = aug_clip(v[i], lo, hi);
    } else {
        aug_rotated(j, s, ry, rx, v);
    }
    return warped;
}

__device__ __forceinline__ void aug_ubyte(const double v[3], bool warped, unsigned char rgb[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {                    // img_as_ubyte: clip(rint(x * 255)) in the image's own float type
        double y = warped ? v[i] * 255.0 : (double)((float)v[i] * 255.f);
        y = rint(y);
        rgb[i] = (unsigned char)(y < 0.0 ? 0 : (y > 255.0 ? 255 : (int)y));
    }
}

// per job with a contrast term: int(mean(luma) + 0.5) of the OUTPUT frame as it stands in front of that term (ImageEnhance.Contrast
// blends with this constant; ImageStat's mean = sum / count in double).  The sum is an integer: any order of addition gives it.
__global__ void __launch_bounds__(256) frames_contrast_mean_kernel(const unsigned char* __restrict__ pool,
                                                                   const MnkAugJob* __restrict__ jobs,
                                                                   const double* __restrict__ rot_range, int H, int W,
                                                                   int* __restrict__ contrast_mean) {
    __shared__ unsigned long long part[256];
    const MnkAugJob j = jobs[blockIdx.x];
    int kc = -1;
    for (int k = 0; k < j.jit_n; ++k)
        if (j.jit_op[k] == 4) kc = k;
    if (kc < 0) {
        if (threadIdx.x == 0) contrast_mean[blockIdx.x] = 0;
        return;
    }
    const AugSrc s{pool + j.strip_offset + (size_t)j.frame * j.in_w * j.channels, j.strip_w, j.in_h, j.in_w, j.channels, j.hflip};
    unsigned long long sum = 0;
    for (int p = threadIdx.x; p < H * W; p += 256) {
        double v[3];
        const bool warped = aug_value(j, s, rot_range, blockIdx.x, p / W, p % W, v);
        unsigned char rgb[3];
        aug_ubyte(v, warped, rgb);
        aug_jitter(rgb, j, kc, 0);
        sum += (unsigned long long)aug_luma(rgb);
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) contrast_mean[blockIdx.x] = (int)((double)part[0] / (double)((long)H * W) + 0.5);
}

// one thread per (job, output pixel): grid (ceil(H * W / 256), jobs); (H, W) = the output (crop) size
__global__ void __launch_bounds__(256) frames_augment_kernel(const unsigned char* __restrict__ pool,
                                                             const MnkAugJob* __restrict__ jobs,
                                                             const double* __restrict__ rot_range,
                                                             const int* __restrict__ contrast_mean, int H, int W, int Cout,
                                                             float* __restrict__ out) {
    const MnkAugJob j = jobs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int h = p / W, w = p - h * W;
    const AugSrc s{pool + j.strip_offset + (size_t)j.frame * j.in_w * j.channels, j.strip_w, j.in_h, j.in_w, j.channels, j.hflip};
    double v[3];
    const bool warped = aug_value(j, s, rot_range, blockIdx.y, h, w, v);
    float o3[3];
    if (j.flags & 4) {
        unsigned char rgb[3];
        aug_ubyte(v, warped, rgb);
        aug_jitter(rgb, j, j.jit_n, contrast_mean ? contrast_mean[blockIdx.y] : 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) o3[i] = (float)((double)rgb[i] * (1.0 / 255));          // img_as_float, then float32
    } else {
#pragma unroll
        for (int i = 0; i < 3; ++i) o3[i] = (float)v[i];
    }
    float* o = out + j.out_offset + p;
    for (int ch = 0; ch < Cout; ++ch) o[(size_t)ch * j.chan_stride] = o3[ch < 3 ? ch : 2];
}

}  // namespace

extern "C" {

int mnk_frames_augment(const unsigned char* pool, const MnkAugJob* jobs_device, int njobs, int any_rotation, double* rot_range,
                       int any_contrast, int* contrast_mean, int H, int W, int Cout, float* out, void* stream) {
    MNK_REQUIRE(pool && jobs_device && out && njobs > 0 && njobs <= 65535 && H > 0 && W > 0 && Cout >= 1 && Cout <= 3);
    MNK_REQUIRE((!any_rotation || rot_range) && (!any_contrast || contrast_mean));
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LAYOUT, s, (double)njobs * H * W * (16.0 * 3 + 4.0 * Cout));
    if (any_rotation) hipLaunchKernelGGL(frames_rotated_range_kernel, dim3(njobs), dim3(256), 0, s, pool, jobs_device, rot_range);
    if (any_contrast)
        hipLaunchKernelGGL(frames_contrast_mean_kernel, dim3(njobs), dim3(256), 0, s, pool, jobs_device, rot_range, H, W, contrast_mean);
    hipLaunchKernelGGL(frames_augment_kernel, dim3((H * W + 255) / 256, njobs), dim3(256), 0, s, pool, jobs_device, rot_range,
                       any_contrast ? contrast_mean : (const int*)nullptr, H, W, Cout, out);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

int mnk_frames_gather(const unsigned char* pool, const MnkFrameJob* jobs_device, int njobs, int H, int W, int Cout, float* out,
                      void* stream) {
    MNK_REQUIRE(pool && jobs_device && out && njobs > 0 && njobs <= 65535 && H > 0 && W > 0 && Cout >= 1 && Cout <= 3);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(K_LAYOUT, s, (double)njobs * H * W * (3.0 + 4.0 * Cout));
    hipLaunchKernelGGL(frames_gather_kernel, dim3((H * W + 255) / 256, njobs), dim3(256), 0, s, pool, jobs_device, H, W, Cout, out);
    MNK_LAUNCH_CHECK();
    return MNK_OK;
}

}  // extern "C"
