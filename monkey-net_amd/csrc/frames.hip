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

}  // namespace

extern "C" {

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
