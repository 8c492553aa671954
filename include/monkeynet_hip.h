/*
 * monkeynet_hip.h -- C-ABI of libmonkeynet_hip.so: the MI355X (gfx950) kernels behind Monkey-Net's
 * frame-generation hot path (KPDetector + DenseMotionModule + MotionTransferGenerator, forward/backward).
 *
 * The reference (AliaksandrSiarohin/monkey-net) has no FFI: its boundary is the Python module API of
 * modules/<name>.py, and all device arithmetic lives in torch functionals.  Each entry point below replaces one
 * torch call site (or a tight group of them) of the reference; the citation says which (paths relative to
 * the reference root).  The Python drop-in `monkey-net_amd/modules/` binds these symbols with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated; nothing is allocated inside the library:
 *     scratch comes from the caller (`ws`, sized by the matching *_workspace_floats query);
 *   - activations are NHWC with the time axis folded into N (frame = b*D + d) and a channel stride `ld*`
 *     (floats per pixel, a multiple of 4, >= the logical channel count; pad channels are kept zero);
 *   - `stream` is a hipStream_t passed as void*;
 *   - return value: 0 on success, <0 on error (MNK_E*); mnk_last_error() gives the message of the last
 *     failure on the calling thread.  No global state except the optional profiling recorder.
 */
#ifndef MONKEYNET_HIP_H
#define MONKEYNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNK_OK 0
#define MNK_EINVAL (-1)     /* invalid argument                              */
#define MNK_ELAUNCH (-2)    /* hipLaunchKernel / runtime failure             */
#define MNK_EWORKSPACE (-3) /* workspace too small                           */
#define MNK_ECOMM (-4)      /* RCCL missing or a collective failed           */

int mnk_version(void);
const char* mnk_last_error(void);
/* 1 when the library was built against the real HIP runtime for gfx950, 0 for the CPU test emulator */
int mnk_is_device_build(void);

/* ---- per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ------------------- */
int mnk_prof_enable(int on);
int mnk_prof_reset(void);
int mnk_prof_num_kernels(void);
const char* mnk_prof_kernel_name(int kernel_id);
/* launches, summed milliseconds and summed algorithmic work (FLOP for MFMA kernels, bytes otherwise) */
int mnk_prof_query(int kernel_id, uint64_t* launches, double* total_ms, double* total_work);
/* the work the launches of the group actually issued (2 x multiply-adds of the form that ran: a sub-pixel form of an up-sampled
 * 3x3 convolution issues 4/9 of the algorithmic figure mnk_prof_query reports; equal to it for every other kernel) */
int mnk_prof_query_executed(int kernel_id, double* executed_work);

/* ---- layout --------------------------------------------------------------------------------------------
 * (B,C,D,H,W) <-> folded NHWC.  `step` > 1 applies the nearest down-scaling F.interpolate(scale_factor=
 * (1,1/step,1/step)) of modules/keypoint_detector.py:98-99, dense_motion_module.py:43-44,
 * movement_embedding.py:43-44 (index pick x[..., ::step, ::step]). */
int mnk_ncdhw_to_nhwc(const float* src, float* dst, int B, int C, int D, int H, int W, int step, int ld_dst,
                      void* stream);
int mnk_nhwc_to_ncdhw(const float* src, int ld_src, float* dst, int B, int C, int D, int H, int W, void* stream);
/* dst[r, dst_off + c] = src[r, src_off + c], c < C (torch.cat / slicing: modules/util.py:185, generator.py:73) */
int mnk_copy_channels(const float* src, int ld_src, int src_off, float* dst, int ld_dst, int dst_off, int C,
                      long rows, int accumulate, void* stream);
/* torch.cat([a, b], dim=channel) on acts in one launch, pad channels written (modules/util.py:185 the last decoder stage;
 * discriminator.py:50-52 the key-point heat-maps behind the frame): out[n][p] = [a[n][p][0..ca) | b[n mod Nb][p][0..cb) | 0...],
 * Nb = N, or N / 2 for the batched [generated | real] discriminator pass that embeds the same key points for both halves.
 * Adjoint: ga = [g[..][0..ca) | 0...]; gb (may be NULL) [m][p] = [g[m][p][ca..) + g[m + Nb][p][ca..) + ... | 0...]. */
int mnk_concat2_fwd(const float* a, int ld_a, int ca, const float* b, int ld_b, int cb, int Nb, float* out, int ld_out, int N,
                    long rows_per_frame, void* stream);
int mnk_concat2_bwd(const float* g, int ld_g, int ca, int cb, int Nb, float* ga, int ld_a, float* gb, int ld_b, int N,
                    long rows_per_frame, void* stream);
/* dst[n,h,w,c] = sum of the 2x2 block of src (backward of the nearest x2 up-sampling, modules/util.py:84) */
int mnk_sumpool2x2(const float* src, int ld_src, float* dst, int ld_dst, int N, int Hs, int Ws, int C, void* stream);
/* nearest resize of a channel block into a slice of another tensor (generator.py:72 kp_skips) and its adjoint */
int mnk_resize_nearest(const float* src, int ld_src, int Hs, int Ws, float* dst, int ld_dst, int dst_off, int Hd,
                       int Wd, int N, int C, void* stream);
int mnk_resize_nearest_bwd(const float* ddst, int ld_dst, int dst_off, int Hd, int Wd, float* dsrc, int ld_src,
                           int Hs, int Ws, int N, int C, void* stream);
/* the same, added to dsrc (several resized copies of one tensor: generator.py:72 resizes the key-point embedding once per
 * skip) */
int mnk_resize_nearest_bwd_accumulate(const float* ddst, int ld_dst, int dst_off, int Hd, int Wd, float* dsrc, int ld_src,
                                      int Hs, int Ws, int N, int C, void* stream);
/* the same with bilinear, align_corners=False weights (interpolation_mode='trilinear' with unchanged depth, vox configs);
 * the adjoint is ADDED to dsrc (zero it first, or let several resized copies of one tensor add up) by a gather per source
 * texel in destination-pixel order: deterministic, no atomics */
int mnk_resize_bilinear(const float* src, int ld_src, int Hs, int Ws, float* dst, int ld_dst, int dst_off, int Hd,
                        int Wd, int N, int C, void* stream);
int mnk_resize_bilinear_bwd(const float* ddst, int ld_dst, int dst_off, int Hd, int Wd, float* dsrc, int ld_src,
                            int Hs, int Ws, int N, int C, void* stream);

/* ---- BatchNorm (sync_batchnorm/batchnorm.py:48-78,113-125; F.batch_norm semantics: (var+eps)^-1/2) -------
 * statistics: sums[0..C) = sum x, sums[C..2C) = sum x^2 over `rows` pixels.  The caller may all-reduce
 * `sums` across ranks (RCCL) before mnk_bn_finalize -- that is the SyncBN exchange. */
size_t mnk_bn_workspace_floats(long rows, int ld);
int mnk_bn_stats(const float* x, int ld, long rows, int C, float* sums, float* ws, size_t ws_floats, void* stream);
/* second stage only: sums[which*C + c] = sum_rb partial[(rb*2 + which)*ld + c]  (partials from a conv epilogue) */
int mnk_bn_stats_finish(const float* partial, int row_blocks, int ld, int C, float* sums, void* stream);
/* mean = sum/count, var = sumsq/count - mean^2, invstd = (var+eps)^-1/2, scale = gamma*invstd;
 * running_mean/var updated with momentum and the unbiased variance (batchnorm.py:119-123) */
int mnk_bn_finalize(const float* sums, double count, const float* gamma, float* running_mean, float* running_var,
                    float momentum, float eps, int C, int update_running, float* mean, float* invstd, float* scale,
                    void* stream);
/* Single-process form of (mnk_bn_stats | mnk_bn_stats_finish) + mnk_bn_finalize with the second stage and the
 * finalisation in ONE launch: statistics either from `pre_partial` (pre_row_blocks conv-epilogue partials) or from a
 * pass over x (workspace as for mnk_bn_stats).  `sums` (optional, 2C) receives what mnk_bn_stats would have written.
 * Not for SyncBN over several ranks (the all-reduce sits between the two stages there). */
int mnk_bn_stats_finalize(const float* x, int ld, long rows, int C, const float* pre_partial, int pre_row_blocks,
                          double count, const float* gamma, float* running_mean, float* running_var, float momentum,
                          float eps, int update_running, float* sums, float* mean, float* invstd, float* scale, float* ws,
                          size_t ws_floats, void* stream);
int mnk_bn_eval_coeffs(const float* gamma, const float* running_mean, const float* running_var, float eps, int C,
                       float* mean, float* invstd, float* scale, void* stream);
/* z = [avgpool2x2] [relu] ((y-mean)*scale + beta)   (modules/util.py:56-57,62,81,87,100-101,106-107) */
int mnk_bn_act_fwd(const float* y, int ld_y, const float* mean, const float* scale, const float* beta, float* z,
                   int ld_z, int z_off, int N, int H, int W, int C, int relu, int pool, void* stream);
/* mnk_bn_finalize + mnk_bn_act_fwd in one launch: `sums` = finished [sum x][sum x^2] of the whole batch (the SyncBN path: they
 * come out of the cross-rank all-reduce, batchnorm.py:95-111, so the single-process fusion of the second stage does not apply);
 * every block derives the constants of its channels, the first row block writes mean / invstd / scale for the backward pass
 * and updates the running statistics (batchnorm.py:113-125).  Same arithmetic as the two-launch form. */
int mnk_bn_act_fwd_sums(const float* y, int ld_y, const float* sums, double count, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, int update_running, float* mean,
                        float* invstd, float* scale, float* z, int ld_z, int z_off, int N, int H, int W, int C, int relu, int pool,
                        void* stream);
/* backward, pass 1: sums[0..C) = sum g, sums[C..2C) = sum g*xhat with g = dL/d(BN output), xhat = (y-mean)*invstd.
 * These are also dbeta and dgamma. */
int mnk_bn_act_bwd_stats(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                         const float* invstd, const float* scale, const float* beta, int N, int H, int W, int C,
                         int relu, int pool, float* sums, float* ws, size_t ws_floats, void* stream);
/* backward, pass 2: dy = scale*(g - sum_g/count - xhat*sum_gx/count) (training) or g*scale (eval).
 * `sums`/`count` may be the all-reduced (SyncBN) values. */
int mnk_bn_act_bwd_apply(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                         const float* invstd, const float* scale, const float* beta, const float* sums, double count,
                         int training, float* dy, int ld_dy, int N, int H, int W, int C, int relu, int pool,
                         void* stream);
/* the same pass, additionally dy_sums[0..C) = column sums of the written dy: the bias gradient of the convolution
 * in front of the norm layer (util.py:54-56,80-81,99-100), which otherwise costs that convolution's backward a pass
 * over dy.  Workspace: mnk_bn_workspace_floats(N*H*W, round_up(C,4)). */
int mnk_bn_act_bwd_apply_colsum(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                                const float* invstd, const float* scale, const float* beta, const float* sums,
                                double count, int training, float* dy, int ld_dy, int N, int H, int W, int C, int relu,
                                int pool, float* dy_sums, float* ws, size_t ws_floats, void* stream);
/* the same with a second gradient of the normalised tensor added in the pass (`addend`, (N,H,W,ld_add) or NULL): dy = BatchNorm
 * backward + addend, dy_sums over the sum.  A residual block's first norm layer (util.py:58-67: `out += x`) receives the
 * gradient of the skip path this way instead of through a separate accumulation pass over both gradients. */
int mnk_bn_act_bwd_apply_add_colsum(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                                    const float* invstd, const float* scale, const float* beta, const float* sums,
                                    double count, int training, const float* addend, int ld_add, float* dy, int ld_dy, int N,
                                    int H, int W, int C, int relu, int pool, float* dy_sums, float* ws, size_t ws_floats,
                                    void* stream);

/* ---- small layers (N*H*W <= mnk_bn_small_rows() pixel rows, 512 by default: the 2x2 / 4x4 levels of the hourglasses): the whole
 * training-mode BatchNorm (+ReLU, +2x2 pool) of one rank in ONE launch per direction instead of four -- a block owns a tile of
 * channels over all rows, so the column sums never leave it.  Forward: optionally sums the split-K partials `ws`
 * ([split][phase][M][ldw], what mnk_conv3x3_fwd / mnk_conv3x3_up_fwd leave behind under MNK_CONV_DEFER_SPLITK) + bias into y
 * first; then statistics, mean / inv-std / scale, running statistics (batchnorm.py:113-125) and the apply pass
 * (util.py:56-57,81-87,100-107).  Backward: sums = [dbeta | dgamma], dy.  Several ranks: mnk_bn_small_fwd_sync / _bwd_sync. */
int mnk_bn_small_rows(void);
int mnk_bn_small_fwd(const float* ws, int splits, int ldw, int phases, const float* bias, float* y, int ld_y, int N, int H, int W,
                     int C, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, float* mean, float* invstd, float* scale, float* z, int ld_z, int relu, int pool, void* stream);
/* (round 6) an EVALUATION-mode norm layer (running statistics: mean / scale from mnk_bn_eval_coeffs) straight from the split-K
 * partials of the convolution in front (MNK_CONV_DEFER_SPLITK, `ws` = [split][phase][M][ldw]): partial sums + bias, affine, ReLU,
 * 2x2 average pool -> z in ONE launch; y is never written.  (H, W) = the convolution's output size (the up-sampled size when
 * phases == 4); ld_z == ldw == round_up(C, 4).  Bit-identical to the split reduction followed by mnk_bn_act_fwd.  What the
 * per-frame loops of reconstruction.py:45-62 / transfer.py run at batch 1 (sync_batchnorm/batchnorm.py:57-59 in eval mode,
 * util.py:56-57,81-87,100-107). */
int mnk_bn_eval_split_fwd(const float* ws, int splits, int ldw, int phases, const float* bias, const float* mean, const float* scale,
                          const float* beta, float* z, int ld_z, int N, int H, int W, int C, int relu, int pool, void* stream);
int mnk_bn_small_bwd(const float* y, int ld_y, const float* dz, int ld_dz, const float* mean, const float* invstd,
                     const float* scale, const float* beta, double count, int N, int H, int W, int C, int relu, int pool,
                     float* sums, float* dy, int ld_dy, void* stream);

/* ---- general normalisation forms: `frames` / per_frame = statistics per frame (nn.InstanceNorm3d of the discriminator,
 * modules/discriminator.py:19-22,29-30: sums [2][frames*C], mean/invstd/scale [frames*C], beta [C]); `slope` selects
 * the fused activation: < 0 none, 0 ReLU, > 0 LeakyReLU(slope) (discriminator.py:31).  With odd H / W the average pool
 * drops the last row / column like F.avg_pool3d (discriminator.py:32).  mnk_bn_* above are the frames = 1 forms. */
size_t mnk_norm_workspace_floats(long rows_per_frame, int frames, int ld);
int mnk_norm_stats(const float* x, int ld, long rows_per_frame, int frames, int C, float* sums, float* ws,
                   size_t ws_floats, void* stream);
int mnk_norm_finalize(const float* sums, double count, const float* gamma, float* running_mean, float* running_var,
                      float momentum, float eps, int C, int frames, int update_running, float* mean, float* invstd,
                      float* scale, void* stream);
int mnk_norm_act_fwd(const float* y, int ld_y, const float* mean, const float* scale, const float* beta, int per_frame,
                     float* z, int ld_z, int z_off, int N, int H, int W, int C, float slope, int pool, void* stream);
int mnk_norm_act_bwd_stats(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                           const float* invstd, const float* scale, const float* beta, int per_frame, int N, int H,
                           int W, int C, float slope, int pool, float* sums, float* ws, size_t ws_floats,
                           void* stream);
int mnk_norm_act_bwd_apply(const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                           const float* invstd, const float* scale, const float* beta, int per_frame, const float* sums,
                           double count, int training, float* dy, int ld_dy, int N, int H, int W, int C, float slope,
                           int pool, void* stream);

/* ---- 3x3 convolution, pad 1, stride 1 (nn.Conv3d (1,3,3): modules/util.py:52-55,79,98,176) -------------
 * implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32).  The input is the channel concatenation of up to two
 * NHWC sources (torch.cat of modules/util.py:185 never materialised); `ups` = 1 reads both sources through the
 * nearest x2 up-sampling of modules/util.py:84.  Weights are first re-packed to [Cout][chunk][tap][16]: the
 * K axis is walked in 16-channel chunks (CXp = C rounded up to 16, zero filled, source 0 then source 1) with the
 * nine taps of a chunk adjacent, so the taps re-read the same pixel rows from L1.  (Reading the (Cout,Cin,3,3)
 * parameter in place was measured 20 % slower on the MI355X: 36-byte strided rows thrash the 32 KB L1.) */
size_t mnk_conv3x3_packed_floats(int Cout, int C0, int C1);
int mnk_conv3x3_pack_fwd(const float* w, float* wp, int Cout, int C0, int C1, void* stream);
/* dgrad weights for input channels [c_start, c_start+c_count): wp[ci][chunk][tap][16] = w[16*chunk+k][c_start+ci][8-tap] */
int mnk_conv3x3_pack_dgrad(const float* w, float* wp, int Cout, int Cin_total, int c_start, int c_count, void* stream);
/* One launch for a training forward: the forward pack and the dgrad packs of source 0 (channels [0, C0), size
 * mnk_conv3x3_packed_floats(C0, Cout, 0)) and of source 1 ([C0, C0+C1), size ..(C1, Cout, 0)); wp_d0 / wp_d1 may be
 * NULL.  The parameter changes every optimiser step (train.py:118,132), so nothing packed can be kept across steps. */
int mnk_conv3x3_pack_all(const float* w, float* wp_fwd, float* wp_d0, float* wp_d1, int Cout, int C0, int C1,
                         void* stream);
/* The same for EVERY 3x3 layer of a model in one launch (a training iteration re-packs ~40 layers; one launch each is
 * launch-bound): `descs_device` is a table of n descriptors in device memory, sorted by tile_begin, where layer i owns
 * the blocks [tile_begin, tile_begin + tiles) with tiles = ((ceil16(C0) + ceil16(C1)) / 16) * ceil(Cout / 16);
 * total_tiles = sum of tiles.  Pointers as for mnk_conv3x3_pack_all (wp_d0 / wp_d1 may be NULL). */
typedef struct MnkPackDesc {
    const float* w;
    float* wp_fwd;
    float* wp_d0;
    float* wp_d1;
    int Cout, C0, C1, tile_begin;
    int flags;      /* bit 0: an up-sampled convolution -- emit the packs of its sub-pixel forms (mnk_conv3x3_up_*) instead */
    int reserved;
} MnkPackDesc;
int mnk_conv3x3_pack_multi(const MnkPackDesc* descs_device, int n, int total_tiles, void* stream);
size_t mnk_conv3x3_workspace_floats(int N, int H, int W, int C0, int C1, int Cout);
/* `stats_partial` (optional, mnk_conv3x3_stats_floats floats; only when that query is > 0): per-block column sums / sums
 * of squares of y -- the BatchNorm statistics of the following norm layer -- from the kernel epilogue of an unsplit launch or
 * from the reduction of a split-K launch's partials (not with MNK_CONV_DEFER_SPLITK), to be finished by
 * mnk_bn_stats_finish(stats_partial, stats_floats / (2*ld_y), ld_y, Cout, sums); needs ld_y == round_up(Cout, 4). */
size_t mnk_conv3x3_stats_floats(int N, int H, int W, int C0, int C1, int Cout);
/* `flags`: MNK_CONV_UPSAMPLED (= 1, the former `ups` argument: both sources are read through the nearest x2 up-sampling)
 * | MNK_CONV_CLEAN_PADS (= 2): the caller vouches that the pad channels [C, ld) of the sources hold zeros (every
 * activation this library writes does).  Without it the gather masks them (a source with garbage / NaN pads is fine,
 * ~10 % slower); with it the 3x3 kernels use raw buffer loads whose out-of-range lanes read zero. */
#define MNK_CONV_UPSAMPLED 1
#define MNK_CONV_CLEAN_PADS 2
/* (= 4, forward only) a split-K launch leaves its partials in `ws` -- [split][phase][M][round_up(Cout, 4)], bias NOT added,
 * mnk_conv3x3_splits / mnk_conv3x3_up_splits of them -- and the caller sums them (mnk_bn_small_fwd does, together with the
 * normalisation that follows); y is not written */
#define MNK_CONV_DEFER_SPLITK 4
int mnk_conv3x3_splits(int N, int H, int W, int C0, int C1, int Cout);
int mnk_conv3x3_up_splits(int N, int H, int W, int C0, int C1, int Cout);
int mnk_conv3x3_fwd(const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, int flags, const float* wp,
                    const float* bias, const float* residual, int ld_res, float* y, int ld_y, int N, int H, int W,
                    int Cout, float* ws, size_t ws_floats, float* stats_partial, void* stream);
/* dw[co][c_start+ci][ky][kx] = sum_pixels dy[p][co] * x[p+tap][ci]; x is one source (C channels) */
size_t mnk_conv3x3_wgrad_workspace_floats(int N, int H, int W, int C, int Cout);
int mnk_conv3x3_wgrad(const float* x, int ld_x, int C, int flags, const float* dy, int ld_dy, int Cout, float* dw,
                      int Cin_total, int c_start, int N, int H, int W, float* ws, size_t ws_floats, void* stream);

/* ---- sub-pixel forms of UpBlock3D: [nearest x2 up-sampling -> 3x3 / pad 1] (modules/util.py:83-85) ------------------------
 * Output pixel (2i + a, 2j + b) of the up-sampled convolution only sees the 2 x 2 low-resolution neighbourhood
 * (rows i + a - 1, i + a; columns j + b - 1, j + b), with weights that are sums of the 3x3 taps falling on the same input
 * pixel.  So the forward is four 2x2 convolutions on the LOW-resolution input (one per phase (a, b), one launch) and the
 * data gradient w.r.t. the low-resolution input is ONE 4x4 / stride 2 / pad 1 convolution over dy -- 4 instead of 9
 * multiply-adds per output and channel pair in both directions, no up-sampled view, no mnk_sumpool2x2 pass.  Same result
 * as mnk_conv3x3_fwd with MNK_CONV_UPSAMPLED up to the rounding of the pre-summed weights (<= 4 terms).  (H, W) = the LOW
 * resolution; y / dy are (N, 2H, 2W) tensors; sources need clean pad channels.
 *   wp_up       [4 phases][Cout][chunk][4 taps][16]           mnk_conv3x3_up_packed_floats
 *   wp_up_dgrad [C][chunk over Cout][16 taps][16]              mnk_conv3x3_up_dgrad_packed_floats  (one source's channels) */
size_t mnk_conv3x3_up_packed_floats(int Cout, int C0, int C1);
size_t mnk_conv3x3_up_dgrad_packed_floats(int Cout, int c_count);
int mnk_conv3x3_up_pack_fwd(const float* w, float* wp, int Cout, int C0, int C1, void* stream);
int mnk_conv3x3_up_pack_dgrad(const float* w, float* wp, int Cout, int Cin_total, int c_start, int c_count, void* stream);
size_t mnk_conv3x3_up_workspace_floats(int N, int H, int W, int C0, int C1, int Cout);
size_t mnk_conv3x3_up_stats_floats(int N, int H, int W, int C0, int C1, int Cout);
int mnk_conv3x3_up_fwd(const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, int flags, const float* wp_up,
                       const float* bias, float* y, int ld_y, int N, int H, int W, int Cout, float* ws, size_t ws_floats,
                       float* stats_partial, void* stream);     /* flags: 0 or MNK_CONV_DEFER_SPLITK */
size_t mnk_conv3x3_up_dgrad_workspace_floats(int N, int H, int W, int Cout, int C);
int mnk_conv3x3_up_dgrad(const float* dy, int ld_dy, int Cout, const float* wp_up_dgrad, float* dx, int ld_dx, int N, int H,
                         int W, int C, float* ws, size_t ws_floats, void* stream);
/* Data-gradient launches that ALSO leave the backward statistics of the BatchNorm (+ activation) layer whose output they
 * differentiate (round 4; sync_batchnorm/batchnorm.py:48-78 backward through modules/util.py:56-62,81-87): dx of the
 * convolution IS dz of the norm layer in front when that layer's output has no other consumer, and the GEMM epilogue holds the
 * dz tile in registers, so stats_partial receives per-block column sums of
 *     g = act'((bn_y - mean) * scale + beta) * dz      and      g * (bn_y - mean) * invstd
 * ([blocks][2][ld_dx], blocks * 2 * ld_dx = mnk_conv3x3_stats_floats(N, H, W, Cout, 0, C) resp.
 * mnk_conv3x3_up_dgrad_stats_floats; 0: this shape's launch plan has no statistics form) -- what mnk_bn_act_bwd_stats makes
 * in a pass of its own over bn_y and dz; mnk_bn_stats_finish sums them.  slope: < 0 none, 0 ReLU, > 0 LeakyReLU.  The norm
 * layer must not pool (dz and bn_y have one geometry). */
int mnk_conv3x3_dgrad_bnstats(const float* dy, int ld_dy, int Cout, const float* wp_dgrad, const float* residual, int ld_res,
                              float* dx, int ld_dx, int N, int H, int W, int C, float* ws, size_t ws_floats, float* stats_partial,
                              const float* bn_y, int ld_bny, const float* bn_mean, const float* bn_invstd, const float* bn_scale,
                              const float* bn_beta, float slope, void* stream);
size_t mnk_conv3x3_up_dgrad_stats_floats(int N, int H, int W, int Cout, int C);
int mnk_conv3x3_up_dgrad_bnstats(const float* dy, int ld_dy, int Cout, const float* wp_up_dgrad, float* dx, int ld_dx, int N, int H,
                                 int W, int C, float* ws, size_t ws_floats, float* stats_partial, const float* bn_y, int ld_bny,
                                 const float* bn_mean, const float* bn_invstd, const float* bn_scale, const float* bn_beta,
                                 float slope, void* stream);

/* ---- general K x K form of the same kernels (stride 1).  Used for the discriminator's nn.Conv3d((1,4,4)) without
 * padding (modules/discriminator.py:17-18; SURVEY.md section 8f row 1, the first "next" component): forward
 * kh = kw = 4, pad = 0; its data gradient is the same kernel on dy with pad = 3 and the flipped pack.
 * Ho = Hi + 2*pad - kh + 1 (same for W); `ups` views the input through the nearest x2 up-sampling (Hi, Wi are the
 * up-sampled sizes).  Packed weights: [Cout][chunk][tap][16], tap = ky*kw + kx. */
size_t mnk_conv2d_packed_floats(int Cout, int C0, int C1, int ntaps);
int mnk_conv2d_pack_fwd(const float* w, float* wp, int Cout, int C0, int C1, int ntaps, void* stream);
int mnk_conv2d_pack_all(const float* w, float* wp_fwd, float* wp_d0, float* wp_d1, int Cout, int C0, int C1, int ntaps,
                        void* stream);
int mnk_conv2d_pack_dgrad(const float* w, float* wp, int Cout, int Cin_total, int c_start, int c_count, int ntaps,
                          void* stream);
size_t mnk_conv2d_workspace_floats(int N, int Ho, int Wo, int C0, int C1, int Cout, int ntaps);
size_t mnk_conv2d_stats_floats(int N, int Ho, int Wo, int C0, int C1, int Cout, int ntaps);
int mnk_conv2d_fwd(const float* x0, int ld0, int C0, const float* x1, int ld1, int C1, int flags, int Hi, int Wi, int kh,
                   int kw, int pad, const float* wp, const float* bias, const float* residual, int ld_res, float* y,
                   int ld_y, int N, int Ho, int Wo, int Cout, float* ws, size_t ws_floats, float* stats_partial,
                   void* stream);
size_t mnk_conv2d_wgrad_workspace_floats(int N, int Ho, int Wo, int C, int Cout, int kh, int kw, int pad);
int mnk_conv2d_wgrad(const float* x, int ld_x, int C, int flags, int Hi, int Wi, int kh, int kw, int pad, const float* dy,
                     int ld_dy, int Cout, float* dw, int Cin_total, int c_start, int N, int Ho, int Wo, float* ws,
                     size_t ws_floats, void* stream);

/* ---- deferred split reductions (new; the reference has no counterpart: torch's conv backward reduces inside cuDNN).
 * A backward pass launches ~60 weight-gradient GEMMs whose pixel-split partials each needed 1-2 small reduction launches
 * (latency bound: ~100 launches of 5-10 us per iteration).  With MNK_WGRAD_DEFER (flags bit 2) mnk_conv2d_wgrad /
 * mnk_conv3x3_wgrad only run the GEMM and leave its partials in `ws`; mnk_conv2d_wgrad_plan says in which layout and how
 * many floats (`ws` must then be a buffer that lives until the reduction); mnk_wgrad_reduce_multi reduces the partials
 * of ANY number of layers in one launch, each into its slice dw[co][c_start + ci][tap] of a (Cout, Cin_total, kh, kw)
 * parameter gradient.  Descriptor table in device memory, sorted by block_begin; layer i owns
 * mnk_wgrad_reduce_blocks(splits, Cout, C) blocks from block_begin; total_blocks = their sum.  Deterministic order. */
#define MNK_WGRAD_DEFER 4
typedef struct MnkWgradPlan {
    int layout;          /* 0: tap-major partials [split][tap][Cout][C]; 1: parameter-major [split][Cout][C * ntaps];
                          * 2: tap-major with the 16 pseudo taps of the sub-pixel form (mnk_conv2d_wgrad_plan2) */
    int splits;          /* 0: the GEMM writes dw itself -- nothing to reduce */
    size_t part_floats;  /* floats of `ws` the GEMM fills under MNK_WGRAD_DEFER */
} MnkWgradPlan;
int mnk_conv2d_wgrad_plan(int N, int Ho, int Wo, int C, int Cout, int kh, int kw, int pad, int ld_x, MnkWgradPlan* plan);
/* the same with the `flags` the GEMM will be called with: an up-sampled 3x3 layer with clean sources (MNK_CONV_UPSAMPLED |
 * MNK_CONV_CLEAN_PADS) runs in its sub-pixel form -- 16 pseudo taps over the low-resolution pixels (4/9 of the multiply-adds)
 * -- and leaves layout 2: [split][16][Cout][C], folded into the nine kernel taps by mnk_wgrad_reduce_multi */
int mnk_conv2d_wgrad_plan2(int N, int Ho, int Wo, int C, int Cout, int kh, int kw, int pad, int ld_x, int flags,
                           MnkWgradPlan* plan);
/* workspace of mnk_conv3x3_wgrad for an up-sampled layer (either form; Ho, Wo = the up-sampled size) */
size_t mnk_conv3x3_up_wgrad_workspace_floats(int N, int Ho, int Wo, int C, int Cout);
typedef struct MnkWgradReduceDesc {
    const float* part;
    float* dw;           /* start of the (Cout, Cin_total, ntaps) gradient */
    int layout, splits, ntaps, Cout, C, Cin_total, c_start;
    int accumulate;      /* 1: add to dw (a second contribution to the same parameter) */
    int block_begin;
    int reserved;
} MnkWgradReduceDesc;
int mnk_wgrad_reduce_blocks(int splits, int Cout, int C);
/* the measured defaults of the launch plans (tile counts, split targets, rows per thread: `tuning_knob("name", ...)` in the
 * kernel sources, e.g. "split_tiles", "wgroup_chunk", "wtap_target", "bn_rpt") are not environment switches: a tuning script
 * sets one by name here; an A/B visit may pass them all in ONE environment variable, MNK_TUNING="name=value,name=value" */
int mnk_set_tuning(const char* name, int value);
/* the live value (whatever set it: the default, MNK_TUNING or mnk_set_tuning) -- host code that must follow a tuning value
 * (mnk.ops.subpixel: "up_subpixel") reads it here instead of parsing the environment */
int mnk_get_tuning(const char* name, int* value);
/* the forward / data-gradient GEMM's launch plan is a rule (tile by channel count, 64-row tiles and split-K for few-tile
 * layers) overridden, for the benchmark configurations' layer shapes, by plans measured on the MI355X (csrc/plan_table.h;
 * "plan_table" = 0 ignores it).  "force_bm" / "force_bn" / "force_splits" (mnk_set_tuning, 0 = off) force a plan for the
 * sweep that makes the table (tools/plan_tune.py); mnk_last_plan reports the plan of the last such launch or size query:
 * {M, Cout, chunks, taps, phases, bm, bn, splits}. */
int mnk_last_plan(long* out8);
int mnk_wgrad_reduce_multi(const MnkWgradReduceDesc* descs_device, int n, int total_blocks, void* stream);

/* ---- grouped weight gradients (new): the tap-major GEMMs of MANY layers in one launch per tile shape.  Launched one by
 * one, each layer must be cut into 2..86 pixel splits to fill 256 CUs and its split partials cross HBM twice (1.3 GB per
 * iteration on BASELINE configs[1]); launched together, the tiles of all layers fill the chip and a layer is only split
 * where its pixel range is long (~1024-pixel chunks; 0.2 GB).  Protocol: fill the geometry of n jobs (one per layer and
 * source; pointers may still be NULL) -> mnk_wgrad_grouped_plan sets `variant` (< 0: not a tap-major shape -- run that
 * layer through mnk_conv2d_wgrad), `splits` and `part_floats` -> give every job its operands and a `part` buffer ->
 * mnk_wgrad_grouped_build serialises the launch tables into HOST memory (mnk_wgrad_grouped_table_bytes(n)); copy them
 * to the device -> mnk_wgrad_grouped_launch(device copy, host copy).  The partials are tap-major [split][tap][Cout][C]:
 * reduce them with mnk_wgrad_reduce_multi (layout 0, `splits` as planned; layout 2 for variants with variant % 4 == 3: the
 * sub-pixel form).  Variants >= 16: narrow 3x3 layers (C, Cout <= 64) on the nine-tap 16x16-MFMA kernel, grouped the same
 * way (~128 blocks per layer instead of the 512 a layer needs alone: the eight 45 -> 45 convolutions of the refinement
 * stack write 9 instead of 37 MB of partials each); their partials are tap-major too (layout 0). */
typedef struct MnkWgradJob {
    const float* x;
    const float* dy;
    float* part;
    size_t part_floats;                      /* out (plan) */
    int ld_x, C, flags, ld_dy, Cout;         /* flags: MNK_CONV_UPSAMPLED | MNK_CONV_CLEAN_PADS */
    int N, Ho, Wo, Hi, Wi, kh, kw, pad;
    int variant, splits;                     /* out (plan) */
    int reserved;
} MnkWgradJob;
int mnk_wgrad_grouped_plan(MnkWgradJob* jobs, int n);
size_t mnk_wgrad_grouped_table_bytes(int n);
int mnk_wgrad_grouped_build(const MnkWgradJob* jobs, int n, void* host_table, size_t table_bytes);
int mnk_wgrad_grouped_launch(const void* device_table, const void* host_table, void* stream);
/* host -> device upload of a small table (launch tables, descriptor arrays) as kernel arguments: 3.5 KB per launch, no
 * page-locked staging, an ordinary kernel node under hipGraph capture (the bytes are frozen at capture time).  `device`
 * must be 16-byte aligned and hold bytes rounded up to 16. */
int mnk_table_upload(const void* host, void* device, size_t bytes, void* stream);

/* ---- optimiser: torch.optim.Adam(lr, betas=(0.5, 0.999)) of train.py:81-83,118-136 for EVERY
 * tensor of a model in one launch.  Formula of torch/optim/adam.py::_single_tensor_adam (no amsgrad / weight decay) in
 * fp32.  `hyper` = 10 floats in DEVICE memory [lr, beta1, beta2, eps, lr / (1 - beta1^t), sqrt(1 - beta2^t), grad_scale, t,
 * 1 - beta1, 1 - beta2] (the last two rounded from the host's doubles, as torch does): a captured hipGraph sees a
 * learning-rate change (MultiStepLR, train.py:91-96) without re-capture; mnk_adam_tick does t += 1 and refreshes slots
 * 4 and 5 (call it once before the launches of an iteration); grad_scale multiplies g first (1 / world size after a sum
 * all-reduce).  A descriptor is a plain range of n floats, or -- wp_fwd != NULL -- a
 * (Cout, C0 + C1, 1, 3, 3) convolution weight whose packed forward / data-gradient layouts (mnk_conv3x3_pack_multi's)
 * are written from the updated values in the same pass.  Table in device memory sorted by block_begin; a descriptor owns
 * mnk_adam_blocks(...) blocks. */
typedef struct MnkAdamDesc {
    float* p;
    const float* g;
    float* m;
    float* v;
    long n;
    float* wp_fwd;
    float* wp_d0;
    float* wp_d1;
    int Cout, C0, C1, block_begin;
    int flags;      /* bit 0: an up-sampled convolution -- emit the packs of its sub-pixel forms (mnk_conv3x3_up_*);
                       bit 1 / bit 2: gt0 / gt1 hold 16 pseudo taps of the sub-pixel weight-gradient form (folded here) */
    int reserved;
    /* optional: the gradient of source 0 / source 1 of a convolution weight taken straight from the tap-major partials of the
     * grouped weight-gradient GEMMs, part[split][tap][Cout][C_source] (mnk_wgrad_grouped_*), summed over gt_splits* <= 3 splits in
     * order -- the bits mnk_wgrad_reduce_multi would have left in `g`, without its pass over them.  NULL: `g` is read. */
    const float* gt0;
    const float* gt1;
    int gt_splits0, gt_splits1;
} MnkAdamDesc;
int mnk_adam_blocks(long n, int Cout, int C0, int C1, int packed);
int mnk_adam_tick(float* hyper, void* stream);
int mnk_adam_multi(const MnkAdamDesc* descs_device, int n, int total_blocks, const float* hyper, void* stream);

/* ---- data-parallel collectives: RCCL all-reduce over xGMI on the CALLER's stream -- in order with
 * the kernels, capturable into the iteration's hipGraph, no hand-over to a communication stream.  One process per GPU.
 * mnk_comm_unique_id: rank 0 makes a 128-byte id (host memory) and the host layer gives it to every rank;
 * mnk_comm_init: every rank (its GPU current) joins; `comm` is the only state.  mnk_comm_available() == 0 when
 * librccl.so.1 cannot be loaded (and always in the CPU emulator build): the host layer then stays on its own transport.
 *   mnk_allreduce_bnstats: the SyncBN exchange -- sums of one norm layer (2C floats) summed over ranks in place; replaces
 *     reduce-to-master + broadcast of sync_batchnorm/batchnorm.py:95-111 (every rank finalises identically afterwards);
 *   mnk_allreduce_grads: a flat gradient buffer summed (average = 0) or averaged (1) over ranks in place, in chunks of
 *     chunk_floats grouped into one RCCL launch; replaces DataParallel's reduce-add + re-broadcast (train.py:104-105). */
int mnk_comm_available(void);
int mnk_comm_unique_id(void* id128);
int mnk_comm_init(const void* id128, int rank, int world, void** comm_out);
int mnk_comm_destroy(void* comm);
int mnk_allreduce_bnstats(void* comm, float* sums, long n, void* stream);
/* the same into a second buffer (the local sums stay: they are this rank's share of the scale / shift gradients) */
int mnk_allreduce_bnstats_to(void* comm, const float* sums, float* out, long n, void* stream);
int mnk_allreduce_grads(void* comm, float* grads, long n, int average, long chunk_floats, void* stream);

/* ---- SyncBN exchange of one node without a collective library (round 4; SURVEY.md section 5, section 7 hard part 6) -------------
 * Replaces sync_batchnorm/batchnorm.py:95-111 + comm.py:102-133 (reduce to the master replica + broadcast through queues) for
 * the <= 8 KB [sum, sum of squares] / [sum g, sum g xhat] vectors of a norm layer: every rank owns a mailbox in its HBM
 * (mnk_p2p_create), exports it as a 64-byte IPC handle (mnk_p2p_export; the host layer gathers the handles of all ranks of the
 * node, in rank order) and maps the others' (mnk_p2p_connect).  mnk_p2p_allreduce is ONE 256-thread kernel on `stream`: push
 * the rank's n <= mnk_p2p_max_floats() floats into its row of every mailbox over xGMI, flag it with the exchange's sequence
 * number (kept in device memory and advanced by the kernel: capturable), wait for every rank's flag in the own mailbox, add
 * the rows in rank order -> out (in == out allowed): every rank gets the same bits.  The wait gives up after timeout_ms
 * (a dead peer must not hang the GPU) and raises the handle's error word, read with mnk_p2p_error (0 = none; it synchronises). */
/* BatchNorm statistics with that exchange INSIDE their second stage (csrc/batchnorm.hip colsum2_final_sync_kernel): the wavefront
 * that finishes a column sum pushes it to every rank and adds the contributions in rank order -- second stage + all-reduce in one
 * launch, so a norm layer costs a data-parallel rank as many launches as a single GPU.  p2p: a connected mnk_p2p handle;
 * sums_local (optional, 2C): this rank's own sums; sums_global (2C): the sums over the ranks, the same bits on every rank.
 * _finish_sync: from per-block partials ([row_blocks][2][ld], a convolution's epilogue); _stats_sync: a pass over x + second stage;
 * _bwd_stats_sync: mnk_bn_act_bwd_stats's pass over (y, dz) + second stage. */
int mnk_bn_stats_finish_sync(void* p2p, const float* partial, int row_blocks, int ld, int C, float* sums_local, float* sums_global,
                             int timeout_ms, void* stream);
int mnk_bn_stats_sync(void* p2p, const float* x, int ld, long rows, int C, float* sums_local, float* sums_global, float* ws,
                      size_t ws_floats, int timeout_ms, void* stream);
int mnk_bn_act_bwd_stats_sync(void* p2p, const float* y, int ld_y, const float* dz, int ld_dz, int dz_off, const float* mean,
                              const float* invstd, const float* scale, const float* beta, int N, int H, int W, int C, int relu,
                              int pool, float* sums_local, float* sums_global, float* ws, size_t ws_floats, int timeout_ms,
                              void* stream);
/* mnk_bn_small_bwd for one rank of a data-parallel run: the one-launch backward of a small norm layer (<= mnk_bn_small_rows()
 * pixel rows per rank) with the exchange of its eight sums per channel quad inside -- statistics, exchange and apply in one
 * launch, as on a single GPU.  count_all_ranks = pixel rows of ALL ranks; sums_local (2C) = this rank's own [sum g, sum g xhat]
 * (its dbeta / dgamma contributions, averaged later with all other gradients: batchnorm.py:48-78 under DataParallel). */
int mnk_bn_small_bwd_sync(void* p2p, const float* y, int ld_y, const float* dz, int ld_dz, const float* mean, const float* invstd,
                          const float* scale, const float* beta, double count_all_ranks, int N, int H, int W, int C, int relu,
                          int pool, float* sums_local, float* dy, int ld_dy, int timeout_ms, void* stream);
/* mnk_bn_small_fwd for one rank of a data-parallel run (round 5, the forward twin): split-K reduction + bias, this rank's column
 * sums, their exchange (eight per channel quad, one round trip), mean / inv-std / scale and the running statistics from the sums
 * over ALL ranks (count = N*H*W * world), and the apply pass -- one launch.  Replaces sync_batchnorm/batchnorm.py:55-78 with its
 * reduce-to-master + broadcast (:95-111) for the few-pixel layers. */
int mnk_bn_small_fwd_sync(void* p2p, const float* ws, int splits, int ldw, int phases, const float* bias, float* y, int ld_y, int N,
                          int H, int W, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* mean, float* invstd, float* scale, float* z, int ld_z, int relu, int pool,
                          int timeout_ms, void* stream);
int mnk_p2p_max_floats(void);
int mnk_p2p_create(int rank, int world, void** handle_out);
int mnk_p2p_export(void* handle, void* ipc_handle64);
int mnk_p2p_connect(void* handle, const void* all_handles);
int mnk_p2p_allreduce(void* handle, const float* in, float* out, int n, int timeout_ms, void* stream);
int mnk_p2p_error(void* handle, int* flag_out);
/* what the mailbox was allocated as: 3 = uncached device memory (hipDeviceMallocUncached: remote writes over xGMI do not pass
 * through the owning device's L2, so the polling side must not cache the words either -- what RCCL uses for its own flags),
 * 1 = fine-grained, 0 = ordinary device memory (enough between processes of ONE device); negative: invalid handle */
int mnk_p2p_memory_kind(void* handle);
int mnk_p2p_destroy(void* handle);

/* ---- grouped 1x1 convolution (SameBlock3D, modules/util.py:118, dense_motion_module.py:24-28) ----------- */
int mnk_gconv1x1_fwd(const float* x, int ld_x, const float* w, const float* bias, float* y, int ld_y, long rows,
                     int groups, int gsize, void* stream);
int mnk_gconv1x1_bwd_data(const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, long rows, int groups,
                          int gsize, void* stream);
size_t mnk_gconv1x1_workspace_floats(long rows, int groups, int gsize);
int mnk_gconv1x1_bwd_weight(const float* x, int ld_x, const float* dy, int ld_dy, float* dw, float* dbias, long rows,
                            int groups, int gsize, float* ws, size_t ws_floats, void* stream);

/* ---- 1x1 convolution + sigmoid, NHWC in -> (B,C,D,H,W) out (generator.py:48,79-80) ---------------------- */
int mnk_conv1x1_sigmoid_fwd(const float* x, int ld_x, int Cin, const float* w, const float* bias, float* out, int B,
                            int D, int H, int W, int Cout, void* stream);
size_t mnk_conv1x1_workspace_floats(long rows, int Cin, int Cout);
/* general form: act = 1 sigmoid, 0 linear (the discriminator's score head nn.Conv3d(C, 1, 1), discriminator.py:59,77).
 * mnk_conv1x1_bwd: dw == NULL (with dbias, ws ignored) computes the data gradient only, dx == NULL the parameter gradients only */
int mnk_conv1x1_fwd(const float* x, int ld_x, int Cin, const float* w, const float* bias, float* out, int B, int D, int H,
                    int W, int Cout, int act, void* stream);
int mnk_conv1x1_bwd(const float* x, int ld_x, int Cin, const float* w, const float* out, const float* dout, float* dx,
                    int ld_dx, float* dw, float* dbias, int B, int D, int H, int W, int Cout, int act, float* ws,
                    size_t ws_floats, void* stream);
int mnk_conv1x1_sigmoid_bwd(const float* x, int ld_x, int Cin, const float* w, const float* out, const float* dout,
                            float* dx, int ld_dx, float* dw, float* dbias, int B, int D, int H, int W, int Cout,
                            float* ws, size_t ws_floats, void* stream);

/* ---- heat-map -> key-point (modules/keypoint_detector.py:43-78,103-107) ---------------------------------
 * softmax over H*W of heat/temperature, +1e-7, mean (x,y) and centred 2x2 covariance (K <= 16).
 * mean [N][K][2], var [N][K][4], stat [N][K][2] = (row max of the scaled logits, softmax denominator). */
int mnk_softmax_kp_fwd(const float* heat, int ld, int N, int H, int W, int K, float temperature, float* mean,
                       float* var, float* stat, void* stream);
int mnk_softmax_kp_bwd(const float* heat, int ld, int N, int H, int W, int K, float temperature,
                       const float* mean, const float* stat, const float* dmean, const float* dvar, float* dheat,
                       int ld_d, void* stream);
/* The integers of the key-point path (north star "bit-exact keypoint indices"; the reference forms integers from key points in
 * exactly one place, the Visualizer):
 *   mnk_heatmap_argmax: index[n][k] = h * W + w of the largest heat-map logit, first occurrence -- the integer form of the
 *     spatial soft-argmax of keypoint_detector.py:103-104 (soft-max and 1/temperature are monotone);
 *   mnk_kp_pixel_index: pixel[i][2] = floor(size * (mean + 1) / 2), size = (W, H) -- logger.py:99-100
 *     (`spatial_size * (kp_array + 1) / 2`, then rasterised by skimage.draw.circle, :104), n = number of key points. */
int mnk_heatmap_argmax(const float* heat, int ld, int N, int H, int W, int K, int* index, void* stream);
int mnk_kp_pixel_index(const float* mean, long n, int W, int H, int* pixel, void* stream);
/* clip_variance (keypoint_detector.py:62-65): out = var * max(clip, sigma_min(var)) / sigma_min(var) over M 2x2
 * matrices, sigma_min by the closed form of modules/util.py:244-255; backward through both factors.
 * reference_mode 0 (default of the host layer): sigma_max from the sum, sigma_min = |det| / sigma_max -- the reference's number
 * in exact arithmetic, finite where its fp32 form cancels; 1: the reference's own sqrt((s1 - s2) / 2) in fp32, NaNs included. */
int mnk_kp_clip_variance_fwd(const float* var, float clip, long M, float* out, int reference_mode, void* stream);
int mnk_kp_clip_variance_bwd(const float* var, float clip, long M, const float* dout, float* dvar, int reference_mode,
                             void* stream);

/* ---- transfer-time key-point normalisation (transfer.py:31-62 normalize_kp) ------------------
 * mnk_kp_hull_area: area of the convex hull of K (3..32) points [K][2] -> *area (device scalar) -- scipy.spatial.ConvexHull
 *   (points).volume of transfer.py:34-35, which the reference evaluates on the host (a device -> host copy per video).
 * mnk_kp_normalize: kp of a driving video mean_v [B][D][K][2] / var_v [B][D][K][4] relative to its first frame, moved onto
 *   the source's key-points mean_a / var_a [B][1][K][.]:  move_location (scaled by sqrt(*area_a / *area_v) when both are
 *   given), clip_mean (clamp to [-1, 1]), adapt_variance (var_v inverse(var_v[:, 0]) var_a, symmetrised, eigenvalues <= 0
 *   lifted to 1e-6: make_symetric_matrix, transfer.py:17-28).  var_out NULL: means only. */
int mnk_kp_hull_area(const float* points, int K, float* area, void* stream);
int mnk_kp_normalize(const float* mean_v, const float* var_v, const float* mean_a, const float* var_a, int B, int D, int K,
                     const float* area_a, const float* area_v, int move_location, int clip_mean, int adapt_variance,
                     float* mean_out, float* var_out, void* stream);

/* ---- key-point -> movement embedding (modules/movement_embedding.py:42-92, keypoint_detector.py:7-40) ----
 * one kernel renders, per slot (background first when add_bg): [heat-map (driving - source when
 * heatmap_diff), (dx,dy) maps, source image translated by kp_source - kp_driving].  Frames f = b*d + j use
 * source key-points / image of batch entry b.  var_* == NULL selects a constant variance `const_var`.
 * norm_const <= 0 selects 'sum' normalisation (movement_embedding.py:34-38); then `norm` [2][Nf][K] must hold
 * the sums from mnk_gaussian_sums. */
int mnk_gaussian_sums(const float* mean, const float* var, float const_var, int Nkp, int h, int w, float* sums,
                      void* stream);
int mnk_movement_embedding_fwd(const float* img, int ld_img, int Cimg, const float* mean_d, const float* var_d,
                               const float* mean_s, const float* var_s, float const_var, int Nb, int d, int h, int w,
                               int K, int add_bg, int use_heatmap, int use_difference, int use_deformed,
                               int heatmap_diff, float norm_const, const float* norm_d, const float* norm_s,
                               float* out, int ld_out, void* stream);
/* gradients w.r.t. the key-points, one row per frame: dmean_d/dvar_d/dmean_s/dvar_s [Nb*d][K][2|4] (the caller sums
 * the source gradients over the d frames of a batch entry).  Overwritten.  dvar_* may both be NULL (constant
 * variance).  The gradient w.r.t. the source image is not produced: no caller of the reference consumes it
 * (SURVEY.md section 8b "Gradients"). */
int mnk_movement_embedding_bwd(const float* img, int ld_img, int Cimg, const float* mean_d, const float* var_d,
                               const float* mean_s, const float* var_s, float const_var, int Nb, int d, int h, int w,
                               int K, int add_bg, int use_heatmap, int use_difference, int use_deformed,
                               int heatmap_diff, float norm_const, const float* norm_d, const float* norm_s,
                               const float* dout, int ld_out, float* dmean_d, float* dvar_d, float* dmean_s,
                               float* dvar_s, void* stream);

/* ---- dense-motion head (modules/dense_motion_module.py:52-76) --------------------------------------------
 * pred [N][h][w][ld]: channels [0,K+1) mask logits (use_mask), last 2 correction (use_correction);
 * delta [N][K+1][2] = kp_source.mean - kp_driving.mean (slot 0 = background = 0);
 * field [N][h][w][2] = sum_k softmax(mask)_k * delta_k + correction + identity grid. */
int mnk_motion_field_fwd(const float* pred, int ld, const float* delta, int N, int h, int w, int K, int use_mask,
                         int use_correction, float* field, void* stream);
int mnk_motion_field_bwd(const float* pred, int ld, const float* delta, const float* dfield, int N, int h, int w,
                         int K, int use_mask, int use_correction, float* dpred, int ld_d, float* ddelta,
                         void* stream);
/* the same with delta taken from the key points inside the kernels (mask form): mean_s / mean_d [N][K][2] = kp_source.mean /
 * kp_driving.mean of the frame (dense_motion_module.py:52-54: the difference, the zero background slot and their backward
 * passes as no launches of their own); dmean_s = +sum_p m_k dfield, dmean_d = -dmean_s */
int mnk_motion_field_kp_fwd(const float* pred, int ld, const float* mean_s, const float* mean_d, int N, int h, int w, int K,
                            int use_correction, float* field, void* stream);
int mnk_motion_field_kp_bwd(const float* pred, int ld, const float* mean_s, const float* mean_d, const float* dfield, int N, int h,
                            int w, int K, int use_correction, float* dpred, int ld_d, float* dmean_s, float* dmean_d,
                            void* stream);

/* ---- bilinear warp (MotionTransferGenerator.deform_input, modules/generator.py:51-58) --------------------
 * out[n,y,x,off+c] = bilinear(inp[n], field'[n,y,x]) with zeros padding, align_corners=True (torch 0.4.1),
 * field' = field resized to (h,w): mode 0 = nearest index pick, 1 = bilinear align_corners=False
 * ('trilinear' with unchanged depth). */
int mnk_deform_fwd(const float* inp, int ld_in, int C, int h, int w, const float* field, int hf, int wf, int mode,
                   float* out, int ld_out, int out_off, int N, void* stream);
/* Backward, DETERMINISTIC (no floating-point atomics; the reference's CPU grid_sample backward is a fixed-order sum per
 * source texel, a loop over output pixels): pass A stores every output pixel's sampling point and its share of the field
 * gradient in the workspace, pass B gathers per source texel -- the pixels that touch it in pixel order -- and per field
 * texel.  dinp (same layout as inp, may be NULL) is WRITTEN, pad channels 0; dfield [N][hf][wf][2] (may be NULL) is ADDED
 * to (several skips share one field: zero it before the first level).  ws: mnk_deform_bwd_workspace_floats floats. */
size_t mnk_deform_bwd_workspace_floats(int C, int h, int w, int N);
int mnk_deform_bwd(const float* inp, int ld_in, int C, int h, int w, const float* field, int hf, int wf, int mode,
                   const float* dout, int ld_out, int out_off, float* dinp, float* dfield, int N, float* ws, size_t ws_floats,
                   void* stream);
/* All warps of one generator pass in ONE launch each way (generator.py:60-78: every decoder level's appearance skip is
 * warped by the same field, resized per level (mode 0: nearest pick, mode 1: bilinear -- 'trilinear' with unchanged depth, the
 * vox configs), and the key-point embedding is resized the same way into the channels behind it):
 *   out_l[n][y][x][c] = grid_sample(inp_l, resize(field))[c]                    for c < C_l            (mnk_deform_fwd)
 *   out_l[n][y][x][emb_off_l + c] = resize(emb)[n][y][x][c]                    for c < ke_l           (mnk_resize_nearest / _bilinear)
 * forward: EVERY channel of out_l is written (pad channels behind C_l / behind the embedding: 0) -- no zero fill by the caller;
 * backward (two launches, deterministic -- see mnk_deform_bwd): dinp_l (may be NULL), dfield (may be NULL: the sum over the
 * levels in order) and demb[N][He][We][ld_emb] (may be NULL: the gathers of mnk_resize_nearest_bwd / mnk_resize_bilinear_bwd summed level after level,
 * pad channels 0) are all WRITTEN, nothing needs a zero fill.  ws: mnk_warp_levels_bwd_workspace_floats floats.  nlevels <= 12. */
typedef struct MnkWarpLevel {
    const float* inp;   /* [N][h][w][ld_in] */
    float* out;         /* forward: [N][h][w][ld_out] */
    const float* dout;  /* backward: gradient of out */
    float* dinp;        /* backward */
    int ld_in, C, h, w, ld_out, ke, emb_off, reserved;
} MnkWarpLevel;
int mnk_warp_levels_fwd(const MnkWarpLevel* levels, int nlevels, const float* field, int hf, int wf, int mode, const float* emb,
                        int ld_emb, int He, int We, int N, void* stream);
size_t mnk_warp_levels_bwd_workspace_floats(const MnkWarpLevel* levels, int nlevels, int N);
int mnk_warp_levels_bwd(const MnkWarpLevel* levels, int nlevels, const float* field, int hf, int wf, int mode, float* dfield,
                        float* demb, int ld_emb, int He, int We, int N, float* ws, size_t ws_floats, void* stream);

/* ---- feature-matching L1 on the discriminator's activations (modules/losses.py:8-12 reconstruction_loss over
 * discriminator maps; train.py:47-51) ---------------------------------------------------------------------------
 * a: act of 2B frames [generated | real] ([2B][rows][ld], rows = H*W pixels per frame, C <= ld true channels).
 * out[i] = weight * mean over pixels and channels c < C of |a[i] - a[B + i]|        (weight * mean_batch(|fake - real|))
 * da[i]  = g[i] * weight * sign(a[i] - a[B + i]) / (rows * C), da[B + i] = -da[i]; pad channels of da are written 0. */
int mnk_pair_l1_fwd(const float* a, int ld, long rows, int C, int B, float weight, float* out, void* stream);
int mnk_pair_l1_bwd(const float* a, int ld, long rows, int C, int B, float weight, const float* g, float* da,
                    void* stream);
/* ... + addend ([2B][rows][ld], may be NULL): the gradient that reaches the same map through its other consumer (the next
 * discriminator block), added in this pass instead of by an accumulation pass of its own */
int mnk_pair_l1_bwd_add(const float* a, int ld, long rows, int C, int B, float weight, const float* g, const float* addend,
                        float* da, void* stream);
/* image-level term of the same loss (losses.py:8-12 on NCDHW frames: train.py:39-42 reconstruction_deformed, map 0 of
 * 'reconstruction'): out[i] = weight * mean_j |a[i][j] - b[i][j]| over n contiguous floats per sample; backward:
 * da = g[i] * weight / n * sign(a - b), db = -da (either may be NULL) */
int mnk_l1_mean_fwd(const float* a, const float* b, long n, int B, float weight, float* out, void* stream);
int mnk_l1_mean_bwd(const float* a, const float* b, long n, int B, float weight, const float* g, float* da, float* db,
                    void* stream);
/* LSGAN terms (losses.py:15-21) from the score maps of the batched discriminator pass, score[2B][n] = [generated | real]:
 * gen[i] = w_gen * mean_j (1 - sf)^2, disc[i] = w_disc * mean_j ((1 - sr)^2 + sf^2); backward from the upstream gradients
 * of the two vectors (either may be NULL) */
int mnk_gan_terms_fwd(const float* score, int n, int B, float w_gen, float w_disc, float* gen, float* disc, void* stream);
int mnk_gan_terms_bwd(const float* score, int n, int B, float w_gen, float w_disc, const float* ggen, const float* gdisc,
                      float* dscore, void* stream);
/* batch means of the per-sample loss vectors and their sum in one launch (train.py:114 `[val.mean() for val in losses]`,
 * :116 `sum(loss_values)`): `vecs` = HOST array of nvec <= 16 device pointers to `len` floats each;
 * means[i] = mean_j vecs[i][j] for i < nvec, means[nvec] = means[0] + ... + means[nvec - 1] (added in this order).
 * backward: gvecs[i][j] = (gmeans[i] + gtotal[0]) / len for the upstream gradients of the means (nvec floats) and of the
 * sum (one float); either may be NULL = zero. */
int mnk_vec_means_fwd(const float* const* vecs, int nvec, int len, float* means, void* stream);
int mnk_vec_means_bwd(const float* gmeans, const float* gtotal, int nvec, int len, float* gvecs, void* stream);

/* ---- device-side input path ------------------------------------------------------------------
 * Replaces, for the integer-exact transforms, what the reference does on the host per sample inside 4 DataLoader workers
 * (train.py:99): read_video's strip -> frames split, gray -> RGB, RGBA -> RGB and img_as_float32 (frames_dataset.py:14-29),
 * RandomFlip (augmentation.py:91-104), RandomCrop = pad_clip(mode='edge') + crop (:135-171), SelectRandomFrames and
 * SplitSourceDriving / VideoToTensor (:324-366).  The decoded uint8 strips of the whole dataset stay resident in `pool`
 * (HBM); one job = one output frame:
 *   out[out_offset + ch * chan_stride + h * W + w] = float32(strip[r][frame * in_w + c][ch]) * float32(1 / 255)
 *   with (r, c) = clamp((h + y1 - pad_top, w + x1 - pad_left)) into the in_h x in_w frame, c mirrored when hflip.
 * `channels` of the strip: 1 / 2 (gray [+ alpha]: replicated), 3 / 4 (RGB [+ alpha: dropped]).  Cout <= 3 planes are written.
 * For a (B, C, D, H, W) batch tensor: out_offset = (b * C * D + d) * H * W, chan_stride = D * H * W. */
typedef struct MnkFrameJob {
    unsigned long long strip_offset;     /* byte offset of the strip (in_h rows of strip_w pixels of `channels` bytes) in pool */
    unsigned long long out_offset;       /* float offset of this frame's channel-0 plane in `out` */
    unsigned long long chan_stride;      /* floats between channel planes */
    int strip_w, in_h, in_w, channels;
    int frame, hflip, x1, y1;
    int pad_top, pad_left, reserved0, reserved1;
} MnkFrameJob;
int mnk_frames_gather(const unsigned char* pool, const MnkFrameJob* jobs_device, int njobs, int H, int W, int Cout, float* out,
                      void* stream);
/* The same batch path with the reference's non-integer augmentations in front of the crop (round 4): RandomRotation
 * (augmentation.py:175-214 = skimage.transform.rotate), RandomResize (:105-133 = skimage.transform.resize, mode 'constant',
 * anti_aliasing=True: one tap for ratios >= 0.8, the multi-tap Gaussian below that -- round 5, flags & 16) and ColorJitter's hue term (:217-320 = img_as_ubyte ->
 * PIL RGB -> HSV -> uint8 hue shift -> RGB -> img_as_float), in the arithmetic of the versions requirements.txt pins
 * (scikit-image 0.14.0, Pillow 5.2.0, torchvision 0.2.1; restated in oracle/augment_restate.py).  flags: 1 rotate (rot = the
 * inverse map  col = rot[0] c + rot[1] r + rot[2], row = rot[3] c + rot[4] r + rot[5]), 2 resize to (new_h, new_w) with order 1
 * (interpolation='bilinear'), 8 the same with order 0 (RandomResize's default 'nearest': what the shipped configs run), 4 colour
 * jitter (round 4: the hue term; round 5: all four terms of ColorJitter in their shuffled order, see jit_* below).
 * vmin / vmax: min / max of the float32 source frame over its channels (skimage clips a warp's output to its input's range);
 * rot_range: njobs x 2 doubles of scratch when any job rotates or anti-alias-filters (any_rotation != 0: the range of the frame the
 * resize samples, which clips an order-1 resize). */
typedef struct MnkAugJob {
    unsigned long long strip_offset;
    unsigned long long out_offset;
    unsigned long long chan_stride;
    double rot[6];
    int strip_w, in_h, in_w, channels;
    int frame, hflip, x1, y1;
    int pad_top, pad_left, new_h, new_w;
    int flags, hue_shift;
    float vmin, vmax;
    /* ColorJitter (flags & 4): jit_n terms in the order the reference's random.shuffle left them; jit_op: 1 brightness, 2 saturation,
     * 3 hue (hue_shift above), 4 contrast; jit_f: the factors (brightness / saturation / contrast: Pillow's ImageEnhance blend
     * factor as a C float) */
    int jit_n;
    int jit_op[4];
    float jit_f[4];
    int reserved;
    /* anti-aliasing of a down-scaling resize (flags & 16; skimage's resize filters with sigma = (in / out - 1) / 2 per axis before it
     * samples): radii int(4 sigma + 0.5) <= 4 of the row / column pass and the half kernels w[0 .. radius] (w[radius] = centre) as
     * scipy.ndimage's gaussian_filter1d makes them */
    int aa_rr, aa_rc;
    double aa_wr[5], aa_wc[5];
} MnkAugJob;
/* any_contrast / contrast_mean (njobs ints of scratch): a job with a contrast term needs int(mean(luma) + 0.5) of its output frame
 * as it stands in front of that term -- one more pre-pass launch (ImageEnhance.Contrast blends with that constant). */
int mnk_frames_augment(const unsigned char* pool, const MnkAugJob* jobs_device, int njobs, int any_rotation, double* rot_range,
                       int any_contrast, int* contrast_mean, int H, int W, int Cout, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MONKEYNET_HIP_H */
