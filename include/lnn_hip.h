/*
 * lnn_hip.h -- C-ABI of liblnn_hip.so: the MI355X (gfx950) kernels behind the hot path of
 * MECLabTUDA/Lifelong-nnUNet (3-D U-Net training step + EWC / LwF regularisers).
 *
 * The reference is pure Python and has no FFI; every entry point below replaces a PyTorch op that the
 * reference reaches on its hot path (file:line relative to the reference repository root).  The
 * reference-side binding a maintainer would add is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative lnn_status; never throws; the message of the
 *     last failure on the calling thread is returned by lnn_last_error().
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator in the build's
 *     host code).  The library never allocates, frees or retains device memory; work is enqueued
 *     asynchronously on the passed hipStream_t (void* here so the header needs no HIP include).
 *   - activations are fp16 ("h"), channels-last NDHWC: element (n,z,y,x,c) at
 *     ((((n*D+z)*H+y)*W+x)*ld + c); `ld` (channel stride, elements) lets a tensor live inside a wider
 *     concat buffer -- this is how torch.cat((x, skip), 1) (generic_ViT_UNet.py:263) is eliminated.
 *     All channel counts and ld must be multiples of 8 (16-byte vectors), except the C==1 image input.
 *   - parameters / gradients / optimiser state are fp32 in the reference's own tensor layouts
 *     (Conv3d weight (K,C,3,3,3), ConvTranspose3d weight (Cin,Cout,2,2,2)), so state_dict keys and
 *     Fisher / theta* dictionaries interoperate (nnUNetTrainerEWC.py:298-304).
 */
#ifndef LNN_HIP_H
#define LNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lnn_stream_t; /* hipStream_t */

enum lnn_status {
    LNN_OK = 0,
    LNN_ERR_BAD_ARG = -1,     /* null pointer, misaligned pointer, unsupported channel count ... */
    LNN_ERR_WORKSPACE = -2,   /* workspace too small */
    LNN_ERR_LAUNCH = -3,      /* hipGetLastError() after launch */
    LNN_ERR_UNSUPPORTED = -4
};

const char* lnn_last_error(void);
int lnn_version(void);
/* number of CUs / clock (kHz) / device name of the current device; for roofline reporting */
int lnn_device_info(int* cu_count, int* clock_khz, char* name, int name_len);

/* ------------------------------------------------------------------------------------------------
 * Weight packing (fp32 parameter tensors -> fp16 MFMA operand panels).  No reference counterpart:
 * cuDNN does this internally for nn.Conv3d / nn.ConvTranspose3d under autocast
 * (nnUNetTrainerMultiHead.py:619-621).
 *   panel element (t, m, kc) (m padded to 32, kc padded to 16, zero filled) = src[m*stride_m + kc*stride_kc + t*stride_t]
 *   stored blocked as [m/32][kc/16][t][m%32][kc%16]: the 32x16 MFMA operand tiles of all taps of one
 *   (row block, channel chunk) are one contiguous run, which the conv kernels stage with fully coalesced loads.
 * ---------------------------------------------------------------------------------------------- */
int lnn_pack_weights(lnn_stream_t s, const float* src, void* dst_h, int ntaps, int M, int KC,
                     long stride_m, long stride_kc, long stride_t);
size_t lnn_packed_weight_elems(int ntaps, int M, int KC);
/* All layers in ONE launch.  desc_dev: n x 9 int64 on the device, per layer
 *   {src_off, dst_off, stride_m, stride_kc, stride_t, ntaps, M, KC, first}
 * (element offsets relative to src_base / dst_base; first = running sum of the layers' padded panel sizes
 * lnn_packed_weight_elems; total = their sum).  n <= 128. */
int lnn_pack_weights_batched(lnn_stream_t s, const float* src_base, void* dst_base_h, const long* desc_dev, int n,
                             long total);

/* ------------------------------------------------------------------------------------------------
 * nn.Conv3d 3x3x3, padding 1, stride 1|2, bias  (module tree test_MultiHead_Module.py:346-415;
 * conv_op / strided-conv pooling chosen at nnViTUNetTrainer.py:101-104,122).
 *   fwd  : y[n,p,k] = b[k] + sum_{c,d} w[k,c,d] x[n, s*p+d-1, c]
 *   dgrad: dx = conv_transpose(dy, w)           (beta = 1 accumulates into dx: skip connections)
 *   wgrad: dwp[t][k][c] += sum_{n,p} dy[n,p,k] x[n,s*p+d-1,c]   (fp32 packed panel, see lnn_unpack_wgrad)
 * x: (N,Di,Hi,Wi,C) ld_x ; y/dy: (N,Do,Ho,Wo,K) ld_y with Do = (Di-1)/s+1.
 * wp_fwd  = lnn_pack_weights(w, 27, K, C, C*27, 27, 1);  wp_dgrad = lnn_pack_weights(w, 27, C, K, 27, C*27, 1)
 * C == 1 (image input) is handled by a dedicated path in fwd / wgrad (x is then (N,Di,Hi,Wi) fp16).
 * Limits (explicit LNN_ERR_BAD_ARG, no slower fallback since round 5): the stride-1 weight gradient addresses a tile through 32-bit
 * buffer-descriptor offsets, so six consecutive z-planes of x (Hi*Wi*ld_x*2 bytes each) must stay below 2 GB; the z-streaming and
 * macro-tile forward / data-gradient kernels need one z-plane / one sample below 2 GB and otherwise hand the layer to the tile kernel.
 * ---------------------------------------------------------------------------------------------- */
int lnn_conv3d_fwd(lnn_stream_t s, const void* x_h, int ld_x, const void* wp_fwd_h, const float* bias,
                   void* y_h, int ld_y, int N, int Di, int Hi, int Wi, int C, int K, int stride);
int lnn_conv3d_dgrad(lnn_stream_t s, const void* dy_h, int ld_dy, const void* wp_dgrad_h, void* dx_h, int ld_dx,
                     int N, int Di, int Hi, int Wi, int C, int K, int stride, int accumulate);
/* Split-K workspace (optional, both may be NULL / 0): small deep layers -- fewer (8x8x8 tile x 32 channel) units than resident
 * blocks and 20..40 serial 16-channel chunk steps per unit -- are latency-bound; with splitk_ws (fp32 scratch, contents don't
 * matter; k * N*Do*Ho*Wo*roundup32(out channels) elements allow a k-way split, k <= 8) the chunk loop of a unit is split over
 * k blocks that write fp32 partial sums to their own slice, a finalize launch adds the slices in a fixed order (deterministic)
 * and converts.  lnn_conv3d_fwd_in_stats takes the same pair.  Larger layers ignore it. */
int lnn_conv3d_dgrad_ws(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di, int Hi, int Wi,
                        int C, int K, int stride, int accumulate, float* splitk_ws, long splitk_elems);
int lnn_conv3d_wgrad(lnn_stream_t s, const void* x_h, int ld_x, const void* dy_h, int ld_dy, float* dwp,
                     int N, int Di, int Hi, int Wi, int C, int K, int stride);
/* The same three ops (stride 1) on the CHANNEL CONCATENATION of two tensors that is never materialised
 * (torch.cat((up, skip), dim=1) at generic_ViT_UNet.py:263 in front of the first decoder conv of a level):
 *   x = cat(x_a[..., :c_a], x_b[..., :C - c_a]),  dx_a / dx_b receive the matching channel ranges of dx.
 * Both parts share one channel stride (ld_x / ld_dx); c_a is a multiple of 32.  Why: with both halves interleaved in
 * one (N,D,H,W,2c) buffer every 16-channel chunk step touches all 128-byte positions of the tile, 32 CUs x 128 KB of
 * lines is an XCD's whole L2 and the top decoder conv fetched 5.9 GB for a 1.26 GB input (profiles/r01_pmc_traffic.json);
 * as two 64-byte-per-voxel tensors the live set halves and the chunk pairs share their lines. */
int lnn_conv3d_fwd_cat(lnn_stream_t s, const void* x_a_h, const void* x_b_h, int ld_x, int c_a, const void* wp_fwd_h,
                       const float* bias, void* y_h, int ld_y, int N, int Di, int Hi, int Wi, int C, int K);
/* Conv3d (3x3x3, padding 1) -> dense output y (ld_y == K) AND the InstanceNorm statistics of that output in one call
 * (ConvDropoutNormNonlin = instnorm(conv(x)), test_MultiHead_Module.py:287-291; statistics as lnn_instnorm_stats: biased
 * variance over D*H*W per (n, c) of the fp16-stored values).  x_b may be NULL (single input tensor; c_a ignored).  Where the
 * stride-1 z-streaming kernel applies (C = 32 / 64), and for the first layer (C == 1), the sums are taken in the conv's
 * epilogue from the values it stores -- the separate 2 B/element statistics pass disappears; otherwise the call runs the
 * convolution followed by lnn_instnorm_stats.
 * ws >= lnn_instnorm_ws_doubles(N, K). */
int lnn_conv3d_fwd_in_stats(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* wp,
                            const float* bias, void* y, int N, int Di, int Hi, int Wi, int C, int K, int stride,
                            float eps, float* mean, float* rstd, double* ws, float* splitk_ws,
                            long splitk_elems);
int lnn_conv3d_dgrad_cat(lnn_stream_t s, const void* dy_h, int ld_dy, const void* wp_dgrad_h, void* dx_a_h, void* dx_b_h,
                         int ld_dx, int c_a, int N, int Di, int Hi, int Wi, int C, int K, int accumulate);
int lnn_conv3d_wgrad_cat(lnn_stream_t s, const void* x_a_h, const void* x_b_h, int ld_x, int c_a, const void* dy_h, int ld_dy,
                         float* dwp, int N, int Di, int Hi, int Wi, int C, int K);
/* lnn_conv3d_dgrad_cat with the optional split-K workspace of lnn_conv3d_dgrad_ws (the lowest-resolution decoder convolution's data
 * gradient, 320 -> 640 channels @ 10x12x10 at the 160x192x160 plan: 100 (band x channel-block) items for 256 CUs without it). */
int lnn_conv3d_dgrad_cat_ws(lnn_stream_t s, const void* dy_h, int ld_dy, const void* wp_dgrad_h, void* dx_a_h, void* dx_b_h,
                            int ld_dx, int c_a, int N, int Di, int Hi, int Wi, int C, int K, int accumulate, float* splitk_ws,
                            long splitk_elems);

/* ------------------------------------------------------------------------------------------------
 * nn.ConvTranspose3d kernel 2, stride 2, no bias (`tu`, convolutional_upsampling=True
 * nnViTUNetTrainer.py:122):  y[n,2p+d,k] = sum_c x[n,p,c] W[c,k,d].
 * x: (N,D,H,W,C) ; y: (N,2D,2H,2W,K).
 * wp_fwd = lnn_pack_weights(W, 8, K, C, 8, K*8, 1); wp_dgrad = lnn_pack_weights(W, 8, C, K, K*8, 8, 1)
 * wgrad panel dwp[d][c][k] (fp32).
 * ---------------------------------------------------------------------------------------------- */
int lnn_convT3d_k2s2_fwd(lnn_stream_t s, const void* x_h, int ld_x, const void* wp_fwd_h, void* y_h, int ld_y,
                         int N, int D, int H, int W, int C, int K);
int lnn_convT3d_k2s2_dgrad(lnn_stream_t s, const void* dy_h, int ld_dy, const void* wp_dgrad_h, void* dx_h,
                           int ld_dx, int N, int D, int H, int W, int C, int K, int accumulate);
int lnn_convT3d_k2s2_wgrad(lnn_stream_t s, const void* x_h, int ld_x, const void* dy_h, int ld_dy, float* dwp,
                           int N, int D, int H, int W, int C, int K);

/* Deterministic weight gradients: the same kernels, but every writer of a block stores its partial sums into its own copy of
 * the panel inside `parts` (fp32 scratch, contents irrelevant) and an ordered reduction adds the copies to dwp -- no atomics,
 * bit-reproducible run to run (the default path finishes with coalesced fp32 atomics: order-dependent in the last bits).
 * parts_elems >= blocks.x * writers * panel elements (<= 64 M floats for every layer of the BASELINE configs); a scratch that is
 * too small is an error (LNN_ERR_BAD_ARG), not a silent fallback. */
int lnn_conv3d_wgrad_det(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int Di, int Hi, int Wi,
                         int C, int K, int stride, float* parts, long parts_elems);
int lnn_conv3d_wgrad_cat_det(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* dy, int ld_dy, float* dwp,
                             int N, int Di, int Hi, int Wi, int C, int K, float* parts, long parts_elems);
int lnn_convT3d_k2s2_wgrad_det(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D, int H,
                               int W, int C, int K, float* parts, long parts_elems);

/* dst[m*stride_m + kc*stride_kc + t*stride_t] (+)= scale * dwp[t][m][kc]  (panel rows padded to 32/32) */
int lnn_unpack_wgrad(lnn_stream_t s, const float* dwp, float* dst, int ntaps, int M, int KC,
                     long stride_m, long stride_kc, long stride_t, float scale, int accumulate);
size_t lnn_wgrad_panel_elems(int ntaps, int M, int KC);
/* Same for all layers in one launch; desc as in lnn_pack_weights_batched with src_off = panel offset, dst_off =
 * gradient offset, first = running sum of ntaps*M*KC (unpadded), total = their sum. */
int lnn_unpack_wgrad_batched(lnn_stream_t s, const float* panel_base, float* dst_base, const long* desc_dev, int n,
                             long total, float scale, int accumulate);

/* ------------------------------------------------------------------------------------------------
 * nn.InstanceNorm3d(eps, affine=True) + nn.LeakyReLU(slope)  (nnViTUNetTrainer.py:111-114; order
 * conv -> instnorm -> lrelu, test_MultiHead_Module.py:287-291).  y: dense (N,V,C) fp16 (ld = C).
 *   stats : mean/rstd[n][c] over the V voxels, biased variance.  ws: >= 2*N*C doubles, zeroed by the call.
 *   fwd   : z = lrelu(gamma*(y-mean)*rstd + beta)  written with ld_z (may target a concat buffer)
 *   bwd   : g = dz * lrelu'(.) ; dy = gamma*rstd*(g - mean_v(g) - xhat*mean_v(g*xhat)) written IN PLACE
 *           over y; dgamma/dbeta/dbias (+)= (fp32; dbias is the conv bias gradient = sum_v dy).
 *           dbias may be NULL: the sum is analytically zero (InstanceNorm removes the mean) -- what it accumulates is the
 *           fp16 rounding of dy -- and without it pass 2 is a pure stream (no block reductions, no finalize launch).
 *           grad_unscale multiplies the parameter gradients (1/loss_scale).
 * ---------------------------------------------------------------------------------------------- */
int lnn_instnorm_stats(lnn_stream_t s, const void* y_h, int N, long V, int C, float eps, float* mean,
                       float* rstd, double* ws);
int lnn_instnorm_lrelu_fwd(lnn_stream_t s, const void* y_h, void* z_h, int ld_z, int N, long V, int C,
                           const float* mean, const float* rstd, const float* gamma, const float* beta,
                           float slope);
/* lnn_instnorm_lrelu_fwd + lnn_seg1x1_fwd of the same activation in one pass (decoder blocks that feed a seg_outputs head,
 * generic_ViT_UNet.py:263-264): logits (N,K,V) fp32 = seg_w (K,C) . z, computed from the fp16-rounded z the kernel writes.
 * K <= 8, C/8 a power of two <= 64; other shapes: call the two functions.  z may be NULL: only the logits are written (for a
 * caller with no other reader of the normalised tensor; the head's backward, lnn_instnorm_lrelu_seg_bwd, rebuilds it from y). */
int lnn_instnorm_lrelu_seg_fwd(lnn_stream_t s, const void* y, void* z, int ld_z, int N, long V, int C, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, float slope, const float* seg_w,
                               float* logits, int K);
int lnn_instnorm_lrelu_bwd(lnn_stream_t s, void* y_inout_h, const void* dz_h, int ld_dz, int N, long V, int C,
                           const float* mean, const float* rstd, const float* gamma, const float* beta,
                           float slope, float* dgamma, float* dbeta, float* dbias, float grad_unscale,
                           double* ws);
size_t lnn_instnorm_ws_doubles(int N, int C);
/* lnn_seg1x1_bwd + lnn_instnorm_lrelu_bwd of a decoder block that feeds a seg_outputs head (generic_ViT_UNet.py:263-264 in
 * backward), without dL/dz in memory: both passes rebuild dz = [dz_prior] + fp16(sum_k dlogits[k] w[k][c]) and the fp16
 * activation z (for d seg_w) in registers from y and the K-channel dlogits.  dz_prior (may be NULL) is the part of dL/dz that
 * is already in memory (the transposed conv of the next level wrote it), channel stride ld_dz.  dy replaces y in place;
 * dgamma / dbeta / seg_dw (+)= grad_unscale * (...).  K <= 4; other shapes: call the two functions.
 * ws: >= lnn_instnorm_lrelu_seg_bwd_ws_doubles(N, C) doubles. */
int lnn_instnorm_lrelu_seg_bwd(lnn_stream_t s, void* y_inout_h, const void* dz_prior_h, int ld_dz, const float* seg_w,
                               const float* dlogits, float* seg_dw, int K, int N, long V, int C, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, float slope, float* dgamma,
                               float* dbeta, float grad_unscale, double* ws);
size_t lnn_instnorm_lrelu_seg_bwd_ws_doubles(int N, int C);
/* The FIRST block of the network (Conv3d(1 -> K) + InstanceNorm + LeakyReLU; no data gradient is needed behind it): pass 1 of
 * lnn_instnorm_lrelu_bwd alone -- ws[(n*C + c)*3 + {0,1}] = sum g, sum g*xhat, dgamma / dbeta (+)= -- with y left untouched;
 * lnn_conv3d_wgrad_c1_in_bwd then builds dy = gamma*rstd*(g - s1/V - xhat*s2/V) tile by tile while it stages the weight
 * gradient's operand, so dy never exists in memory (the apply pass and the re-read of dy disappear).
 * ws: >= lnn_instnorm_ws_doubles(N, C), the same workspace for both calls.  dwp: the C == 1 panel of lnn_conv3d_wgrad;
 * parts / parts_elems: deterministic-mode scratch as in lnn_conv3d_wgrad_det, or NULL / 0. */
int lnn_instnorm_lrelu_bwd_sums(lnn_stream_t s, const void* y_h, const void* dz_h, int ld_dz, int N, long V, int C,
                                const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                                float* dgamma, float* dbeta, float grad_unscale, double* ws);
int lnn_conv3d_wgrad_c1_in_bwd(lnn_stream_t s, const void* x_h, const void* y_h, const void* dz_h, int ld_dz, float* dwp, int N,
                               int D, int H, int W, int K, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, float slope, const double* ws, float* parts, long parts_elems);
/* Pass 1 of lnn_instnorm_lrelu_bwd taken in the epilogue of the data gradient that PRODUCES dz (ConvDropoutNormNonlin pairs of a
 * StackedConvLayers stage, test_MultiHead_Module.py:287-291: block 0 = conv, instnorm, lrelu -> block 1 = conv ...; the backward of
 * block 1's convolution writes dL/dz of block 0).  lnn_conv3d_dgrad_in_bwd_sums = lnn_conv3d_dgrad_ws(stride 1, no accumulate) into
 * dx, followed by lnn_instnorm_lrelu_bwd_sums(u, dx, ...) for the block whose convolution output is u (dense [N][V][C] fp16,
 * untouched): same outputs (dx, ws sums, dgamma / dbeta (+)=).  For 32 -> 32 channels on long z columns (the highest resolution)
 * the reduce is fused into the z-streaming data-gradient kernel (dz is reduced from the registers it is stored from, u arrives by
 * direct-to-LDS loads beside the input planes): dz and u are not read a second time (-4 of the 10 B per element the backward of
 * the normalisation moves); other shapes run the two calls.  lnn_instnorm_lrelu_bwd_apply is pass 2 alone (dy in place over y
 * from the sums in ws).  ws: >= lnn_instnorm_ws_doubles(N, C), the same workspace for both calls. */
int lnn_conv3d_dgrad_in_bwd_sums(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di,
                                 int Hi, int Wi, int C, int K, const void* u_h, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, float slope, float* dgamma, float* dbeta,
                                 float grad_unscale, double* ws, float* splitk_ws, long splitk_elems);
int lnn_instnorm_lrelu_bwd_apply(lnn_stream_t s, void* y_h, const void* dz_h, int ld_dz, int N, long V, int C, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, float slope, double* ws);
/* Small volumes (the two lowest levels of a 160x192x160 plan: <= lnn_instnorm_small_volume() = 2048 voxels per sample).  There a
 * ConvDropoutNormNonlin block (test_MultiHead_Module.py:287-291, 394-415) is ~1.5 MB and launch-latency-bound: statistics ->
 * statistics finalize -> normalise were three dependent ~6 us launches behind the convolution, reduce -> sums -> apply three in the
 * backward.
 * lnn_conv3d_fwd_in_lrelu = lnn_conv3d_fwd_in_stats + lnn_instnorm_lrelu_fwd (same outputs: y, mean, rstd, z; x_b / c_a: second part
 * of a channel concatenation or NULL / 0): on small volumes the normalisation is ONE launch behind the convolution; on larger
 * volumes it IS the two calls.
 * lnn_conv3d_dgrad_in_bwd (small volumes only, else LNN_ERR_BAD_ARG) = lnn_conv3d_dgrad_ws(stride 1, no accumulate) into dx +
 * lnn_instnorm_lrelu_bwd(u, dx, ...) for the block whose convolution output is u: afterwards dx holds dL/dz, u holds dL/du in place,
 * dgamma / dbeta (+)= the affine gradients, ws[(n C + c) 3 + {0, 1}] the sums.
 * lnn_instnorm_lrelu_bwd itself takes the one-launch path on small volumes (dbias == NULL). */
int lnn_instnorm_small_volume(void);
int lnn_conv3d_fwd_in_lrelu(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* wp, const float* bias,
                            void* y, int N, int Di, int Hi, int Wi, int C, int K, int stride, float eps, float* mean, float* rstd,
                            const float* gamma, const float* beta, float slope, void* z_h, int ld_z, double* ws, float* splitk_ws,
                            long splitk_elems);
int lnn_conv3d_dgrad_in_bwd(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di, int Hi,
                            int Wi, int C, int K, void* u_h, const float* mean, const float* rstd, const float* gamma,
                            const float* beta, float slope, float* dgamma, float* dbeta, float grad_unscale, double* ws,
                            float* splitk_ws, long splitk_elems);

/* ------------------------------------------------------------------------------------------------
 * seg_outputs[u]: nn.Conv3d 1x1x1, no bias (test_MultiHead_Module.py:427-431).
 *   fwd : logits[n][k][v] (fp32, NCDHW as the reference returns them, generic_ViT_UNet.py:282-284)
 *   bwd : dz[n,v,c] (+)= sum_k dlogits[n][k][v] w[k][c]  (fp16, dlogits already carry the loss scale)
 *         dw[k][c] += grad_unscale * sum_{n,v} dlogits[n][k][v] z[n,v,c]
 * ---------------------------------------------------------------------------------------------- */
int lnn_seg1x1_fwd(lnn_stream_t s, const void* z_h, int ld_z, const float* w, float* logits, int N, long V,
                   int C, int K);
int lnn_seg1x1_bwd(lnn_stream_t s, const void* z_h, int ld_z, const float* w, const float* dlogits, void* dz_h,
                   int ld_dz, float* dw, int N, long V, int C, int K, int accumulate_dz, float grad_unscale, float* ws);
/* ws: >= lnn_seg1x1_bwd_ws_floats(N, C) floats of scratch for the per-block dw partials (summed by a small finalize
 * launch), or NULL: fp32 atomics straight into dw (slower: ~2000 blocks contend for K*C addresses). */
size_t lnn_seg1x1_bwd_ws_floats(int N, int C);

/* ------------------------------------------------------------------------------------------------
 * DC_and_CE_loss({'batch_dice','smooth':1e-5,'do_bg':False},{}) for ONE deep-supervision level
 * (constructed nnUNetTrainerMultiHead.py:1385; upstream formula SURVEY.md A.3).
 *   fwd : stats[n][k][3] = soft tp/fp/fn (double), ce_sum (double) -> loss = CE + Dice (fp32, *out_loss)
 *   bwd : dlogits = gscale * d(loss)/d(logits)   (gscale = ds weight * upstream grad * loss scale)
 * labels: (N,1,V) float holding integers (nnUNetTrainerMultiHead.py:606-608).  ws >= lnn_dice_ce_ws_doubles.
 * Limits (error code, not a fallback): 2 <= K <= 8 logit channels, 1 <= N <= 4096 samples per call (the per-rank batch).
 * ---------------------------------------------------------------------------------------------- */
int lnn_dice_ce_fwd(lnn_stream_t s, const float* logits, const float* labels, int N, int K, long V,
                    int batch_dice, float smooth, float* out_loss, double* ws);
/* One deep-supervision level of MultipleOutputLoss2 (weights recipe MH.py:1373-1383): as lnn_dice_ce_fwd, and
 * total[0] = (accumulate ? total[0] : 0) + weight * out_loss[0] -- the weighted sum over the levels without a host-side
 * multiply / add per level (stream-ordered read-modify-write of one float). */
int lnn_dice_ce_fwd_ds(lnn_stream_t s, const float* logits, const float* labels, int N, int K, long V, int batch_dice,
                       float smooth, float* out_loss, double* ws, float weight, float* total, int accumulate);
int lnn_dice_ce_bwd(lnn_stream_t s, const float* logits, const float* labels, int N, int K, long V,
                    int batch_dice, float smooth, const double* ws, float gscale, const float* gscale_dev,
                    float dice_scale, float* dlogits);
/* gscale_dev (may be NULL): one device float multiplied into gscale.  dice_scale multiplies the Dice part only.
 * Data-parallel batch Dice: the first N*K*3 doubles of ws (tp/fp/fn) are all-reduced (sum) by the
 * caller between fwd and bwd, the loss is recomputed with lnn_dice_ce_loss_from_totals, and dice_scale = world size
 * (the Dice term is global, its gradient w.r.t. the local logits must survive the 1/world of the gradient average). */
int lnn_dice_ce_loss_from_totals(lnn_stream_t s, const double* ws, int N, int K, long V, int batch_dice, float smooth,
                                 float* out_loss);
size_t lnn_dice_ce_ws_doubles(int N, int K);

/* argmax + per-sample hard TP/FP/FN for the foreground classes (nnUNetTrainerMultiHead.py:938-951).
 * counts: (N, K-1, 3) float, zeroed by the call. */
int lnn_online_dice_counts(lnn_stream_t s, const float* logits, const float* labels, int N, int K, long V,
                           float* counts);

/* ------------------------------------------------------------------------------------------------
 * MiB (loss part): cross-entropy of softmax(x) against a per-voxel target distribution q,
 *   out = scale * mean_v sum_k q_k * (logsumexp(x) - x_k),   dx = gscale * scale / count * (softmax(x) - q).
 *   soft = 0: target = labels (N,1,V) float integers, q = one-hot, voxels with label == ignore_index are skipped and the
 *             mean runs over the counted voxels  (RobustCrossEntropyLoss(ignore_index=255), deep_supervision.py:393);
 *   soft = 1: target = teacher logits (N,K,V), q = softmax(alpha * target), scale = 1/K gives
 *             UnbiasedKnowledgeDistillationLoss for equal class sets (knowledge_distillation.py:11-32 as used at
 *             deep_supervision.py:409-413).
 * x: (N,K,V) fp32 logits.  ws: >= 2 doubles (sum, count), filled by fwd and read by bwd.
 * ---------------------------------------------------------------------------------------------- */
int lnn_target_ce_fwd(lnn_stream_t s, const float* x, const float* target, int soft, int N, int K, long V, float alpha,
                      int ignore_index, float scale, float* out, double* ws);
int lnn_target_ce_bwd(lnn_stream_t s, const float* x, const float* target, int soft, int N, int K, long V, float alpha,
                      int ignore_index, float scale, const double* ws, float gscale, const float* gscale_dev, float* dx);

/* ------------------------------------------------------------------------------------------------
 * Sliding-window inference (predict.py:208-219 -> upstream SegmentationNetwork._internal_predict_3D_3Dconv_tiled):
 * one tile:  agg[k, o + u] += weight * gauss[u] * softmax(logits)[k, flip(u)],   nb[o + u] += gauss[u] (if add_nb)
 *   logits (K, pd,ph,pw) fp32 of ONE tile as the network produced it from the tile mirrored along flip_mask
 *   (bit 2 = z, 1 = y, 0 = x); gauss (pd,ph,pw) importance map or NULL (= 1); agg (K, D,H,W), nb (D,H,W) fp32;
 *   weight = 1 / number of mirror passes.  Tiles of one volume must be accumulated in stream order (plain +=).
 * finalize: agg /= nb (class probabilities, in place), seg[v] = argmax_k.
 * ---------------------------------------------------------------------------------------------- */
int lnn_softmax_accumulate(lnn_stream_t s, const float* logits, const float* gauss, float* agg, float* nb, int K,
                           int pd, int ph, int pw, int D, int H, int W, int oz, int oy, int ox, int flip_mask,
                           float weight, int add_nb);
int lnn_softmax_finalize(lnn_stream_t s, float* agg, const float* nb, int K, long V, int* seg);

/* ------------------------------------------------------------------------------------------------
 * LwF distillation (deep_supervision.py:194-196):
 *   out = (1/N) sum_{n,k,v} softmax(t/T)_k * (logsoftmax(t/T)_k - logsoftmax(y/T)_k)
 * pred / teach: (N,K,V) fp32.  out: one float.  ws: >= lnn_kl_logits_ws_doubles(N) doubles of scratch (per-block partial
 * sums, folded in a fixed order: the result is deterministic; contents on entry do not matter).
 * ---------------------------------------------------------------------------------------------- */
int lnn_kl_logits(lnn_stream_t s, const float* pred, const float* teach, int N, int K, long V, float T,
                  float* out, double* ws);
size_t lnn_kl_logits_ws_doubles(int N);

/* ------------------------------------------------------------------------------------------------
 * PLOP / POD (plop/nnUNetTrainerPLOP.py, pod/nnUNetTrainerPOD.py; deep_supervision.py:217-381, embeddings.py:3-41).
 * Both value-only (the reference's hooks store detached conv outputs, PLOP.py:352-357).
 *
 * lnn_plop_pseudo_labels  (MultipleOutputLossPLOP._pseudo_label_loss, DS.py:292-318): per voxel of the OLD model's
 *   logits x_old (N,K,D,H,W fp32): probs = softmax, pseudo = argmax, valid = entropy(probs)/max_entropy <
 *   thresholds[pseudo] (entropy as crossentropy.py:6-16), bg = (y == 0).
 *     labels_not_pseudo = y, 255 where bg & valid       (target of the first CE term, DS.py:306-309)
 *     labels_pseudo     = 255, pseudo where bg & valid  (target of the second CE term, DS.py:313-317)
 *     num / den (N*W ints): counts of bg & valid / bg voxels per (sample, last-axis column) -- the reference sums
 *     the (B,D,H,W) masks over dims (1,2) only (DS.py:299,301), so its adaptive factor is a (B,W) table.
 *   y: (N,D*H*W) fp32 labels.
 *
 * lnn_local_pod  (embeddings.local_POD, embeddings.py:9-41) on two equally strided 5-D views h, h_old of logical
 *   shape (N,C,D,S,S) (fp16 when is_fp16 else fp32; element strides sn,sc,sd,sy,sx):
 *     pod = mean over (n, 2C, d) of the L2 norm, over all windows of scales 1..scales-1 (scale 0 yields no window:
 *           range(0, W-w, w) is empty for w == W, embeddings.py:30) and their in-window offsets, of the width- /
 *           height-pooled window means of (h - h_old);
 *     if dist_inout: dist = (dist + pod_lambda * pod) / num_layers   (DS.py:270-276: the division is inside the loop);
 *     if pod_out:    pod_out[0] = pod.
 *   ws: >= 2*N*C*D floats.  The last two dims must be equal (the reference's torch.cat fails otherwise) and
 *   S >> (scales-1) must be > 0 (its assert, embeddings.py:26-27).
 * ---------------------------------------------------------------------------------------------- */
int lnn_plop_pseudo_labels(lnn_stream_t s, const float* x_old, const float* y, const float* thresholds, float max_entropy,
                           int N, int K, int D, int H, int W, float* labels_not_pseudo, float* labels_pseudo, int* num,
                           int* den);
int lnn_local_pod(lnn_stream_t s, const void* h, const void* h_old, int is_fp16, int N, int C, int D, int S, long sn, long sc,
                  long sd, long sy, long sx, int scales, float pod_lambda, int num_layers, float* ws, float* dist_inout,
                  float* pod_out);

/* ------------------------------------------------------------------------------------------------
 * Flat-arena parameter kernels (fp32, n elements).
 *   EWC penalty  (deep_supervision.py:80):  out = lambda/2 * sum F (theta-theta*)^2 ;
 *                 bwd: grad += gscale * lambda * F (theta-theta*)
 *   Fisher       (nnUNetTrainerEWC.py:303): F = (g*unscale)^2 ; accumulate / EMA variants
 *   grad norm + clip (nnUNetTrainerMultiHead.py:629,640): sumsq of unscaled grads, non-finite flag
 *   SGD Nesterov (nnUNetTrainerMultiHead.py:299-300): fused unscale*clip, weight decay, momentum, update
 * ---------------------------------------------------------------------------------------------- */
int lnn_ewc_penalty_fwd(lnn_stream_t s, const float* theta, const float* theta_star, const float* fisher,
                        long n, float lambda, float* out, double* ws);
int lnn_ewc_penalty_bwd(lnn_stream_t s, const float* theta, const float* theta_star, const float* fisher,
                        long n, float lambda, float gscale, const float* gscale_dev, float* grad);
int lnn_fisher_square(lnn_stream_t s, const float* grad, float* fisher, long n, float unscale);
int lnn_fisher_accumulate(lnn_stream_t s, const float* grad, float* fisher, long n, float unscale, float weight);
int lnn_fisher_ema(lnn_stream_t s, const float* grad, float* fisher, long n, float unscale, float alpha);
/* Riemannian Walk running statistics, rw/nnUNetTrainerRW.py:231-265 (_update_f_s_values), fused over a range of the
 * flat arenas:  g = grad*inv_scale*min(1, max_norm/(sqrt(ctrl[0])+1e-6))  (the unscaled, clipped gradient the
 * reference reads from param.grad after its optimiser step; ctrl = {sum g^2, #non-finite} from lnn_gradnorm_sumsq,
 * may be NULL: no clipping);  if have_prev: score += max(0, g*(prev-theta)/(0.5*F*(theta-prev)^2 + eps));
 * prev = theta;  F = alpha*g^2 + (1-alpha)*F.  Nothing is updated when ctrl[1] > 0 (skipped step). */
int lnn_rw_update(lnn_stream_t s, const float* theta, float* prev, const float* grad, float* fisher, float* score,
                  long n, float inv_scale, float max_norm, const double* ctrl, float alpha, float eps, int have_prev);
/* out2[0] += sum (g*unscale)^2 (double), out2[1] += number of non-finite elements (double);
 * zero_first != 0 clears the pair before (several arena ranges can accumulate into one pair).  Deterministic two-stage
 * reduction (fixed summation order, no atomics: data-parallel ranks holding identical gradients get identical clip
 * coefficients): out2 must have room for lnn_flat_reduce_ws_doubles() doubles -- the pair first, block partials after. */
int lnn_gradnorm_sumsq(lnn_stream_t s, const float* grad, long n, float unscale, double* out2, int zero_first);
/* doubles the `ws` / `out2` argument of lnn_ewc_penalty_fwd / lnn_gradnorm_sumsq must hold */
long lnn_flat_reduce_ws_doubles(void);
int lnn_sgd_nesterov_step(lnn_stream_t s, float* theta, float* momentum_buf, const float* grad, long n,
                          float lr, float momentum, float weight_decay, float grad_scale, int first_step);

/* as above with the unscale / clip coefficient / inf-skip decision taken ON DEVICE from ctrl = out2 of
 * lnn_gradnorm_sumsq: coef = min(1, max_norm/(sqrt(ctrl[0])+1e-6)); ctrl[1] > 0 skips the step
 * (GradScaler.step + clip_grad_norm_, nnUNetTrainerMultiHead.py:627-631).  momentum_buf must start zeroed. */
int lnn_sgd_nesterov_step_clipped(lnn_stream_t s, float* theta, float* momentum_buf, const float* grad, long n,
                                  float lr, float momentum, float weight_decay, float inv_scale, float max_norm,
                                  const double* ctrl);

/* ------------------------------------------------------------------------------------------------
 * fp32-STORAGE path (the reference's fp16=False branch, multihead/nnUNetTrainerMultiHead.py:632-641; --fp32 at
 * run/run_training.py:71): direct fp32 kernels with fp64 accumulation in a fixed order (bit-reproducible).  A parity
 * mode, not a fast path.  Activations NDHWC fp32 with channel stride ld; weights in PyTorch's layouts straight from the
 * parameter arena (Conv3d (K,C,3,3,3); ConvTranspose3d (Cin,Cout,2,2,2); 1x1x1 (K,C)); *_wgrad ADD into dw; logits NCDHW.
 * lnn_f32_instnorm_lrelu_bwd overwrites y with dL/dy and adds dgamma / dbeta (ws >= 2*N*C doubles); the conv-bias
 * gradient under InstanceNorm is identically zero and is not written.
 * ---------------------------------------------------------------------------------------------- */
int lnn_f32_conv3d_fwd(lnn_stream_t s, const float* x, int ld_x, const float* w, const float* bias, float* y, int ld_y,
                       int N, int Di, int Hi, int Wi, int C, int K, int stride);
int lnn_f32_conv3d_dgrad(lnn_stream_t s, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N, int Di,
                         int Hi, int Wi, int C, int K, int stride, int accumulate);
int lnn_f32_conv3d_wgrad(lnn_stream_t s, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N, int Di,
                         int Hi, int Wi, int C, int K, int stride);
int lnn_f32_convT3d_k2s2_fwd(lnn_stream_t s, const float* x, int ld_x, const float* w, float* y, int ld_y, int N, int D,
                             int H, int W, int C, int K);
int lnn_f32_convT3d_k2s2_dgrad(lnn_stream_t s, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N,
                               int D, int H, int W, int C, int K, int accumulate);
int lnn_f32_convT3d_k2s2_wgrad(lnn_stream_t s, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N,
                               int D, int H, int W, int C, int K);
/* generic geometry of the fp32 parity path (same meaning as the lnn_*_g entries above; weights in PyTorch's layouts
 * (K,C,kz,ky,kx) / (Cin,Cout,sz,sy,sx)) */
int lnn_f32_conv3d_fwd_g(lnn_stream_t s, const float* x, int ld_x, const float* w, const float* bias, float* y, int ld_y, int N,
                         int Di, int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx);
int lnn_f32_conv3d_dgrad_g(lnn_stream_t s, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N, int Di,
                           int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, int accumulate);
int lnn_f32_conv3d_wgrad_g(lnn_stream_t s, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N, int Di,
                           int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx);
int lnn_f32_convT3d_fwd_g(lnn_stream_t s, const float* x, int ld_x, const float* w, float* y, int ld_y, int N, int D, int H,
                          int W, int C, int K, int sz, int sy, int sx);
int lnn_f32_convT3d_dgrad_g(lnn_stream_t s, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N, int D,
                            int H, int W, int C, int K, int sz, int sy, int sx, int accumulate);
int lnn_f32_convT3d_wgrad_g(lnn_stream_t s, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N, int D,
                            int H, int W, int C, int K, int sz, int sy, int sx);
/* multi-channel image: (N, C, V) fp32 (the data dict's layout, nnUNetTrainerMultiHead.py:606-608) -> channels-last with channel
 * stride ld, fp16 for the MFMA path (ld = 16: the first convolution runs with its input channels zero-padded to one 16-channel
 * chunk) or fp32 for the parity path; channels [C, ld) are not written */
int lnn_image_to_cl_h(lnn_stream_t s, const float* src, void* dst_h, int N, int C, long V, int ld);
int lnn_f32_image_to_cl(lnn_stream_t s, const float* src, float* dst, int N, int C, long V, int ld);
int lnn_f32_instnorm_lrelu_fwd(lnn_stream_t s, const float* y, int ld_y, float* z, int ld_z, int N, long V, int C, float eps,
                               float* mean, float* rstd, const float* gamma, const float* beta, float slope);
int lnn_f32_instnorm_lrelu_bwd(lnn_stream_t s, float* y, int ld_y, const float* dz, int ld_dz, int N, long V, int C,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                               float* dgamma, float* dbeta, double* ws);
int lnn_f32_seg1x1_fwd(lnn_stream_t s, const float* z, int ld_z, const float* w, float* logits, int N, long V, int C, int K);
int lnn_f32_seg1x1_bwd(lnn_stream_t s, const float* z, int ld_z, const float* w, const float* dlogits, float* gz, int ld_gz,
                       float* dw, int N, long V, int C, int K, int accumulate);

/* ------------------------------------------------------------------------------------------------
 * Generic geometry (round 4).  nnU-Net builds Generic_UNet from the plans' `conv_kernel_sizes` / `pool_op_kernel_sizes`
 * (nnUNetTrainerMultiHead.py:348-369): per axis a kernel extent of 1 or 3 (padding k / 2) and a stride of 1 or 2; the
 * transposed convolution of a level has kernel == stride == that level's pooling.  Same layouts as above (activations NDHWC
 * fp16 with a channel stride, weights as blocked panels from lnn_pack_weights with ntaps = kz*ky*kx taps in (z, y, x) order,
 * fp32 gradient panels of lnn_wgrad_panel_elems(ntaps, ...)).  Replaces torch.nn.functional.conv3d / conv_transpose3d and
 * their autograd for those shapes.  Di/Hi/Wi = the convolution's INPUT extents, output extents (D - 1) / s + 1;
 * D/H/W of the transposed entries = ITS input extents, output D * s.  splitk_ws (optional, may be NULL): fp32 scratch that lets
 * small volumes split the contraction over more waves; parts (optional): scratch of the deterministic weight gradient.
 * The tensors a call gathers from must be smaller than 2 GB and a weight-gradient call may loop over at most 2^24 voxels per launch
 * (error otherwise, not a fallback: split the batch).
 * ---------------------------------------------------------------------------------------------- */
int lnn_conv3d_fwd_g(lnn_stream_t s, const void* x, int ld_x, const void* wp, const float* bias, void* y, int ld_y, int N,
                     int Di, int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, float* splitk_ws,
                     long splitk_elems);
int lnn_conv3d_dgrad_g(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di, int Hi,
                       int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, int accumulate, float* splitk_ws,
                       long splitk_elems);
int lnn_conv3d_wgrad_g(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int Di, int Hi,
                       int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, float* parts, long parts_elems);
int lnn_convT3d_fwd_g(lnn_stream_t s, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N, int D, int H, int W,
                      int C, int K, int sz, int sy, int sx, float* splitk_ws, long splitk_elems);
int lnn_convT3d_dgrad_g(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int D, int H,
                        int W, int C, int K, int sz, int sy, int sx, int accumulate, float* splitk_ws, long splitk_elems);
int lnn_convT3d_wgrad_g(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D, int H,
                        int W, int C, int K, int sz, int sy, int sx, float* parts, long parts_elems);
/* lnn_convT3d_k2s2_fwd / _dgrad with the optional split-K scratch (as lnn_conv3d_dgrad_ws) */
int lnn_convT3d_k2s2_fwd_ws(lnn_stream_t s, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N, int D, int H,
                            int W, int C, int K, float* splitk_ws, long splitk_elems);
int lnn_convT3d_k2s2_dgrad_ws(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int D,
                              int H, int W, int C, int K, int accumulate, float* splitk_ws, long splitk_elems);

/* fp32 <-> fp16 helpers for the image input (N,1,D,H,W f32 -> fp16, same memory order when C == 1) */
int lnn_cast_f32_to_h(lnn_stream_t s, const float* src, void* dst_h, long n);

#ifdef __cplusplus
}
#endif
#endif /* LNN_HIP_H */
