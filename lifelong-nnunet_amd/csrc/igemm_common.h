// Parameter blocks shared by the implicit-GEMM convolution kernels (internal).
#pragma once
#include "lnn_common.h"

struct TapTable {
    int ntaps;
    int taps_per_group;
    unsigned short pos_off[27];  // offset (in LDS tile positions) of the tap
    unsigned char slot[27];      // weight panel slot of the tap
};

struct ConvParams {
    const half_t* x;
    const half_t* x2;          // second input tensor of a channel concatenation (channels >= csplit), same ld_x
    half_t* y2;                // second output tensor (channels >= msplit), same ld_y
    int csplit = 0x7fffffff, msplit = 0x7fffffff;   // "never": a single tensor
    const half_t* wp;
    const float* bias;
    half_t* y;
    int ld_x, ld_y;
    int N, Di, Hi, Wi, Do, Ho, Wo;
    int C, M, Mpad, KCpad;
    int wtaps;                 // tap slots of the weight panel (27, 8 or 1)
    int Ld, Lh, Lw;
    int tiles_z, tiles_y, tiles_x;
    int os, par_z, par_y, par_x;
    int pad_lo;
    int accumulate;
    float* stats_pws = nullptr;   // v9 only: fused InstanceNorm statistics partials [2][stats_nblk][N][M] (see igemm_conv_v9.hip)
    int stats_nblk = 0;
    // v9 data gradient only: pass 1 of the InstanceNorm + LeakyReLU backward of the block that produced this layer's input, taken in
    // the epilogue (igemm_conv_v9.hip, EPI = 2).  red_u = that block's convolution output [N][Do][Ho][Wo][red_ld], M channels;
    // partials of sum g and sum g u land in stats_pws like the statistics
    const half_t* red_u = nullptr;
    int red_ld = 0;
    const float *red_mean = nullptr, *red_rstd = nullptr, *red_gamma = nullptr, *red_beta = nullptr;
    float red_slope = 0.f;
    // z-streaming kernel with explicit axis strides (igemm_conv_v9.hip, GS instances: the permuted-axes walk of a [1,3,3] convolution):
    // element strides of the {walk, footprint-row, footprint-column} axes and of a sample, bytes a plane descriptor may span
    int gs_in[3] = {0, 0, 0}, gs_out[3] = {0, 0, 0};
    long gs_in_n = 0, gs_out_n = 0;
    unsigned gs_nrec_in = 0, gs_nrec_out = 0;
    int ksplit = 1;               // macro-tile kernel: the 16-channel chunk range is split over ksplit blocks which write fp32 partial
    float* scratch = nullptr;     //   sums to scratch[part][voxel][Mpad]; lnn_launch_splitk_finalize adds the slices and converts
    unsigned long long* dbg;   // optional phase-cycle accumulators (LNN_DEBUG_PHASES), null in production
    TapTable taps;
};


// Weight panels are blocked for the kernels' staging loads: [Mpad/32][KCpad/16][tap slot][32 rows][16 channels]
// fp16, i.e. the (32 output channels x 16 input channels) operand tiles of ALL taps of one (row block, chunk) are one
// contiguous run (27 x 1 KB): a wave-wide 16-byte-per-lane load covers 8 cache lines instead of 32.
__host__ __device__ __forceinline__ long lnn_panel_off(int slot, int m, int kc, int wtaps, int KCpad) {
    return ((((long)(m >> 5) * (KCpad >> 4) + (kc >> 4)) * wtaps + slot) * 32 + (m & 31)) * 16 + (kc & 15);
}

// tile kernel ("v5", igemm_conv_tile.hip): 8x8x8-voxel x 32-channel units, persistent blocks, software-pipelined staging
int lnn_launch_conv_s1_tile(hipStream_t s, ConvParams& p, const char* name);
// v9 (igemm_conv_v9.hip): z-streaming, register-resident weights, direct-to-LDS input ring (C = 32 / 64, M % 32 == 0)
bool lnn_conv_s1_v9_supported(const ConvParams& p);
int lnn_conv_s1_v9_stats_slots(const ConvParams& p);
bool lnn_conv_s1_v9_red_supported(const ConvParams& p);      // a fused-reduce instance (EPI = 2) exists for this shape
// igemm_conv_mt.hip: scratch slices -> fp16 output (+ bias, + old value when accumulating)
// norm_act.hip: mean / rstd from per-slot partial sums pws[a][slot][n*C + c] (the finalize half of lnn_instnorm_stats)
int lnn_launch_splitk_finalize(hipStream_t s, const ConvParams& p, const char* name);
// norm_act.hip, small volumes (<= lnn_instnorm_small_volume() voxels per sample): statistics + normalise + LeakyReLU in ONE launch;
// the whole InstanceNorm + LeakyReLU backward in ONE launch
int lnn_launch_in_small_fwd(hipStream_t s, const void* y, void* z, int ld_z, int N, long V, int C, float eps, const float* gamma,
                            const float* beta, float slope, float* mean, float* rstd);
int lnn_launch_in_small_bwd(hipStream_t s, void* y, const void* dz, int ld_dz, int N, long V, int C, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, float slope, double* ws, float* dgamma, float* dbeta, float unscale);
int lnn_launch_in_stats_finalize(hipStream_t s, const float* pws, int nslots, int N, int C, long V, float eps, float* mean, float* rstd);
int lnn_launch_in_bwd_sums_raw(hipStream_t s, const float* pws, int nslots, int N, int C, const float* mean, const float* rstd,
                               double* ws, float* dgamma, float* dbeta, float unscale);
int lnn_launch_conv_s1_v9(hipStream_t s, ConvParams& p, const char* name);
// [1,3,3] stride-1 convolution forward / data gradient on the z-streaming kernel with permuted axes (walk along H); -1 = not covered
int lnn_conv_k133_on_v9(hipStream_t s, const void* x, int ld_x, const void* wp, const float* bias, void* y, int ld_y, int N, int D, int H,
                        int W, int C, int M, int flip, const char* name);
// macro-tile kernel with in-block split-K over the taps (igemm_conv_mt.hip): the deep levels (>= 128 channels, short volumes)
bool lnn_conv_s1_mt_supported(const ConvParams& p);
double lnn_conv_s1_mt_efficiency(const ConvParams& p);
int lnn_conv_s1_mt_ksplit(const ConvParams& p, const float* ws, long ws_elems);
int lnn_launch_conv_s1_mt(hipStream_t s, ConvParams& p, float* ws, long ws_elems, const char* name);
// single-launch resolution-doubling kernels (igemm_up2.hip): stride-2 conv dgrad / transposed conv k2s2 forward,
// all eight output parity classes per block
int lnn_launch_up2_dgrad(hipStream_t s, ConvParams& p, const char* name);
int lnn_launch_up2_convT(hipStream_t s, ConvParams& p, const char* name);
// resolution-halving kernels (igemm_down2.hip): stride-2 conv forward / transposed conv k2s2 dgrad
int lnn_launch_down2_conv(hipStream_t s, ConvParams& p, const char* name);
int lnn_launch_down2_convT_dgrad(hipStream_t s, ConvParams& p, const char* name);
// z-streaming stride-2 3x3x3 conv forward (igemm_down2s.hip): C = 32 / 64, M % 64 == 0, even input extents
bool lnn_down2s_supported(const ConvParams& p);
int lnn_down2s_stats_slots(const ConvParams& p);
int lnn_launch_down2s(hipStream_t s, ConvParams& p, const char* name);
