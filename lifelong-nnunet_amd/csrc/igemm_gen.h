// Parameter blocks of the generic-geometry kernels (igemm_gen.hip).  Internal.
#pragma once
#include "lnn_common.h"

// A tap as ONE aligned 32-bit word (a byte-struct table in the kernel arguments is fetched with per-lane vector loads and a
// vmcnt(0) in front of every use; words are scalar loads): bytes = {dz + 8, dy + 8, dx + 8, weight-panel slot}
typedef unsigned GenTap;
static inline GenTap gen_tap(int dz, int dy, int dx, int slot) {
    return (unsigned)(dz + 8) | ((unsigned)(dy + 8) << 8) | ((unsigned)(dx + 8) << 16) | ((unsigned)slot << 24);
}

struct GenParams {
    const half_t* x = nullptr;        // gathered operand
    const half_t* wp = nullptr;       // blocked weight panel [Mpad/32][KCpad/16][wtaps][32][16]
    const float* bias = nullptr;
    half_t* y = nullptr;
    float* scratch = nullptr;         // split-K partial sums [ksplit][output voxel][Mpad]
    int ld_x = 0, ld_y = 0;
    int N = 0, Di = 0, Hi = 0, Wi = 0, Do = 0, Ho = 0, Wo = 0, Ld = 0, Lh = 0, Lw = 0;
    int soz = 1, soy = 1, sox = 1, siz = 1, siy = 1, six = 1;
    int C = 0, M = 0, Mpad = 0, KCpad = 0, wtaps = 0;
    int nclass = 0;
    int accumulate = 0, ksplit = 1;
    int vgroups = 0, mgroups = 0, nbpc = 0, xcd_order = 0;
    unsigned x_bytes = 0, wp_bytes = 0;
    GenTap taps[27] = {};
    unsigned cls_first[9] = {};       // taps of class k: [cls_first[k], cls_first[k + 1])
    unsigned cls_par[8] = {};         // output parity of class k: bytes {z, y, x}
};

struct GenWParams {
    const half_t* p = nullptr;        // operand taken at the loop voxel (rows of the panel)
    const half_t* q = nullptr;        // gathered operand (columns of the panel)
    float* dwp = nullptr;             // fp32 panel [wtaps][Mpad][Cpad]
    float* parts = nullptr;           // deterministic mode: one panel copy per voxel part, added in order afterwards
    long parts_elems = 0, part_stride = 0;
    int ld_p = 0, ld_q = 0;
    int N = 0, Ld = 0, Lh = 0, Lw = 0, Qd = 0, Qh = 0, Qw = 0;
    int siz = 1, siy = 1, six = 1;
    int M = 0, C = 0, Mpad = 0, Cpad = 0;
    int ntaps = 0, wtaps = 0, tgroups = 0, vparts = 1;
    unsigned p_bytes = 0, q_bytes = 0;
    GenTap taps[27] = {};
};

// tap tables / parity classes of kind 0 conv forward, 1 conv data gradient, 2 transposed-conv forward, 3 transposed-conv data
// gradient for kernel extents k[3] (1 or 3; transposed conv: = stride) and strides st[3] (1 or 2); returns the number of taps
int lnn_gen_geometry(GenParams& p, int kind, const int k[3], const int st[3]);
// ws / ws_elems: optional fp32 scratch for split-K (small volumes)
int lnn_launch_gen(hipStream_t s, GenParams& p, float* ws, long ws_elems, const char* name);
int lnn_launch_gen_wgrad(hipStream_t s, GenWParams& p, const char* name);

// host entry points (hipStream_t flavour of the C-ABI lnn_*_g functions; also what the isotropic entry points hand their small
// volumes to, see lnn_gen_prefers)
int lnn_gen_conv3d_fwd(hipStream_t s, const void* x, int ld_x, const void* wp, const float* bias, void* y, int ld_y, int N, int Di,
                       int Hi, int Wi, int C, int K, const int k[3], const int st[3], float* ws, long ws_elems);
int lnn_gen_conv3d_dgrad(hipStream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di, int Hi, int Wi,
                         int C, int K, const int k[3], const int st[3], int accumulate, float* ws, long ws_elems);
int lnn_gen_convT3d_fwd(hipStream_t s, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N, int D, int H, int W, int C,
                        int K, const int st[3], float* ws, long ws_elems);
int lnn_gen_convT3d_dgrad(hipStream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int D, int H, int W,
                          int C, int K, const int st[3], int accumulate, float* ws, long ws_elems);
int lnn_gen_conv3d_wgrad(hipStream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int Di, int Hi, int Wi,
                         int C, int K, const int k[3], const int st[3], float* parts, long parts_elems);
int lnn_gen_convT3d_wgrad(hipStream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D, int H, int W,
                          int C, int K, const int st[3], float* parts, long parts_elems);
// true when an ISOTROPIC layer of this kind and size (output voxels, N x extents) should run on the generic kernels: the volumes
// the tile kernels cannot fill the chip with, per op as measured (rule and numbers in igemm_conv.hip).
// lnn_debug_set_gen_mode(1 / 0) forces / forbids the generic kernels for every isotropic op (parity tests); LNN_GEN=0 forbids.
enum { LNN_GEN_OP_CONV_S1 = 0, LNN_GEN_OP_CONV_S2 = 1, LNN_GEN_OP_CONVT_FWD = 2, LNN_GEN_OP_CONVT_DGRAD = 3, LNN_GEN_OP_WGRAD = 4 };
bool lnn_gen_prefers(int op, long voxels);
