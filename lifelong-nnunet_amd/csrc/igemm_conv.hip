// Implicit-GEMM 3-D convolution family on gfx950 MFMA (v_mfma_f32_32x32x16_f16), channels-last fp16:
// the C-ABI entry points, the kernel selection, and the first-layer (C == 1) kernel.
//
// Production kernels (one file each, selected in the wrappers at the bottom of this file):
//   stride-1 conv fwd / dgrad   igemm_conv_v9.hip (z-streaming: 32 / 64 / 128 input channels, >= 32 planes), igemm_conv_mt.hip (macro
//                               tile + in-block split-K: every other layer with >= 128 input channels), igemm_conv_tile.hip (the "v5" tile
//                               kernel: everything else -- narrow layers on short volumes, odd channel counts).  The tile kernels v7 / v8 of
//                               rounds 1-5 were retired in round 6: the macro-tile kernel is faster on every shape they served
//                               (profiles/r06_kbench_mt_first.txt)
//   stride-2 conv fwd, convT dgrad      igemm_down2.hip          stride-2 dgrad, convT fwd      igemm_up2.hip
// (The generic first-version kernel -- one non-pipelined kernel for every op, stride-2 dgrad as 8 launches -- was the A/B
// baseline of rounds 1-2 and was removed in round 3; profiles/r01_* hold its numbers.)
//
//   OUT[n, os*l+par, m] = bias[m] + sum_{tap} sum_{c} IN[n, IS*l + off(tap) - pad_lo, c] * WP[slot(tap)][m][c]
//
// GEMM view per tap: D[m][voxel] += A[m][c] * B[c][voxel]  (A = weight panel rows, B = input voxels),
// i.e. MFMA rows = output channels, MFMA columns = 32 output voxels, so that each lane ends up with 4
// consecutive output channels of one voxel per accumulator quad -> 8-byte channels-last stores.
#include "lnn_common.h"
#include "igemm_common.h"
#include "igemm_gen.h"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------
// First layer (C == 1): im2col inside LDS, taps are the contraction dimension (27 -> 32).
//   y[n,p,m] = b[m] + sum_tap w[m][tap] x[n, p + tap - 1]
// x tile: (TZ+2)(TY+2)(TX+2) halves.  Weight panel wp[0][Mpad][32].
// ------------------------------------------------------------------------------------------------
// Persistent blocks: grid (nb, N, M / 32); a block walks the 4x8x8-voxel tiles t = blockIdx.x, + nb, ... of ONE sample, so the
// per-(sample, channel) sums of the STATS epilogue (the InstanceNorm statistics of the next op, on the fp16-rounded values
// it stores) stay in registers until the block ends: one partial row per block, p.stats_pws[a][blockIdx.x][n][c].
// The next tile's input halves are fetched into registers before the current tile's MFMAs; the epilogue exchanges
// accumulator quads between the two half-waves (v_permlane32_swap) so that a lane stores 8 consecutive channels (16 bytes).
typedef unsigned c1_uint4 __attribute__((ext_vector_type(4)));

template <bool STATS>
__global__ __launch_bounds__(256, 3) void conv_c1_fwd_kernel(const ConvParams p, int tiles_per_sample) {
    constexpr int TZ = 4, TY = 8, TX = 8, VT = TZ * TY * TX / 128, PZ = TZ + 2, PY = TY + 2, PX = TX + 2, P = PZ * PY * PX;
    constexpr int NL = (P + 255) / 256;
    __shared__ half_t xl[P];
    __shared__ float sred[STATS ? 4 * 2 * 32 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.y;
    const int m0 = blockIdx.z * 32;
    const long xbase_n = (long)n * p.Di * p.Hi * p.Wi;
    const int v = lane & 31, hk = lane >> 5;
    // weight fragments straight from global (tiny, L2 resident)
    half8 a[2];
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16) a[k16] = *reinterpret_cast<const half8*>(p.wp + lnn_panel_off(0, m0 + v, k16 * 16 + hk * 8, 1, 32));
    // tap offsets of this lane's 16 contraction slots
    int toff[2][8];
#pragma unroll
    for (int k16 = 0; k16 < 2; ++k16)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kc = k16 * 16 + hk * 8 + j;
            toff[k16][j] = kc < 27 ? ((kc / 9) * PY + (kc / 3) % 3) * PX + kc % 3 : 0;
        }
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r >> 2) * 8 + hk * 4 + (r & 3);
        bv[r] = (p.bias && m < p.M) ? p.bias[m] : 0.f;
    }
    float ssum[16], ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ssum[r] = ssq[r] = 0.f;

    half_t pre[NL];
    auto fetch = [&](int t) {
        const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, tz = t / (p.tiles_x * p.tiles_y);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int pos = tid + i * 256;
            const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
            const int iz = tz * TZ + pz - 1, iy = ty * TY + py - 1, ix = tx * TX + px - 1;
            half_t val = 0;
            if (pos < P && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                val = p.x[xbase_n + ((long)iz * p.Hi + iy) * p.Wi + ix];
            pre[i] = val;
        }
    };
    int t = blockIdx.x;
    if (t < tiles_per_sample) fetch(t);
    for (; t < tiles_per_sample; t += gridDim.x) {
        __syncthreads();                      // every wave is done reading the previous tile
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (tid + i * 256 < P) xl[tid + i * 256] = pre[i];
        __syncthreads();
        if (t + (int)gridDim.x < tiles_per_sample) fetch(t + gridDim.x);
        const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, tz = t / (p.tiles_x * p.tiles_y);
        const int lz0 = tz * TZ, ly0 = ty * TY, lx0 = tx * TX;
#pragma unroll 1
        for (int vt = 0; vt < VT; ++vt) {      // not unrolled: keeps the STATS variant at three blocks per CU
            const int tile = wave * VT + vt;
            const int z = tile / (TY / 4), y = (tile % (TY / 4)) * 4 + (v >> 3), x = v & 7;
            const int base = (z * PY + y) * PX + x;
            floatx16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                half8 b;
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = xl[base + toff[k16][j]];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k16], b, acc, 0, 0, 0);
            }
            const int lz = lz0 + z, ly = ly0 + y, lx = lx0 + x;
            const bool ok = lz < p.Do && ly < p.Ho && lx < p.Wo;
            half_t* yrow = p.y + ((((long)n * p.Do + (ok ? lz : 0)) * p.Ho + (ok ? ly : 0)) * p.Wo + (ok ? lx : 0)) * p.ld_y;
            half_t r[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = (half_t)(acc[i] + bv[i]);
            if constexpr (STATS) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float f = ok ? (float)r[i] : 0.f;
                    ssum[i] += f;
                    ssq[i] = __builtin_fmaf(f, f, ssq[i]);
                }
            }
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                // quads 2 qp (vdst) and 2 qp + 1 (src): lanes < 32 end with channels 8 (2 qp) .. +7, lanes >= 32 with 8 (2 qp + 1) .. +7
                const int q0 = 2 * qp * 4, q1 = (2 * qp + 1) * 4;
                const half2v a0 = {r[q0 + 0], r[q0 + 1]}, a1 = {r[q0 + 2], r[q0 + 3]};
                const half2v b0 = {r[q1 + 0], r[q1 + 1]}, b1 = {r[q1 + 2], r[q1 + 3]};
                const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, b0), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1), false, false);
                const c1_uint4 o = {s0[0], s1[0], s0[1], s1[1]};
                const int m = m0 + (2 * qp + hk) * 8;
                if (ok && m < p.M) *reinterpret_cast<c1_uint4*>(yrow + m) = o;
            }
        }
    }
    if constexpr (STATS) {
        // butterfly over the 32 voxel lanes of each half-wave, then over the 4 waves through LDS; threads 0..63 write the row
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { ssum[i] += __shfl_xor(ssum[i], o, 64); ssq[i] += __shfl_xor(ssq[i], o, 64); }
        if ((lane & 31) == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = (i >> 2) * 8 + hk * 4 + (i & 3);
                sred[(wave * 2 + 0) * 32 + c] = ssum[i];
                sred[(wave * 2 + 1) * 32 + c] = ssq[i];
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int aidx = tid >> 5, c = tid & 31;
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) tot += sred[(w * 2 + aidx) * 32 + c];
            if (m0 + c < p.M)
                p.stats_pws[(((long)aidx * gridDim.x + blockIdx.x) * gridDim.y + n) * p.M + m0 + c] = tot;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
// debug hook: phase-cycle accumulators for the v3 kernel (tools/kbench.py --phases)
static unsigned long long* g_dbg = nullptr;
extern "C" int lnn_debug_set_phase_buffer(void* dev_ptr_6x_u64) { g_dbg = (unsigned long long*)dev_ptr_6x_u64; return LNN_OK; }

// runtime override for the parity tests (lnn_debug_force_conv_kernel): -1 = automatic selection,
// 5 / 9 / 10 = that stride-1 kernel (5 = tile kernel, 9 = z-streaming, 10 = macro tile) for every layer it supports
int g_force_conv = -1;

// v9 (z-streaming, register-resident weights): the kernel for the 32- / 64-input-channel layers of the two highest
// resolutions.  LNN_CONV_V9=0 forbids it (A/B measurements); lnn_debug_force_conv_kernel(9) forces it wherever supported.
bool use_v9(const ConvParams& p) {
    if (!lnn_conv_s1_v9_supported(p)) return false;
    if (g_force_conv >= 0) return g_force_conv == 9;
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("LNN_CONV_V9");
        v = e ? (e[0] == '1' ? 1 : 0) : -1;
    }
    if (v == 0) return false;
    if (v == 1) return true;
    // automatic: not on short z columns (the column walk has ~4 plane steps of fixed cost per item).  16 since round 6: on the
    // 20-plane levels of a Task005_Prostate-shaped plan (64 / 128 channels @ 20x160x128 / 20x80x64) the z-streaming kernel runs
    // 780-1020 TFLOP/s where the tile kernel ran 420-700 (step 19.26 -> 17.64 ms, gpurun_out/r6f)
    static int min_planes = 0;
    if (!min_planes) { const char* e = getenv("LNN_CONV_V9_MIN_PLANES"); min_planes = e && atoi(e) > 0 ? atoi(e) : 16; }
    return p.Ld >= min_planes;
}

// stride-2 conv forward: z-streaming kernel (igemm_down2s.hip) for 32 / 64 input channels with >= 16 output planes;
// lnn_debug_force_down2_kernel: -1 automatic, 0 the tile kernel (igemm_down2.hip), 1 the streaming kernel wherever supported
int g_force_down2 = -1;
bool use_down2s(const ConvParams& p) {
    if (!lnn_down2s_supported(p)) return false;
    if (g_force_down2 >= 0) return g_force_down2 == 1;
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("LNN_DOWN2S");
        v = e ? (e[0] == '1' ? 1 : 0) : -1;
    }
    if (v == 0) return false;
    if (v == 1) return true;
    return p.Ld >= 16;
}

// Macro-tile kernel with in-block split-K (igemm_conv_mt.hip, round 6): every stride-1 layer with >= 128 input channels that the
// z-streaming kernel does not take -- the deep levels (its bands fill 1.0 / 0.94 of their MFMA columns on levels 3 / 4 of the
// 160x192x160 plan; level 5's 5x6x5 volume fills 0.2 of them and is still 5-8 % faster than the retired split-K tile kernel was), the
// 256-input-channel layers of level 2 (256 -> 128 @ 40x48x40: 0.296 vs 0.309 ms on the retired v8) and, with column bands, the wide
// 20-plane levels of anisotropic plans (lnn_conv_s1_mt_efficiency reports the fill of a shape).
// LNN_CONV_MT=1 / 0 forces / forbids it (A/B measurements); lnn_debug_force_conv_kernel(10) forces it wherever supported.
bool use_mt(const ConvParams& p) {
    if (!lnn_conv_s1_mt_supported(p)) return false;
    if (g_force_conv >= 0) return g_force_conv == 10;
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("LNN_CONV_MT");
        v = e ? (e[0] == '1' ? 1 : 0) : -1;
    }
    if (v == 0) return false;
    if (v == 1) return true;
    return p.C >= 128;
}

}  // namespace

// Small volumes go to the flattened-voxel split-K kernels of igemm_gen.hip (see there).  Which ones: measured per op on the
// 160x192x160 plan (profiles/r04_gen_vs_tile_kernels.txt; output voxels = N x output extents):
//   stride-1 conv fwd / dgrad   512 < voxels <= 4096   (level 4: 73 / 66 -> 54 / 53 us; level 5 stays on the split-K tile kernel: 25 us)
//   stride-2 conv fwd / dgrad   voxels <= 4096         (enc4.0 84 -> 48 us, enc5.0 92 -> 41 us)
//   transposed-conv dgrad       voxels <= 20000        (tu0 51 -> 26, tu1 47 -> 27, tu2 88 -> 53 us)
//   transposed-conv fwd, every weight gradient: never (the tile kernels are as fast or faster)
// LNN_GEN=0 forbids them (A/B measurements); lnn_debug_set_gen_mode(1 / 0) forces / forbids them for every op (parity tests).
static int g_gen_mode = -1;      // lnn_debug_set_gen_mode: -1 automatic, 0 never, 1 wherever supported
extern "C" int lnn_debug_set_gen_mode(int mode) {
    LNN_REQUIRE(mode >= -1 && mode <= 1, "lnn_debug_set_gen_mode: %d is not one of -1, 0, 1", mode);
    g_gen_mode = mode;
    return LNN_OK;
}
bool lnn_gen_prefers(int op, long voxels) {
    if (g_gen_mode >= 0) return g_gen_mode == 1;
    if (g_force_conv >= 0) return false;          // a specialised kernel is pinned by a parity test
    static int off = -1;
    if (off < 0) { const char* e = getenv("LNN_GEN"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off) return false;
    switch (op) {
        case LNN_GEN_OP_CONV_S1: return voxels > 512 && voxels <= 4096;
        case LNN_GEN_OP_CONV_S2: return voxels <= 4096;
        case LNN_GEN_OP_CONVT_DGRAD: return voxels <= 20000;
        default: return false;
    }
}

namespace {
int check_act(const void* ptr, int ld, int C, const char* what) {
    LNN_REQUIRE(ptr != nullptr, "%s: null pointer", what);
    LNN_REQUIRE(lnn_aligned16(ptr), "%s: pointer not 16-byte aligned", what);
    LNN_REQUIRE(C > 0 && C % 8 == 0, "%s: channel count %d must be a positive multiple of 8", what, C);
    LNN_REQUIRE(ld >= C && ld % 8 == 0, "%s: ld %d must be >= C (%d) and a multiple of 8", what, ld, C);
    return LNN_OK;
}

}  // namespace

extern "C" int lnn_debug_force_down2_kernel(int which) {
    LNN_REQUIRE(which >= -1 && which <= 1, "lnn_debug_force_down2_kernel: %d is not one of -1, 0, 1", which);
    g_force_down2 = which;
    return LNN_OK;
}

extern "C" int lnn_debug_force_conv_kernel(int which) {
    LNN_REQUIRE(which == -1 || which == 5 || which == 9 || which == 10, "lnn_debug_force_conv_kernel: %d is not one of -1, 5, 9, 10", which);
    g_force_conv = which;
    return LNN_OK;
}

namespace {
int conv3d_fwd_impl(lnn_stream_t s_, const void* x, const void* x2, int c_a, int ld_x, const void* wp, const float* bias, void* y,
                    int ld_y, int N, int Di, int Hi, int Wi, int C, int K, int stride, float* stats_pws = nullptr,
                    int* stats_slots = nullptr, float* splitk_ws = nullptr, long splitk_elems = 0) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(stride == 1 || stride == 2, "lnn_conv3d_fwd: stride %d unsupported", stride);
    LNN_REQUIRE(N > 0 && Di > 0 && Hi > 0 && Wi > 0, "lnn_conv3d_fwd: bad dims");
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_conv3d_fwd: weight panel null/misaligned");
    LNN_REQUIRE(bias == nullptr || lnn_aligned16(bias), "lnn_conv3d_fwd: bias misaligned");
    if (int e = check_act(y, ld_y, K, "lnn_conv3d_fwd(y)")) return e;
    ConvParams p{};
    p.x = (const half_t*)x; p.wp = (const half_t*)wp; p.bias = bias; p.y = (half_t*)y;
    if (x2) { p.x2 = (const half_t*)x2; p.csplit = c_a; }
    p.ld_x = ld_x; p.ld_y = ld_y; p.N = N; p.Di = Di; p.Hi = Hi; p.Wi = Wi;
    p.Do = (Di - 1) / stride + 1; p.Ho = (Hi - 1) / stride + 1; p.Wo = (Wi - 1) / stride + 1;
    p.C = C; p.M = K; p.Mpad = lnn_round_up(K, 32); p.wtaps = C == 1 ? 1 : 27;
    p.Ld = p.Do; p.Lh = p.Ho; p.Lw = p.Wo; p.os = 1; p.pad_lo = 1; p.accumulate = 0;
    if (C == 1) {
        LNN_REQUIRE(stride == 1, "lnn_conv3d_fwd: C == 1 path supports stride 1 only");
        LNN_REQUIRE(x != nullptr, "lnn_conv3d_fwd: null x");
        p.KCpad = 32;
        p.tiles_z = lnn_cdiv(p.Ld, 4); p.tiles_y = lnn_cdiv(p.Lh, 8); p.tiles_x = lnn_cdiv(p.Lw, 8);
        const int tps = p.tiles_z * p.tiles_y * p.tiles_x;
        int nb = 768 / N;                                  // x N samples: three resident 4-wave blocks per CU on 256 CUs
        if (nb < 1) nb = 1;
        if (nb > tps) nb = tps;
        dim3 grid((unsigned)nb, (unsigned)N, (unsigned)lnn_cdiv(K, 32));
        if (stats_pws && ld_y == K) {
            p.stats_pws = stats_pws;
            p.stats_nblk = nb;
            hipLaunchKernelGGL((conv_c1_fwd_kernel<true>), grid, dim3(256), 0, s, p, tps);
            *stats_slots = nb;
        } else {
            hipLaunchKernelGGL((conv_c1_fwd_kernel<false>), grid, dim3(256), 0, s, p, tps);
        }
        LNN_CHECK_LAUNCH("lnn_conv3d_fwd(C=1)");
        return LNN_OK;
    }
    if (int e = check_act(x, ld_x, x2 ? c_a : C, "lnn_conv3d_fwd(x)")) return e;
    p.KCpad = lnn_round_up(C, 16);
    p.taps.ntaps = 27;
    if (stride == 1) {
        constexpr int PY = 10, PX = 10;
        for (int t = 0; t < 27; ++t) {
            p.taps.pos_off[t] = (unsigned short)(((t / 9) * PY + (t / 3) % 3) * PX + t % 3);
            p.taps.slot[t] = (unsigned char)t;
        }
        if (!use_v9(p) && use_mt(p)) return lnn_launch_conv_s1_mt(s, p, splitk_ws, splitk_elems, "lnn_conv3d_fwd(s1,mt)");
    }
    if (!x2 && lnn_gen_prefers(stride == 1 ? LNN_GEN_OP_CONV_S1 : LNN_GEN_OP_CONV_S2, (long)N * p.Do * p.Ho * p.Wo)) {
        const int k3[3] = {3, 3, 3}, st3[3] = {stride, stride, stride};
        return lnn_gen_conv3d_fwd(s, x, ld_x, wp, bias, y, ld_y, N, Di, Hi, Wi, C, K, k3, st3, splitk_ws, splitk_elems);
    }
    if (stride == 1) {
        p.dbg = g_dbg;
        // fused InstanceNorm statistics (dense output tensor only; the partials must fit the 1024 slots of lnn_instnorm_ws_doubles:
        // true up to 256 CUs -- a larger part falls back to the separate statistics pass)
        if (stats_pws && use_v9(p) && ld_y == K && lnn_conv_s1_v9_stats_slots(p) <= 1024) {
            p.stats_pws = stats_pws;
            const int rc = lnn_launch_conv_s1_v9(s, p, "lnn_conv3d_fwd(s1,v9,stats)");
            *stats_slots = p.stats_nblk;
            return rc;
        }
        if (use_v9(p)) return lnn_launch_conv_s1_v9(s, p, "lnn_conv3d_fwd(s1,v9)");
        return lnn_launch_conv_s1_tile(s, p, "lnn_conv3d_fwd(s1,v5)");
    }
    if (use_down2s(p)) {
        if (stats_pws && ld_y == K && lnn_down2s_stats_slots(p) <= 1024) {      // fused InstanceNorm statistics (dense output tensor only)
            p.stats_pws = stats_pws;
            const int rc = lnn_launch_down2s(s, p, "lnn_conv3d_fwd(s2,down2s,stats)");
            *stats_slots = p.stats_nblk;
            return rc;
        }
        return lnn_launch_down2s(s, p, "lnn_conv3d_fwd(s2,down2s)");
    }
    return lnn_launch_down2_conv(s, p, "lnn_conv3d_fwd(s2,down2)");
}

int check_cat(const void* b, int c_a, int C, int stride, const char* what) {
    LNN_REQUIRE(b != nullptr && lnn_aligned16(b), "%s: second tensor null/misaligned", what);
    LNN_REQUIRE(stride == 1, "%s: stride 1 only", what);
    LNN_REQUIRE(c_a > 0 && c_a < C && c_a % 32 == 0 && (C - c_a) % 8 == 0, "%s: split %d of %d channels must be a multiple of 32", what, c_a, C);
    return LNN_OK;
}
}  // namespace

extern "C" int lnn_conv3d_fwd(lnn_stream_t s, const void* x, int ld_x, const void* wp, const float* bias, void* y,
                              int ld_y, int N, int Di, int Hi, int Wi, int C, int K, int stride) {
    return conv3d_fwd_impl(s, x, nullptr, 0, ld_x, wp, bias, y, ld_y, N, Di, Hi, Wi, C, K, stride);
}

extern "C" int lnn_conv3d_fwd_cat(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* wp,
                                  const float* bias, void* y, int ld_y, int N, int Di, int Hi, int Wi, int C, int K) {
    if (int e = check_cat(x_b, c_a, C, 1, "lnn_conv3d_fwd_cat")) return e;
    LNN_REQUIRE(ld_x >= c_a && ld_x >= C - c_a, "lnn_conv3d_fwd_cat: ld_x %d smaller than a part (%d / %d)", ld_x, c_a, C - c_a);
    return conv3d_fwd_impl(s, x_a, x_b, c_a, ld_x, wp, bias, y, ld_y, N, Di, Hi, Wi, C, K, 1);
}

extern "C" int lnn_conv3d_fwd_in_stats(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* wp,
                                       const float* bias, void* y, int N, int Di, int Hi, int Wi, int C, int K, int stride,
                                       float eps, float* mean, float* rstd, double* ws, float* splitk_ws, long splitk_elems) {
    LNN_REQUIRE(mean && rstd && ws, "lnn_conv3d_fwd_in_stats: null output/workspace");
    if (x_b) {
        if (int e = check_cat(x_b, c_a, C, stride, "lnn_conv3d_fwd_in_stats")) return e;
        LNN_REQUIRE(ld_x >= c_a && ld_x >= C - c_a, "lnn_conv3d_fwd_in_stats: ld_x %d smaller than a part (%d / %d)", ld_x, c_a, C - c_a);
    }
    const long V = (long)((Di - 1) / stride + 1) * ((Hi - 1) / stride + 1) * ((Wi - 1) / stride + 1);
    float* pws = reinterpret_cast<float*>(ws + (size_t)N * K * 3);          // same region lnn_instnorm_stats uses
    int slots = 0;
    if (int e = conv3d_fwd_impl(s, x_a, x_b, c_a, ld_x, wp, bias, y, K, N, Di, Hi, Wi, C, K, stride, pws, &slots, splitk_ws, splitk_elems))
        return e;
    if (slots > 0) return lnn_launch_in_stats_finalize((hipStream_t)s, pws, slots, N, K, V, eps, mean, rstd);
    return lnn_instnorm_stats(s, y, N, V, K, eps, mean, rstd, ws);          // kernel without the fused epilogue: separate pass
}

// conv + InstanceNorm + LeakyReLU of one ConvDropoutNormNonlin block (test_MultiHead_Module.py:287-291) in one call: y = conv(x) + bias,
// mean / rstd of y per (sample, channel), z = LeakyReLU(gamma * (y - mean) * rstd + beta).  Up to lnn_instnorm_small_volume() output
// voxels per sample (the two lowest levels of the 160x192x160 plan) the statistics, their finalize and the normalisation are ONE
// launch behind the convolution (norm_act.hip: in_small_fwd_kernel) instead of three; larger volumes: exactly
// lnn_conv3d_fwd_in_stats + lnn_instnorm_lrelu_fwd.
extern "C" int lnn_conv3d_fwd_in_lrelu(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* wp,
                                       const float* bias, void* y, int N, int Di, int Hi, int Wi, int C, int K, int stride, float eps,
                                       float* mean, float* rstd, const float* gamma, const float* beta, float slope, void* z, int ld_z,
                                       double* ws, float* splitk_ws, long splitk_elems) {
    LNN_REQUIRE(mean && rstd && ws && gamma && beta && z, "lnn_conv3d_fwd_in_lrelu: null output / parameter / workspace");
    LNN_REQUIRE(stride == 1 || stride == 2, "lnn_conv3d_fwd_in_lrelu: stride %d unsupported", stride);
    const long V = (long)((Di - 1) / stride + 1) * ((Hi - 1) / stride + 1) * ((Wi - 1) / stride + 1);
    static int no_small = -1;
    if (no_small < 0) { const char* e = getenv("LNN_IN_SMALL"); no_small = (e && e[0] == '0') ? 1 : 0; }
    if (V > lnn_instnorm_small_volume() || no_small) {
        if (int e = lnn_conv3d_fwd_in_stats(s, x_a, x_b, ld_x, c_a, wp, bias, y, N, Di, Hi, Wi, C, K, stride, eps, mean, rstd, ws,
                                            splitk_ws, splitk_elems)) return e;
        return lnn_instnorm_lrelu_fwd(s, y, z, ld_z, N, V, K, mean, rstd, gamma, beta, slope);
    }
    if (x_b) {
        if (int e = check_cat(x_b, c_a, C, stride, "lnn_conv3d_fwd_in_lrelu")) return e;
        LNN_REQUIRE(ld_x >= c_a && ld_x >= C - c_a, "lnn_conv3d_fwd_in_lrelu: ld_x %d smaller than a part (%d / %d)", ld_x, c_a, C - c_a);
    }
    if (int rc = conv3d_fwd_impl(s, x_a, x_b, c_a, ld_x, wp, bias, y, K, N, Di, Hi, Wi, C, K, stride, nullptr, nullptr, splitk_ws, splitk_elems))
        return rc;
    return lnn_launch_in_small_fwd((hipStream_t)s, y, z, ld_z, N, V, K, eps, gamma, beta, slope, mean, rstd);
}

namespace {
int conv3d_dgrad_impl(lnn_stream_t s_, const void* dy, int ld_dy, const void* wp, void* dx, void* dx2, int c_a, int ld_dx, int N,
                      int Di, int Hi, int Wi, int C, int K, int stride, int accumulate, float* splitk_ws = nullptr,
                      long splitk_elems = 0) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(stride == 1 || stride == 2, "lnn_conv3d_dgrad: stride %d unsupported", stride);
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_conv3d_dgrad: weight panel null/misaligned");
    if (int e = check_act(dy, ld_dy, K, "lnn_conv3d_dgrad(dy)")) return e;
    if (int e = check_act(dx, ld_dx, dx2 ? c_a : C, "lnn_conv3d_dgrad(dx)")) return e;
    const int Do = (Di - 1) / stride + 1, Ho = (Hi - 1) / stride + 1, Wo = (Wi - 1) / stride + 1;
    ConvParams p{};
    if (dx2) { p.y2 = (half_t*)dx2; p.msplit = c_a; }
    // roles: gathered input = dy (K channels), output = dx (C channels); panel wp[slot][C][K]
    p.x = (const half_t*)dy; p.wp = (const half_t*)wp; p.bias = nullptr; p.y = (half_t*)dx;
    p.ld_x = ld_dy; p.ld_y = ld_dx; p.N = N; p.Di = Do; p.Hi = Ho; p.Wi = Wo; p.Do = Di; p.Ho = Hi; p.Wo = Wi;
    p.C = K; p.M = C; p.Mpad = lnn_round_up(C, 32); p.KCpad = lnn_round_up(K, 16); p.wtaps = 27;
    p.accumulate = accumulate;
    if (stride == 1) {
        // dx[q] = sum_d w[d]^T dy[q - d + 1]  -> tap offset d' = 2 - d uses slot d
        p.Ld = Di; p.Lh = Hi; p.Lw = Wi; p.os = 1; p.pad_lo = 1;
        p.taps.ntaps = 27;
        constexpr int PY = 10, PX = 10;
        for (int t = 0; t < 27; ++t) {
            const int dz = t / 9, dyy = (t / 3) % 3, dxx = t % 3;
            p.taps.pos_off[t] = (unsigned short)((dz * PY + dyy) * PX + dxx);
            p.taps.slot[t] = (unsigned char)((2 - dz) * 9 + (2 - dyy) * 3 + (2 - dxx));
        }
        if (!use_v9(p) && use_mt(p)) return lnn_launch_conv_s1_mt(s, p, splitk_ws, splitk_elems, "lnn_conv3d_dgrad(s1,mt)");
    }
    if (!dx2 && lnn_gen_prefers(stride == 1 ? LNN_GEN_OP_CONV_S1 : LNN_GEN_OP_CONV_S2, (long)N * Di * Hi * Wi)) {
        const int k3[3] = {3, 3, 3}, st3[3] = {stride, stride, stride};
        return lnn_gen_conv3d_dgrad(s, dy, ld_dy, wp, dx, ld_dx, N, Di, Hi, Wi, C, K, k3, st3, accumulate, splitk_ws, splitk_elems);
    }
    if (stride == 1) {
        p.dbg = g_dbg;
        if (use_v9(p)) return lnn_launch_conv_s1_v9(s, p, "lnn_conv3d_dgrad(s1,v9)");
        return lnn_launch_conv_s1_tile(s, p, "lnn_conv3d_dgrad(s1,v5)");
    }
    // stride 2: dx[2l+par] = sum over taps d with (par - d + 1) even: dy[l + (par - d + 1)/2]
    //   par = 0 -> d = 1 (offset 0);  par = 1 -> d = 0 (offset +1), d = 2 (offset 0)
    return lnn_launch_up2_dgrad(s, p, "lnn_conv3d_dgrad(s2,up2)");
}
}  // namespace

extern "C" int lnn_conv3d_dgrad(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N,
                                int Di, int Hi, int Wi, int C, int K, int stride, int accumulate) {
    return conv3d_dgrad_impl(s, dy, ld_dy, wp, dx, nullptr, 0, ld_dx, N, Di, Hi, Wi, C, K, stride, accumulate);
}

extern "C" int lnn_conv3d_dgrad_ws(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di, int Hi,
                                   int Wi, int C, int K, int stride, int accumulate, float* splitk_ws, long splitk_elems) {
    return conv3d_dgrad_impl(s, dy, ld_dy, wp, dx, nullptr, 0, ld_dx, N, Di, Hi, Wi, C, K, stride, accumulate, splitk_ws, splitk_elems);
}

// Data gradient of a stride-1 3x3x3 convolution whose input was produced by an InstanceNorm + LeakyReLU block, TOGETHER with pass 1
// of that block's backward (sum g, sum g xhat per (sample, channel), the affine gradients): afterwards dx holds dL/dz of the block
// and ws its sums, exactly as after lnn_conv3d_dgrad_ws + lnn_instnorm_lrelu_bwd_sums(u, dx, ...); lnn_instnorm_lrelu_bwd_apply (or
// the first layer's fused weight gradient) is the second half.  Where a fused instance exists (igemm_conv_v9.hip EPI = 2: 32 -> 32
// channels on long z columns, the highest resolution) the reduce rides the data gradient's epilogue and dz / u are not read again;
// everywhere else this IS the two calls.  u = the block's convolution output (dense, C channels), untouched.
static int g_last_dgrad_reduce_fused = 0;
extern "C" int lnn_debug_last_dgrad_reduce_fused(void) { return g_last_dgrad_reduce_fused; }

extern "C" int lnn_conv3d_dgrad_in_bwd_sums(lnn_stream_t s_, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N,
                                            int Di, int Hi, int Wi, int C, int K, const void* u, const float* mean, const float* rstd,
                                            const float* gamma, const float* beta, float slope, float* dgamma, float* dbeta,
                                            float grad_unscale, double* ws, float* splitk_ws, long splitk_elems) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(u && lnn_aligned16(u) && mean && rstd && gamma && beta && ws, "lnn_conv3d_dgrad_in_bwd_sums: null / misaligned parameter");
    const long V = (long)Di * Hi * Wi;
    ConvParams p{};
    p.x = (const half_t*)dy; p.y = (half_t*)dx; p.ld_x = ld_dy; p.ld_y = ld_dx; p.N = N;
    p.Di = Di; p.Hi = Hi; p.Wi = Wi; p.Do = Di; p.Ho = Hi; p.Wo = Wi; p.C = K; p.M = C; p.os = 1;
    p.Ld = Di; p.Lh = Hi; p.Lw = Wi; p.pad_lo = 1;
    p.red_ld = C;
    static int no_fuse = -1;
    if (no_fuse < 0) { const char* e = getenv("LNN_NO_FUSED_IN_BWD_REDUCE"); no_fuse = (e && e[0] == '1') ? 1 : 0; }
    const bool fused = !no_fuse && wp && dx && dy && lnn_aligned16(dx) && lnn_aligned16(dy) && use_v9(p) && lnn_conv_s1_v9_red_supported(p) &&
                       !lnn_gen_prefers(LNN_GEN_OP_CONV_S1, (long)N * V);
    g_last_dgrad_reduce_fused = fused ? 1 : 0;
    if (!fused) {
        if (int e = conv3d_dgrad_impl(s_, dy, ld_dy, wp, dx, nullptr, 0, ld_dx, N, Di, Hi, Wi, C, K, 1, 0, splitk_ws, splitk_elems)) return e;
        return lnn_instnorm_lrelu_bwd_sums(s_, u, dx, ld_dx, N, V, C, mean, rstd, gamma, beta, slope, dgamma, dbeta, grad_unscale, ws);
    }
    LNN_REQUIRE(lnn_aligned16(wp), "lnn_conv3d_dgrad_in_bwd_sums: weight panel misaligned");
    if (int e = check_act(dy, ld_dy, K, "lnn_conv3d_dgrad_in_bwd_sums(dy)")) return e;
    if (int e = check_act(dx, ld_dx, C, "lnn_conv3d_dgrad_in_bwd_sums(dx)")) return e;
    p.wp = (const half_t*)wp; p.bias = nullptr;
    p.Mpad = lnn_round_up(C, 32); p.KCpad = lnn_round_up(K, 16); p.wtaps = 27;
    p.taps.ntaps = 27;
    for (int t = 0; t < 27; ++t) {           // as conv3d_dgrad_impl: tap offset d' = 2 - d uses slot d
        const int dz = t / 9, dyy = (t / 3) % 3, dxx = t % 3;
        p.taps.pos_off[t] = (unsigned short)((dz * 10 + dyy) * 10 + dxx);
        p.taps.slot[t] = (unsigned char)((2 - dz) * 9 + (2 - dyy) * 3 + (2 - dxx));
    }
    p.dbg = g_dbg;
    float* pws = reinterpret_cast<float*>(ws + (size_t)N * C * 3);           // the region lnn_instnorm_lrelu_bwd_sums uses
    p.stats_pws = pws;
    p.red_u = (const half_t*)u; p.red_mean = mean; p.red_rstd = rstd; p.red_gamma = gamma; p.red_beta = beta; p.red_slope = slope;
    if (int e = lnn_launch_conv_s1_v9(s, p, "lnn_conv3d_dgrad_in_bwd_sums(s1,v9,reduce)")) return e;
    return lnn_launch_in_bwd_sums_raw(s, pws, p.stats_nblk, N, C, mean, rstd, ws, dgamma, dbeta, grad_unscale);
}

// Data gradient of a stride-1 3x3x3 convolution TOGETHER with the whole InstanceNorm + LeakyReLU backward of the block that produced
// its input (small volumes only: V = Di Hi Wi <= lnn_instnorm_small_volume()): afterwards dx holds dL/dz, u -- that block's
// convolution output on entry -- dL/du in place, and dgamma / dbeta (+)= its affine gradients: lnn_conv3d_dgrad_ws(stride 1, no
// accumulate) into dx + lnn_instnorm_lrelu_bwd(u, dx, ...) with the three passes of the latter as one launch.
extern "C" int lnn_conv3d_dgrad_in_bwd(lnn_stream_t s_, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di,
                                       int Hi, int Wi, int C, int K, void* u, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, float slope, float* dgamma, float* dbeta, float grad_unscale, double* ws,
                                       float* splitk_ws, long splitk_elems) {
    LNN_REQUIRE(u && lnn_aligned16(u) && mean && rstd && gamma && beta && ws, "lnn_conv3d_dgrad_in_bwd: null / misaligned parameter");
    const long V = (long)Di * Hi * Wi;
    LNN_REQUIRE(V <= lnn_instnorm_small_volume(), "lnn_conv3d_dgrad_in_bwd: %ld voxels per sample (limit %d): use lnn_conv3d_dgrad_in_bwd_sums",
                V, lnn_instnorm_small_volume());
    LNN_REQUIRE(C % 8 == 0, "lnn_conv3d_dgrad_in_bwd: %d channels (multiple of 8)", C);
    if (int rc = conv3d_dgrad_impl(s_, dy, ld_dy, wp, dx, nullptr, 0, ld_dx, N, Di, Hi, Wi, C, K, 1, 0, splitk_ws, splitk_elems)) return rc;
    return lnn_launch_in_small_bwd((hipStream_t)s_, u, dx, ld_dx, N, V, C, mean, rstd, gamma, beta, slope, ws, dgamma, dbeta, grad_unscale);
}

extern "C" int lnn_conv3d_dgrad_cat(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx_a, void* dx_b, int ld_dx,
                                    int c_a, int N, int Di, int Hi, int Wi, int C, int K, int accumulate) {
    if (int e = check_cat(dx_b, c_a, C, 1, "lnn_conv3d_dgrad_cat")) return e;
    LNN_REQUIRE(ld_dx >= c_a && ld_dx >= C - c_a, "lnn_conv3d_dgrad_cat: ld_dx %d smaller than a part (%d / %d)", ld_dx, c_a, C - c_a);
    return conv3d_dgrad_impl(s, dy, ld_dy, wp, dx_a, dx_b, c_a, ld_dx, N, Di, Hi, Wi, C, K, 1, accumulate);
}

extern "C" int lnn_conv3d_dgrad_cat_ws(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx_a, void* dx_b, int ld_dx,
                                       int c_a, int N, int Di, int Hi, int Wi, int C, int K, int accumulate, float* splitk_ws,
                                       long splitk_elems) {
    if (int e = check_cat(dx_b, c_a, C, 1, "lnn_conv3d_dgrad_cat_ws")) return e;
    LNN_REQUIRE(ld_dx >= c_a && ld_dx >= C - c_a, "lnn_conv3d_dgrad_cat_ws: ld_dx %d smaller than a part (%d / %d)", ld_dx, c_a, C - c_a);
    return conv3d_dgrad_impl(s, dy, ld_dy, wp, dx_a, dx_b, c_a, ld_dx, N, Di, Hi, Wi, C, K, 1, accumulate, splitk_ws, splitk_elems);
}

namespace {
int convT3d_k2s2_fwd_impl(lnn_stream_t s_, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N, int D, int H, int W, int C,
                          int K, float* ws, long ws_elems) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_convT3d_k2s2_fwd: weight panel null/misaligned");
    if (int e = check_act(x, ld_x, C, "lnn_convT3d_k2s2_fwd(x)")) return e;
    if (int e = check_act(y, ld_y, K, "lnn_convT3d_k2s2_fwd(y)")) return e;
    if (lnn_gen_prefers(LNN_GEN_OP_CONVT_FWD, (long)N * D * H * W * 8)) {
        const int st3[3] = {2, 2, 2};
        return lnn_gen_convT3d_fwd(s, x, ld_x, wp, y, ld_y, N, D, H, W, C, K, st3, ws, ws_elems);
    }
    ConvParams p{};
    p.x = (const half_t*)x; p.wp = (const half_t*)wp; p.y = (half_t*)y; p.ld_x = ld_x; p.ld_y = ld_y;
    p.N = N; p.Di = D; p.Hi = H; p.Wi = W; p.Do = 2 * D; p.Ho = 2 * H; p.Wo = 2 * W;
    p.C = C; p.M = K; p.Mpad = lnn_round_up(K, 32); p.KCpad = lnn_round_up(C, 16); p.wtaps = 8;
    p.Ld = D; p.Lh = H; p.Lw = W; p.os = 2; p.pad_lo = 0;
    return lnn_launch_up2_convT(s, p, "lnn_convT3d_k2s2_fwd(up2)");
}
}  // namespace

extern "C" int lnn_convT3d_k2s2_fwd(lnn_stream_t s, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N,
                                    int D, int H, int W, int C, int K) {
    return convT3d_k2s2_fwd_impl(s, x, ld_x, wp, y, ld_y, N, D, H, W, C, K, nullptr, 0);
}
extern "C" int lnn_convT3d_k2s2_fwd_ws(lnn_stream_t s, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N,
                                       int D, int H, int W, int C, int K, float* splitk_ws, long splitk_elems) {
    return convT3d_k2s2_fwd_impl(s, x, ld_x, wp, y, ld_y, N, D, H, W, C, K, splitk_ws, splitk_elems);
}

namespace {
int convT3d_k2s2_dgrad_impl(lnn_stream_t s_, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int D, int H, int W,
                            int C, int K, int accumulate, float* ws, long ws_elems) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_convT3d_k2s2_dgrad: weight panel null/misaligned");
    if (int e = check_act(dy, ld_dy, K, "lnn_convT3d_k2s2_dgrad(dy)")) return e;
    if (int e = check_act(dx, ld_dx, C, "lnn_convT3d_k2s2_dgrad(dx)")) return e;
    if (lnn_gen_prefers(LNN_GEN_OP_CONVT_DGRAD, (long)N * D * H * W)) {
        const int st3[3] = {2, 2, 2};
        return lnn_gen_convT3d_dgrad(s, dy, ld_dy, wp, dx, ld_dx, N, D, H, W, C, K, st3, accumulate, ws, ws_elems);
    }
    ConvParams p{};
    // dx[l, c] = sum_d sum_k dy[2l + d, k] W[c, k, d]: gathered input = dy with stride 2, 8 taps
    p.x = (const half_t*)dy; p.wp = (const half_t*)wp; p.y = (half_t*)dx; p.ld_x = ld_dy; p.ld_y = ld_dx;
    p.N = N; p.Di = 2 * D; p.Hi = 2 * H; p.Wi = 2 * W; p.Do = D; p.Ho = H; p.Wo = W;
    p.C = K; p.M = C; p.Mpad = lnn_round_up(C, 32); p.KCpad = lnn_round_up(K, 16); p.wtaps = 8;
    p.Ld = D; p.Lh = H; p.Lw = W; p.os = 1; p.pad_lo = 0; p.accumulate = accumulate;
    if (use_down2s(p)) return lnn_launch_down2s(s, p, "lnn_convT3d_k2s2_dgrad(down2s)");
    return lnn_launch_down2_convT_dgrad(s, p, "lnn_convT3d_k2s2_dgrad(down2)");
}
}  // namespace

extern "C" int lnn_convT3d_k2s2_dgrad(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx,
                                      int N, int D, int H, int W, int C, int K, int accumulate) {
    return convT3d_k2s2_dgrad_impl(s, dy, ld_dy, wp, dx, ld_dx, N, D, H, W, C, K, accumulate, nullptr, 0);
}
extern "C" int lnn_convT3d_k2s2_dgrad_ws(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx,
                                         int N, int D, int H, int W, int C, int K, int accumulate, float* splitk_ws, long splitk_elems) {
    return convT3d_k2s2_dgrad_impl(s, dy, ld_dy, wp, dx, ld_dx, N, D, H, W, C, K, accumulate, splitk_ws, splitk_elems);
}
