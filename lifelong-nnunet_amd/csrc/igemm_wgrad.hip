// Weight-gradient implicit GEMM on gfx950 MFMA: the contraction runs over VOXELS, so both operands
// are "transposed" relative to their channels-last storage.  The transposition is done by the LDS
// hardware transpose read ds_read_b64_tr_b16 (gfx950), never by scalar LDS traffic.
//
//   DWP[slot(tap)][m][c] += sum_{n, l} P[n, l, m] * Q[n, IS*l + off(tap) - pad_lo, c]
//
//   conv3d      : P = dy (at output voxel l),  Q = x  (gathered),  IS = stride, 27 taps
//   convT k2s2  : P = x  (at input voxel l),   Q = dy (gathered),  IS = 2,      8 taps
//
// Block = 256 threads = 4 waves, one (32 m) x (32 c) panel tile for ALL taps; wave w owns taps
// w, w+4, ... (<= 7 accumulators of 16 VGPRs).  Each block walks `tiles_per_block` spatial tiles
// of TZ x TY x 8 loop voxels, accumulating in registers, and finishes with fp32 atomics into the
// packed panel (coalesced: c is the fastest index).
#include "lnn_common.h"
#include "igemm_gen.h"
#include <cstdlib>

namespace {

struct WTapTable {
    int ntaps;
    unsigned short pos_off[27];
    unsigned char slot[27];
};

struct WgradParams {
    const half_t* p;
    const half_t* q;
    const half_t* q2 = nullptr;       // second tensor of a channel concatenation of Q (channels >= csplit), same ld_q
    int csplit = 0x7fffffff;
    float* dwp;
    int ld_p, ld_q;
    int N, Ld, Lh, Lw, Qd, Qh, Qw;
    int M, C, Mpad, Cpad;
    int tiles_z, tiles_y, tiles_x, tiles_total, tiles_per_block;
    int pad_lo;
    // deterministic mode (lnn_*_wgrad_det): every writer of a block stores its partial sums to its own copy of the panel,
    // parts[(tile chunk * part_writers + writer) * part_stride + panel index]; wg_reduce_parts adds the copies in order
    float* parts = nullptr;
    long parts_elems = 0, part_stride = 0;
    int part_writers = 1;
    int xcd_group = 0, xcd_panels = 0;   // xcd_group > 0: 1-D grid in XCD-aware order over (tile chunk, panel), see wg_block()
    int debug = 0;                    // LNN_WGRAD_DEBUG (measurements only): 1 = skip the epilogue atomics, 4 = phase timers
    unsigned long long* dbgbuf = nullptr;   // LNN_WGRAD_PHASEBUF: 6 x u64 {issue, mfma, barrier1, store, barrier2, tiles}
    // first layer with the InstanceNorm / LeakyReLU backward folded in (lnn_conv3d_wgrad_c1_in_bwd): p = y (the conv output, NOT
    // overwritten), gz = dL/dz; dy = gamma*rstd * (g - s1/V - xhat * s2/V) is rebuilt per tile from the sums of the reduce pass
    const half_t* gz = nullptr;
    int ld_gz = 0;
    const float *in_mean = nullptr, *in_rstd = nullptr, *in_gamma = nullptr, *in_beta = nullptr;
    const double* in_sums = nullptr;       // [(n * M + m) * 3 + {0, 1}] = sum g, sum g * xhat
    float in_slope = 0.f;
    WTapTable taps;
};

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ half4 lds_tr16(const char* addr) {
    fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4_t*)(addr));
    half4 o;
    __builtin_memcpy(&o, &r, 8);
    return o;
}

// panel accumulation: coalesced fp32 atomics (default) or, in deterministic mode, a plain store into this writer's panel copy
__device__ __forceinline__ void wg_out(const WgradParams& p, int chunk, int writer, long idx, float v) {
    if (p.parts) p.parts[((long)chunk * p.part_writers + writer) * p.part_stride + idx] = v;
    else atomicAdd(p.dwp + idx, v);
}
// Block -> (tile chunk, panel).  Default: a (chunks, panels) grid.  XCD-aware order (xcd_group > 0, 1-D grid): the hardware
// deals consecutive workgroups round-robin to the 8 XCDs, each with its own 4 MB L2.  Blocks that walk the SAME tiles for
// different panels are "siblings"; groups of xcd_group siblings (consecutive panels of one chunk) get consecutive slots of ONE
// XCD -- they start together, stream the same operand tiles at the same time and the second to ask finds the lines (and the
// other half of a 128-byte line of a 64-channel tensor) in that L2.  The groups themselves go round-robin over the XCDs, so
// every XCD gets the same share whatever the chunk count.  In the (chunks, panels) grid two siblings are `chunks` workgroups
// apart: on different XCDs unless chunks % 8 == 0.
struct WgBlock { int chunk, panel; };
__device__ __forceinline__ WgBlock wg_block(const WgradParams& p) {
    if (p.xcd_group == 0) return {(int)blockIdx.x, (int)blockIdx.y};
    const int l = blockIdx.x, s = l >> 3, g = p.xcd_group, gpc = p.xcd_panels / g;   // gpc = groups per chunk
    const int u = (s / g) * 8 + (l & 7);                                               // group index
    return {u / gpc, (u % gpc) * g + s % g};                                           // chunk >= chunks: nothing to do
}
__global__ __launch_bounds__(256) void wg_reduce_parts_kernel(const float* __restrict__ parts, int nslots, long slot_elems,
                                                              float* __restrict__ panel) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < slot_elems; i += (long)gridDim.x * 256 * 4) {
        floatx4 a = *reinterpret_cast<const floatx4*>(parts + i);
        for (int k = 1; k < nslots; ++k) a += *reinterpret_cast<const floatx4*>(parts + (long)k * slot_elems + i);
        floatx4* o = reinterpret_cast<floatx4*>(panel + i);
        *o = *o + a;
    }
}

// ------------------------------------------------------------------------------------------------
// Stride-2 gathers (wgrad of the stride-2 3x3x3 conv, EXT = 3, pad 1; wgrad of the 2x2x2 stride-2 transposed conv,
// EXT = 2, pad 0):  DWP[tap][m][c] += sum_l P[l, m] * Q[2l + d - pad, c].
// The generic kernel staged a 5x9x17-position Q tile per 64 loop voxels and per 32 P channels and read it with a
// 128-byte lane stride (2-way bank conflicts); these layers are bound by the Q read (4x the P operand), so:
//   * the Q tile is stored in LDS as EIGHT PARITY SUB-TILES (z, y, x parity of the position): tap d reads sub-tile
//     (d & 1 per dimension) at loop coordinate l + (d >> 1) with UNIT stride, i.e. exactly the conflict-free
//     linear-64-byte-row pattern of the stride-1 kernels;
//   * one block owns TWO 32-row P panels (64 P channels) x one 32-column Q panel: the Q tile is read from HBM once
//     per 64 output rows;
//   * 512 threads (8 waves, taps wave, wave+8, ...: <= 4 taps x 2 panels = 8 accumulators), one block per CU,
//     register prefetch of the next tile (global loads issued before the MFMAs of the current tile, written to LDS between two barriers).
// ------------------------------------------------------------------------------------------------
template <int EXT>
__global__ __launch_bounds__(512, 2) void igemm_wgrad_s2_v2_kernel(const WgradParams p) {
    constexpr int TY = 8, TX = 8, TV = 64;                         // 1 x 8 x 8 loop voxels
    constexpr int PZ = EXT, PY = 2 * (TY - 1) + EXT, PX = 2 * (TX - 1) + EXT, P = PZ * PY * PX;
    constexpr int QZ = 1 + (EXT == 3), QY = TY + (EXT == 3), QX = TX + (EXT == 3);
    constexpr int ST = QZ * QY * QX * 64;                          // bytes per parity sub-tile
    constexpr int QB = 8 * ST;
    constexpr int NTAP = EXT * EXT * EXT, TPW = (NTAP + 7) / 8, MP = 2, NT = 512;
    constexpr int QN = (P * 4 + NT - 1) / NT, PN = (TV * MP * 4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ql = smem;        // [8 parity sub-tiles][QZ][QY][QX][64 B]
    char* const pl = smem + QB;   // [MP][TV][64 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpanels = p.Cpad / 32;
    const WgBlock blk = wg_block(p);
    const int m0 = (blk.panel / cpanels) * 32 * MP, c0 = (blk.panel % cpanels) * 32;
    const int hk = lane >> 5, cb = 16 * ((lane >> 4) & 1), sj = (lane & 15) >> 2, sq = lane & 3;
    const int chb = (cb + 4 * sq) * 2;
    const int p_addr = (8 * hk + sj) * 64 + chb;
    const int q_lane = (hk * QX + sj) * 64 + chb;

    floatx16 acc[TPW][MP];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int a = 0; a < MP; ++a)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][a][j] = 0.f;

    const int t_begin = blk.chunk * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.tiles_total);
    if (t_begin >= t_end) return;

    int qrel[QN], qlds[QN], prel[PN];
#pragma unroll
    for (int i = 0; i < QN; ++i) {
        const int idx = min(i * NT + tid, P * 4 - 1);
        const int pos = idx >> 2, c8 = idx & 3;
        const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
        qrel[i] = ((pz * p.Qh + py) * p.Qw + px) * p.ld_q + c8 * 8;
        const int par = ((pz & 1) * 2 + (py & 1)) * 2 + (px & 1);
        qlds[i] = par * ST + (((pz >> 1) * QY + (py >> 1)) * QX + (px >> 1)) * 64 + c8 * 16;
    }
#pragma unroll
    for (int i = 0; i < PN; ++i) {
        const int idx = i * NT + tid;                 // (mp, voxel, c8)
        const int c8 = idx & 3, vox = (idx >> 2) % TV, mp = idx / (TV * 4);
        prel[i] = ((vox / TX) * p.Lw + vox % TX) * p.ld_p + mp * 32 + c8 * 8;
    }

    half8 qr[QN], pr[PN];
    unsigned qok = 0, pok = 0;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_tile = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int lz0 = t % p.Ld; t /= p.Ld;          // TZ = 1: one loop z-plane per tile
        const int n = t;
        const int ly0 = ty * TY, lx0 = tx * TX;
        const int iz0 = 2 * lz0 - p.pad_lo, iy0 = 2 * ly0 - p.pad_lo, ix0 = 2 * lx0 - p.pad_lo;
        const long qbase = ((((long)n * p.Qd + iz0) * p.Qh + iy0) * p.Qw + ix0) * p.ld_q + c0;
        const long pbase = ((((long)n * p.Ld + lz0) * p.Lh + ly0) * p.Lw + lx0) * p.ld_p + m0;
        const bool interior = iz0 >= 0 && iy0 >= 0 && ix0 >= 0 && iz0 + PZ <= p.Qd && iy0 + PY <= p.Qh && ix0 + PX <= p.Qw &&
                              ly0 + TY <= p.Lh && lx0 + TX <= p.Lw && c0 + 32 <= p.C && m0 + 32 * MP <= p.M;
        if (interior) {
            const half_t* qp = p.q + qbase;
            const half_t* pp = p.p + pbase;
#pragma unroll
            for (int i = 0; i < QN; ++i) qr[i] = *reinterpret_cast<const half8*>(qp + qrel[i]);
#pragma unroll
            for (int i = 0; i < PN; ++i) pr[i] = *reinterpret_cast<const half8*>(pp + prel[i]);
            qok = 0xFFFFFFFFu; pok = 0xFFFFFFFFu;
        } else {
            qok = 0; pok = 0;
#pragma unroll
            for (int i = 0; i < QN; ++i) {
                const int idx = min(i * NT + tid, P * 4 - 1);
                const int pos = idx >> 2, c8 = idx & 3;
                const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
                const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
                const bool ok = (unsigned)iz < (unsigned)p.Qd && (unsigned)iy < (unsigned)p.Qh && (unsigned)ix < (unsigned)p.Qw &&
                                c0 + c8 * 8 < p.C;
                qr[i] = *reinterpret_cast<const half8*>(p.q + (ok ? qbase + qrel[i] : 0));
                qok |= (ok ? 1u : 0u) << i;
            }
#pragma unroll
            for (int i = 0; i < PN; ++i) {
                const int idx = i * NT + tid;
                const int c8 = idx & 3, vox = (idx >> 2) % TV, mp = idx / (TV * 4);
                const bool ok = ly0 + vox / TX < p.Lh && lx0 + vox % TX < p.Lw && m0 + mp * 32 + c8 * 8 < p.M;
                pr[i] = *reinterpret_cast<const half8*>(p.p + (ok ? pbase + prel[i] : 0));
                pok |= (ok ? 1u : 0u) << i;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < QN; ++i)
            if (i * NT + tid < P * 4) *reinterpret_cast<half8*>(ql + qlds[i]) = ((qok >> i) & 1u) ? qr[i] : zero8;
#pragma unroll
        for (int i = 0; i < PN; ++i) *reinterpret_cast<half8*>(pl + (i * NT + tid) * 16) = ((pok >> i) & 1u) ? pr[i] : zero8;
    };

    int tapaddr[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int tap = min(__builtin_amdgcn_readfirstlane(wave) + 8 * ti, NTAP - 1);
        const int dz = EXT == 3 ? tap / 9 : tap >> 2, dy = EXT == 3 ? (tap / 3) % 3 : (tap >> 1) & 1, dx = EXT == 3 ? tap % 3 : tap & 1;
        const int par = ((dz & 1) * 2 + (dy & 1)) * 2 + (dx & 1);
        tapaddr[ti] = q_lane + par * ST + (((dz >> 1) * QY + (dy >> 1)) * QX + (dx >> 1)) * 64;
    }
    load_tile(t_begin);
    store_tile();
    __syncthreads();
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool more = tile + 1 < t_end;
        if (more) load_tile(tile + 1);
#pragma unroll
        for (int ch = 0; ch < TV / 16; ++ch) {
            const int pimm = ch * 1024, qimm = 2 * ch * QX * 64;     // chunk rows 2ch, 2ch+1 (+hk in the lane address)
            half8 a[MP];
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                const half4 a0 = lds_tr16(pl + mp * TV * 64 + pimm + p_addr), a1 = lds_tr16(pl + mp * TV * 64 + pimm + 256 + p_addr);
                a[mp] = half8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            }
#pragma unroll
            for (int ti = 0; ti < TPW; ++ti) {
                if (wave + 8 * ti < NTAP) {
                    const half4 b0 = lds_tr16(ql + qimm + tapaddr[ti]), b1 = lds_tr16(ql + qimm + 256 + tapaddr[ti]);
                    const half8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                    for (int mp = 0; mp < MP; ++mp) acc[ti][mp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mp], b, acc[ti][mp], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (more) store_tile();
        __syncthreads();
    }
    const int c = c0 + (lane & 31);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int tap = wave + 8 * ti;
        if (tap < NTAP) {
            const long pbase_ = (long)tap * p.Mpad * p.Cpad;
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                if (m0 + mp * 32 >= p.Mpad) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + mp * 32 + 8 * (r >> 2) + 4 * hk + (r & 3);
                    wg_out(p, blk.chunk, 0, pbase_ + (long)m * p.Cpad + c, acc[ti][mp][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// First layer (C == 1): DWP[0][m][tap] += sum_l dy[l, m] * x[l + tap - 1]     (tap padded to 32)
// A = dy^T by transpose reads, B[voxel][tap] gathered from a single-channel LDS tile.
// ------------------------------------------------------------------------------------------------
// FUSED: the operand tile is dy of the first block's InstanceNorm + LeakyReLU, rebuilt from y and dL/dz while it is staged (the
// normalisation backward's apply pass -- read dz, read y, write dy -- and this kernel's read of dy disappear: the first layer has
// no data gradient, so nothing else needs dy).
template <int TZ, int TY, bool FUSED>
__global__ __launch_bounds__(256) void wgrad_c1_kernel(const WgradParams p) {
    constexpr int TX = 8, TV = TZ * TY * TX, PZ = TZ + 2, PY = TY + 2, PX = TX + 2, P = PZ * PY * PX;
    constexpr int ROWB = 80, XN = (P + 255) / 256, PN = TV * 4 / 256;
    __shared__ __attribute__((aligned(16))) char pl[TV * ROWB];
    __shared__ half_t xl[P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WgBlock blk = wg_block(p);
    const int m0 = blk.panel * 32;
    const int hk = lane >> 5, cb = 16 * ((lane >> 4) & 1), sj = (lane & 15) >> 2, sq = lane & 3;
    const int p_lane = (8 * hk + sj) * ROWB + (cb + 4 * sq) * 2;
    const int tapn = lane & 31;
    const int toff = tapn < 27 ? ((tapn / 9) * PY + (tapn / 3) % 3) * PX + tapn % 3 : 0;
    floatx16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const int t_begin = blk.chunk * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.tiles_total);
    if (t_begin >= t_end) return;
    // register prefetch of the next tile (round 2): the kernel is a streaming reduction over dy (4 MFMAs per wave and tile), and
    // load -> barrier -> compute -> barrier without overlap left it at 1.7 TB/s
    half_t xr[XN];
    half8 pr[PN];
    half8 gr[FUSED ? PN : 1];
    float fsc[8], fsh[8], fca[8], fcb[8];      // FUSED: per-channel constants of this thread's octet (idx & 3 = tid & 3) for sample fn
    int fn = -1;
    // FUSED: dy of the staged tile from (y, dz): same expressions as in_lrelu_seg_bwd_apply_kernel (norm_act.hip).  Out-of-volume
    // voxels carry y = dz = 0 -> their dy would be the constant ca, not 0: the loader's zero fill is applied AFTER the transform
    // through the per-load ok mask kept in mk.
    unsigned mk = 0;
    auto load_consts = [&](int n) {
        const float invV = 1.0f / ((float)p.Ld * (float)p.Lh * (float)p.Lw);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = m0 + (tid & 3) * 8 + e;
            const bool ok = c < p.M;
            const float rs = ok ? p.in_rstd[n * p.M + c] : 0.f, mu = ok ? p.in_mean[n * p.M + c] : 0.f;
            const float ga = ok ? p.in_gamma[c] : 0.f, be = ok ? p.in_beta[c] : 0.f;
            const float m1 = ok ? (float)(p.in_sums[((long)n * p.M + c) * 3 + 0] * (double)invV) : 0.f;
            const float m2 = ok ? (float)(p.in_sums[((long)n * p.M + c) * 3 + 1] * (double)invV) : 0.f;
            fsc[e] = ga * rs;
            fsh[e] = be - mu * fsc[e];
            fca[e] = -fsc[e] * (m1 - mu * rs * m2);
            fcb[e] = -fsc[e] * rs * m2;
        }
    };
    auto load_tile = [&](int tile) {
        int t = tile;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int tz = t % p.tiles_z; t /= p.tiles_z;
        const int n = t;
        const int lz0 = tz * TZ, ly0 = ty * TY, lx0 = tx * TX;
        const long qbase = (long)n * p.Qd * p.Qh * p.Qw;
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int pos = min(i * 256 + tid, P - 1);
            const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
            const int iz = lz0 + pz - 1, iy = ly0 + py - 1, ix = lx0 + px - 1;
            const bool ok = (unsigned)iz < (unsigned)p.Qd && (unsigned)iy < (unsigned)p.Qh && (unsigned)ix < (unsigned)p.Qw;
            const half_t val = p.q[ok ? qbase + ((long)iz * p.Qh + iy) * p.Qw + ix : 0];
            xr[i] = ok ? val : (half_t)0;
        }
        const long pbase = (long)n * p.Ld * p.Lh * p.Lw;
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < PN; ++i) {
            const int idx = i * 256 + tid;
            const int vox = idx >> 2, c8 = idx & 3;
            const int x = vox % TX, y = (vox / TX) % TY, z = vox / (TX * TY);
            const int lz = lz0 + z, ly = ly0 + y, lx = lx0 + x;
            const bool ok = lz < p.Ld && ly < p.Lh && lx < p.Lw && m0 + c8 * 8 < p.M;
            const half8 val = *reinterpret_cast<const half8*>(p.p + (ok ? (pbase + ((long)lz * p.Lh + ly) * p.Lw + lx) * p.ld_p + m0 + c8 * 8 : 0));
            pr[i] = ok ? val : zero8;
            if constexpr (FUSED) {
                mk = i == 0 ? (ok ? 1u : 0u) : (mk | ((ok ? 1u : 0u) << i));
                const half8 g = *reinterpret_cast<const half8*>(p.gz + (ok ? (pbase + ((long)lz * p.Lh + ly) * p.Lw + lx) * p.ld_gz + m0 + c8 * 8 : 0));
                gr[i] = ok ? g : zero8;
            }
        }
        if constexpr (FUSED) {
            if (n != fn) { fn = n; load_consts(n); }       // (a block's tile range rarely crosses a sample)
        }
    };

    load_tile(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                                   // every wave is done with the previous tile image
#pragma unroll
        for (int i = 0; i < XN; ++i)
            if (i * 256 + tid < P) xl[i * 256 + tid] = xr[i];
#pragma unroll
        for (int i = 0; i < PN; ++i) {
            const int idx = i * 256 + tid;
            half8 v = pr[i];
            if constexpr (FUSED) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xf = (float)pr[i][e];
                    const float t = xf * fsc[e] + fsh[e];
                    const float g = (float)gr[i][e] * (t > 0.f ? 1.f : p.in_slope);
                    v[e] = ((mk >> i) & 1u) ? (half_t)(fsc[e] * g + (xf * fcb[e] + fca[e])) : (half_t)0;
                }
            }
            *reinterpret_cast<half8*>(pl + (idx >> 2) * ROWB + (idx & 3) * 16) = v;
        }
        __syncthreads();
        if (tile + 1 < t_end) load_tile(tile + 1);        // in flight during the MFMAs below and the next barrier
        // each wave takes chunks wave, wave+4, ...
        for (int ch = wave; ch < TV / 16; ch += 4) {
            const char* pa = pl + ch * 16 * ROWB + p_lane;
            half4 a0 = lds_tr16(pa), a1 = lds_tr16(pa + 4 * ROWB);
            half8 a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            // B: lane = tap n, k-slot (hk, r, j) <-> voxel u = 8*hk + 4*r + j of the chunk (same map as A)
            const int row = 2 * ch + hk, z = row / TY, y = row % TY;
            const int base = (z * PY + y) * PX + toff;
            half8 b;
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = xl[base + u];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    }
    if (tapn < 27 || p.parts) {           // deterministic mode: the 5 pad columns of every panel copy are written too (zeros)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 8 * (r >> 2) + 4 * hk + (r & 3);
            wg_out(p, blk.chunk, wave, (long)m * 32 + tapn, tapn < 27 ? acc[r] : 0.f);
        }
    }
}

// deterministic mode: size check + the ordered reduction of the panel copies into the panel
int wg_prepare_parts(WgradParams& p, unsigned gridx, int writers, long slot_elems, const char* name) {
    if (!p.parts) return LNN_OK;
    LNN_REQUIRE((long)gridx * writers * slot_elems <= p.parts_elems,
                "%s: deterministic scratch too small (%ld floats needed, %ld given)", name, (long)gridx * writers * slot_elems, p.parts_elems);
    p.part_stride = slot_elems;
    p.part_writers = writers;
    return LNN_OK;
}
// grid of a (chunks x panels) launch: XCD-aware 1-D order (wg_block) in sibling groups of 4 (2) when the panel count allows;
// LNN_WGRAD_XCD=0 keeps the plain (chunks, panels) grid (A/B measurements)
dim3 wg_grid(WgradParams& p, unsigned chunks, unsigned panels) {
    static int xcd = -1;
    if (xcd < 0) { const char* e = getenv("LNN_WGRAD_XCD"); xcd = (e && e[0] == '0') ? 0 : 1; }
    const int g = panels % 4 == 0 ? 4 : (panels % 2 == 0 ? 2 : 1);
    if (!xcd || g == 1) { p.xcd_group = 0; return dim3(chunks, panels); }
    p.xcd_group = g; p.xcd_panels = (int)panels;
    const unsigned groups = chunks * (panels / g);
    return dim3((unsigned)lnn_round_up((int)groups, 8) * g);
}
int wg_reduce_parts(hipStream_t s, const WgradParams& p, unsigned gridx, long slot_elems, const char* name) {
    if (!p.parts) return LNN_OK;
    const long v4 = slot_elems / 4;
    const int blocks = (int)((v4 + 255) / 256 < 2048 ? (v4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(wg_reduce_parts_kernel, dim3(blocks), dim3(256), 0, s, p.parts, (int)(gridx * p.part_writers), slot_elems, p.dwp);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}

// ------------------------------------------------------------------------------------------------
// Stride-1 weight gradient (kernel "v5"; the register-prefetch kernels v1-v4 it grew out of were deleted in round 3).
// Phase timers on the register-prefetch design showed per wave and tile 4.6 k cycles ISSUING the prefetch loads (the wave
// blocks while the CU's L1 absorbs 57 KB at ~18 B/clk), 4.9 k in the MFMA loop, 1.5 k writing the registers to LDS, 0.8 k in
// two barriers: neither HBM nor LDS latency nor the epilogue atomics was the limit, the issue and store phases were.  So:
//   * tiles arrive by DMA (buffer_load ... lds, 16 B per lane straight into the linear [position][64 B] tile image; lanes
//     outside the volume / channel range carry an out-of-range offset and the descriptor zero-fills them = the conv's zero
//     padding), issued one at a time BETWEEN the MFMA groups of the current tile: no staging registers, no LDS store
//     phase, the wave never waits for the address pipe;
//   * ONE barrier per tile (vmcnt(0) + s_barrier), dy tile double-buffered;
//   * one 8-wave block per CU: wave = (tap group 0..3, tile half 0..1), 8 chunks x 7 taps = 56 MFMAs per wave and tile,
//     operand reads software-pipelined two groups ahead and shared between the three dx taps of a row;
//   * a z-ring of input planes: the 6x10x10 halo of a 4x8x8 tile is 2.34x its voxels and L2 keeps none of it (PMC 2.5x the
//     algorithmic bytes without the ring).  A block walks a column of tiles in z, so 2 of the 6 input planes of the next
//     tile are already in LDS: the input tile lives in a ring of 12 plane slots (10x10 positions x 64 B, padded to 7
//     wave-wide DMA rows = 7 KB), only the 4 NEW planes are fetched per tile (6 when the next tile starts a new column),
//     bytes per tile 54.4 -> 41.6 KB.  Out-of-volume planes use a zero-length descriptor.
// ------------------------------------------------------------------------------------------------
// The DMA is issued through inline asm: hipcc's waitcnt pass treats the transposing LDS read (an intrinsic without a
// memory operand) like an LDS store and puts `s_waitcnt vmcnt(0)` in front of EVERY such read once an LDS-DMA load it knows
// about is in flight (seen in the ISA of the builtin version) -- which serialises each DMA against the MFMA loop.  Hidden
// VMEM operations only make compiler-generated vmcnt waits stricter, never wrong (loads return in order).
// rs = raw buffer descriptor {base[31:0], base[47:32], num_records, flags}; lds_addr = byte address of the wave's 1 KB run.
__device__ __forceinline__ void wg_dma16(uint4v rs, unsigned lds_addr, int voffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voffset), "s"(rs)
                 : "memory");       // m0 is reserved by the backend (it never allocates it), so no clobber entry is needed
}
__device__ __forceinline__ void wg_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint4v wg_rsrc_n(const void* base, unsigned num_records) {
    const unsigned long long a = (unsigned long long)base;
    uint4v r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, num_records, 0x00020000u};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]);
    r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]);
    return r;
}

// Round 5 measured the six plane fetches of the next tile through RUNNING scalars (plane pointer, ring offset, plane index set once per
// tile: 161 instead of 226 scalar instructions in the MFMA blocks of a tile): 1-2 % SLOWER on every layer (dec4.0 1.235 / 1.253 vs
// 1.220 / 1.226 ms, enc0.1 0.578 / 0.573 vs 0.562 / 0.562 ms, alternating runs on one box, profiles/r05_wgrad_v5_scalar_work.txt) --
// the descriptor arithmetic below is independent work the scalar unit does beside the MFMAs, the running form a serial chain.
__global__ __launch_bounds__(512, 1) void igemm_wgrad_s1_v5_kernel(const WgradParams p) {
    constexpr int TZ = 4, TY = 8, TX = 8, PY = 10, PX = 10, PP = PY * PX, TPW = 7, NT = 512;
    constexpr int SLOTB = 7 * 1024, NSLOT = 12, QB = NSLOT * SLOTB, PB = 2 * NT * 16;
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), tg = wave & 3, hf = wave >> 2;
    const WgBlock blk = wg_block(p);
    const int m0 = (blk.panel / (p.Cpad / 32)) * 32, c0 = (blk.panel % (p.Cpad / 32)) * 32;
    const half_t* const qsrc = c0 >= p.csplit ? p.q2 : p.q;
    const int cq = c0 >= p.csplit ? c0 - p.csplit : c0;
    const int hk = lane >> 5, cb = 16 * ((lane >> 4) & 1), sj = (lane & 15) >> 2, sq = lane & 3;
    const int chb = (cb + 4 * sq) * 2;
    const int p_lane = QB + (8 * hk + sj) * 64 + chb + hf * 8 * 1024;      // + (dy tile image) * PB + chunk * 1024

    // this wave's operand rows: k = 0, 1 -> (dz, dy) rows 2 tg, 2 tg + 1 with their three dx taps; k = 2 -> tap (2, 2, dx = tg)
    // (a dropped duplicate for tg = 3)
    int rowlane[3], rowdz[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int r = k < 2 ? 2 * tg + k : 8, dx0 = k < 2 ? 0 : min(tg, 2);
        rowdz[k] = r / 3;
        rowlane[k] = (hk * PX + sj) * 64 + chb + ((r % 3) * PX + dx0) * 64;
    }
    auto tap_of = [&](int ti) { return ti < 6 ? (2 * tg + ti / 3) * 3 + ti % 3 : (tg < 3 ? 24 + tg : 27); };

    floatx16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

    const int t_begin = blk.chunk * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.tiles_total);
    if (t_begin >= t_end) return;

    // DMA lane constants.  Input plane: waves 0..6 cover its 400 16-byte pieces (piece = 64 wave + lane).
    const int qpos = (wave * 64 + lane) >> 2, qc8 = lane & 3;
    const int qpy = qpos / PX, qpx = qpos % PX;
    const bool qst = qpos < PP;
    const int qrel = qst ? ((qpy * p.Qw + qpx) * p.ld_q + qc8 * 8) * 2 : OOB;
    const long qplane = (long)p.Qh * p.Qw * p.ld_q;            // elements per input plane
    int prel[2], pco[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = j * NT + tid, vox = idx >> 2, c8 = idx & 3;
        const int x = vox % TX, y = (vox / TX) % TY, z = vox / (TX * TY);
        prel[j] = (((z * p.Lh + y) * p.Lw + x) * p.ld_p + c8 * 8) * 2;
        pco[j] = z | (y << 8) | (x << 16) | (c8 << 24);
    }

    struct Tile {
        const half_t* qorg;    // input position (lz0 - 1, ly0 - 1, lx0 - 1) of sample n, channel cq
        const half_t* porg;    // dy position (lz0, ly0, lx0), channel m0
        int lz0, tz, qv, pxy[2], pv[2];
    };
    const long pplane = (long)p.Lh * p.Lw * p.ld_p;
    auto set_pv = [&](Tile& t) {                               // z extent of the dy tile (the y / x / channel part is per column)
        const bool zin = t.lz0 + TZ <= p.Ld;
#pragma unroll
        for (int j = 0; j < 2; ++j) t.pv[j] = (zin || t.lz0 + (pco[j] & 255) < p.Ld) ? t.pxy[j] : OOB;
    };
    auto prep = [&](int tile) {                                // full decode: first tile of the block / of a column
        Tile t;
        int r = tile;
        t.tz = r % p.tiles_z; r /= p.tiles_z;
        const int tx = r % p.tiles_x; r /= p.tiles_x;
        const int ty = r % p.tiles_y; r /= p.tiles_y;
        const int n = r;
        const int lz0 = t.tz * TZ, ly0 = ty * TY, lx0 = tx * TX;
        t.lz0 = lz0;
        t.qorg = qsrc + ((((long)n * p.Qd + (lz0 - 1)) * p.Qh + (ly0 - 1)) * p.Qw + (lx0 - 1)) * p.ld_q + cq;
        t.porg = p.p + ((((long)n * p.Ld + lz0) * p.Lh + ly0) * p.Lw + lx0) * p.ld_p + m0;
        const bool inner = ly0 >= 1 && lx0 >= 1 && ly0 + TY + 1 <= p.Qh && lx0 + TX + 1 <= p.Qw && c0 + 32 <= p.C;
        if (inner) {
            t.qv = qrel;
        } else {
            const int iy = ly0 - 1 + qpy, ix = lx0 - 1 + qpx;
            const bool ok = qst && (unsigned)iy < (unsigned)p.Qh && (unsigned)ix < (unsigned)p.Qw && c0 + qc8 * 8 < p.C;
            t.qv = ok ? qrel : OOB;
        }
        const bool pin = ly0 + TY <= p.Lh && lx0 + TX <= p.Lw && m0 + 32 <= p.M;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = pin || (ly0 + ((pco[j] >> 8) & 255) < p.Lh && lx0 + ((pco[j] >> 16) & 255) < p.Lw && m0 + (pco[j] >> 24) * 8 < p.M);
            t.pxy[j] = ok ? prel[j] : OOB;
        }
        set_pv(t);
        return t;
    };
    auto advance = [&](const Tile& c, int tile_next) {         // next tile: same column -> a few adds (no divisions)
        if (c.tz + 1 < p.tiles_z) {
            Tile t = c;
            t.tz = c.tz + 1;
            t.lz0 = c.lz0 + TZ;
            t.qorg = c.qorg + TZ * qplane;
            t.porg = c.porg + TZ * pplane;
            set_pv(t);
            return t;
        }
        return prep(tile_next);
    };
    auto dma_q = [&](const Tile& t, int rel, int slot) {       // input plane `rel` (0..5) of tile t -> ring slot
        const int iz = t.lz0 - 1 + rel;
        const uint4v rs = wg_rsrc_n(t.qorg + rel * qplane, (unsigned)iz < (unsigned)p.Qd ? 0x7fffffffu : 0u);
        if (wave < 7) wg_dma16(rs, lds0 + slot * SLOTB + wave * 1024, t.qv);
    };
    auto dma_p = [&](const Tile& t, int j, int img) {
        wg_dma16(wg_rsrc_n(t.porg, 0x7fffffffu), lds0 + QB + img * PB + j * (NT * 16) + wave * 1024, t.pv[j]);
    };
    auto wrap = [](int v) { return v >= NSLOT ? v - NSLOT : v; };
    Tile cur = prep(t_begin);
    int base = 0, img = 0;
#pragma unroll
    for (int rel = 0; rel < 6; ++rel) dma_q(cur, rel, rel);
    dma_p(cur, 0, 0);
    dma_p(cur, 1, 0);
    wg_wait_all();
    __syncthreads();

    // operand addresses of a tile: (row k, plane zq of this wave's tile half) -> ring slot; dy tile image
    int qaddr[3][2], pa;
    auto addr_for = [&](int b, int im, int (&qa)[3][2], int& pav) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int zq = 0; zq < 2; ++zq) qa[k][zq] = rowlane[k] + wrap(wrap(b + 2 * hf + zq + rowdz[k])) * SLOTB;
        pav = p_lane + im * PB;
    };
    addr_for(0, 0, qaddr, pa);

    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    const bool timed = (p.debug & 4) && p.dbgbuf;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool more = tile + 1 < t_end;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (timed) t0 = __builtin_readcyclecounter();
        // (moving this decode behind the first MFMA group -- round 3 -- moved its 320 cycles into the MFMA phase and changed nothing:
        // the loop is bound by the wave's own instruction stream, not by the shared matrix pipe)
        Tile nxt = cur;
        if (more) nxt = advance(cur, tile + 1);
        const bool same_col = nxt.tz != 0;                  // z runs fastest: the next tile continues this column
        const int nin = more ? (same_col ? 4 : 6) : 0, rel0 = same_col ? 2 : 0;
        const int nbase = wrap(base + (same_col ? 4 : 6));
        int qaddr_n[3][2], pa_n;

        if (timed) t1 = __builtin_readcyclecounter();
        constexpr int NCH = 8;
        half4 al[2], ah[2];
        auto rdA = [&](int ch) {
            al[ch & 1] = lds_tr16(smem + pa + ch * 1024);
            ah[ch & 1] = lds_tr16(smem + pa + ch * 1024 + 256);
        };
        uint2v g0[3], g1[3], g2[3];
        auto rdG = [&](int g) {
            const int ch = g / 3, k = g % 3, sl = g % 3;
            const char* qa = smem + qaddr[k][ch / 4] + ((2 * ch) % TY) * PX * 64;
            g0[sl] = __builtin_bit_cast(uint2v, lds_tr16(qa));
            g1[sl] = __builtin_bit_cast(uint2v, lds_tr16(qa + 256));
            if (k < 2) g2[sl] = __builtin_bit_cast(uint2v, lds_tr16(qa + 512));
        };
        rdA(0);
        rdG(0);
        rdG(1);
#pragma unroll
        for (int g = 0; g < 3 * NCH; ++g) {
            const int ch = g / 3, k = g % 3, sl = g % 3;
            if (g + 2 < 3 * NCH) rdG(g + 2);
            if (k == 1 && ch + 1 < NCH) rdA(ch + 1);
            if (g == 12) addr_for(nbase, img ^ 1, qaddr_n, pa_n);     // next tile's addresses, off the critical path
            if (g % 2 == 1 && g / 2 < 8) {                  // 8 DMA issue points, every second group (every group: +-0, round 2)
                const int i = g / 2;
                if (i < 6) {
                    if (i < nin) dma_q(nxt, rel0 + i, wrap(wrap(base + 6 + i)));
                } else if (more) {
                    dma_p(nxt, i - 6, img ^ 1);
                }
            }
            const half8 a = {al[ch & 1][0], al[ch & 1][1], al[ch & 1][2], al[ch & 1][3],
                             ah[ch & 1][0], ah[ch & 1][1], ah[ch & 1][2], ah[ch & 1][3]};
            const uint2v d01 = g0[sl], d23 = g1[sl], d45 = g2[sl];
            const uint4v w0 = {d01[0], d01[1], d23[0], d23[1]};
            if (k < 2) {
                const uint4v w1 = {__builtin_amdgcn_alignbit(d01[1], d01[0], 16), __builtin_amdgcn_alignbit(d23[0], d01[1], 16),
                                   __builtin_amdgcn_alignbit(d23[1], d23[0], 16), __builtin_amdgcn_alignbit(d45[0], d23[1], 16)};
                const uint4v w2 = {d01[1], d23[0], d23[1], d45[0]};
                acc[3 * k + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, w0), acc[3 * k + 0], 0, 0, 0);
                acc[3 * k + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, w1), acc[3 * k + 1], 0, 0, 0);
                acc[3 * k + 2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, w2), acc[3 * k + 2], 0, 0, 0);
            } else {
                acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8, w0), acc[6], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (timed) t2 = __builtin_readcyclecounter();
        wg_wait_all();                   // the next tile's planes / dy tile are complete ...
        if (timed) t3 = __builtin_readcyclecounter();
        __syncthreads();                 // ... and every wave is done reading this tile
        if (timed) {
            const unsigned long long t4 = __builtin_readcyclecounter();
            ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;
        }
        base = nbase;
        img ^= 1;
        cur = nxt;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int zq = 0; zq < 2; ++zq) qaddr[k][zq] = qaddr_n[k][zq];
        pa = pa_n;
    }
    if (timed && lane == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd(p.dbgbuf + i, ph[i]);
        atomicAdd(p.dbgbuf + 5, (unsigned long long)(t_end - t_begin));
    }
    const int c = c0 + (lane & 31);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int tap = tap_of(ti);
        if (tap < 27) {
            const long pbase_ = (long)tap * p.Mpad * p.Cpad;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 8 * (r >> 2) + 4 * hk + (r & 3);
                wg_out(p, blk.chunk, hf, pbase_ + (long)m * p.Cpad + c, acc[ti][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Stride-2 gathers on the v5 recipe (round 3, "s2s"): same operand layout and MFMA loop as igemm_wgrad_s2_v2_kernel above (eight
// parity sub-tiles, two 32-row P panels x one 32-column Q panel per block, taps wave, wave + 8, ...), but
//   * a block walks a z COLUMN of tiles (1 loop plane x 8 x 8 loop voxels each); the Q planes live in an LDS ring of 2 EXT slots:
//     with EXT = 3 plane 2 lz + 1 serves tile lz (tap dz = 2) and tile lz + 1 (dz = 0), so a step fetches 2 planes instead of 3;
//   * tiles arrive by LDS-DMA (buffer_load ... lds), one tile ahead, issued one instruction at a time between the MFMA groups: no
//     staging registers, no LDS store phase, ONE barrier per tile.  The (y, x)-parity de-interleave happens in the DMA: a lane's
//     GLOBAL offset is chosen so that the wave's linear 1 KB run in LDS is already [parity][qy][qx][64 B].  Out-of-volume rows /
//     columns / channels carry an out-of-range offset, out-of-volume planes a zero-length descriptor -> zeros (= the padding);
//   * the register-prefetch kernel spent a global round trip per tile (27 MFMAs per wave = 0.8 us of matrix work against ~3.8 us
//     per tile measured): it was paced by load latency, not by HBM or the matrix pipe.
// ------------------------------------------------------------------------------------------------
template <int EXT>
__global__ __launch_bounds__(512, 2) void igemm_wgrad_s2s_kernel(const WgradParams p) {
    constexpr int TY = 8, TX = 8, TV = 64, MP = 2, NT = 512;
    constexpr int QY = TY + (EXT == 3), QX = TX + (EXT == 3), PY = 2 * (TY - 1) + EXT, PX = 2 * (TX - 1) + EXT;
    constexpr int SUB = QY * QX * 64;                    // bytes of one (y, x)-parity sub-plane
    constexpr int NPI = (4 * SUB + 1023) / 1024;         // DMA instructions (1 KB wave runs) per plane
    constexpr int SLOTB = NPI * 1024, KPW = (NPI + 7) / 8, NSLOT = 2 * EXT;
    constexpr int QB = NSLOT * SLOTB, PB = MP * TV * 64;
    constexpr int NTAP = EXT * EXT * EXT, TPW = (NTAP + 7) / 8;
    constexpr int NIP = 4 * TPW;                         // DMA issue points per tile (one per MFMA group)
    constexpr int DPP = (EXT * KPW + 1 + NIP - 1) / NIP; // DMA instructions per issue point
    constexpr int OOB = (int)0x80000000;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WgBlock blk = wg_block(p);
    const int cpanels = p.Cpad / 32;
    const int m0 = (blk.panel / cpanels) * 32 * MP, c0 = (blk.panel % cpanels) * 32;
    const int hk = lane >> 5, cb = 16 * ((lane >> 4) & 1), sj = (lane & 15) >> 2, sq = lane & 3;
    const int chb = (cb + 4 * sq) * 2;
    const int p_addr = (8 * hk + sj) * 64 + chb;
    const int q_lane = (hk * QX + sj) * 64 + chb;

    floatx16 acc[TPW][MP];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int a = 0; a < MP; ++a)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][a][j] = 0.f;

    const int t_begin = blk.chunk * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.tiles_total);
    if (t_begin >= t_end) return;

    // DMA lane constants.  Q plane: instruction j = k * 8 + wave covers slab pieces [64 j, 64 j + 64); piece = (position, 16-byte part)
    int qrel[KPW], qpk[KPW];
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
        const int sidx = (k * 8 + wave) * 64 + lane, pos = sidx >> 2, c8 = sidx & 3;
        const int par = pos / (QY * QX), rem = pos % (QY * QX), qy = rem / QX, qx = rem % QX;
        const int py = 2 * qy + (par >> 1), px = 2 * qx + (par & 1);
        const bool st = k * 8 + wave < NPI && par < 4 && py < PY && px < PX;
        qrel[k] = st ? ((py * p.Qw + px) * p.ld_q + c8 * 8) * 2 : OOB;
        qpk[k] = st ? (py | (px << 8) | (c8 << 16)) : -1;
    }
    int prel, ppk;
    {
        const int sidx = wave * 64 + lane, c8 = sidx & 3, vox = (sidx >> 2) & 63, mp = sidx >> 8;
        prel = (((vox / TX) * p.Lw + vox % TX) * p.ld_p + mp * 32 + c8 * 8) * 2;
        ppk = (vox / TX) | ((vox % TX) << 8) | ((mp * 32 + c8 * 8) << 16);
    }
    const long qplane = (long)p.Qh * p.Qw * p.ld_q, pplane = (long)p.Lh * p.Lw * p.ld_p;

    struct Tile {
        const half_t* qorg;    // Q position (2 lz - pad, 2 ly0 - pad, 2 lx0 - pad) of sample n, channel c0
        const half_t* porg;    // P position (lz, ly0, lx0), channel m0
        int lz, qv[KPW], pv;
    };
    auto prep = [&](int tile) {                                // full decode: first tile of the block / of a column (z runs fastest)
        Tile t;
        int r = tile;
        t.lz = r % p.Ld; r /= p.Ld;
        const int tx = r % p.tiles_x; r /= p.tiles_x;
        const int ty = r % p.tiles_y; r /= p.tiles_y;
        const int n = r;
        const int ly0 = ty * TY, lx0 = tx * TX;
        const int iy0 = 2 * ly0 - p.pad_lo, ix0 = 2 * lx0 - p.pad_lo, iz0 = 2 * t.lz - p.pad_lo;
        t.qorg = p.q + ((((long)n * p.Qd + iz0) * p.Qh + iy0) * p.Qw + ix0) * p.ld_q + c0;
        t.porg = p.p + ((((long)n * p.Ld + t.lz) * p.Lh + ly0) * p.Lw + lx0) * p.ld_p + m0;
        const bool inner = iy0 >= 0 && ix0 >= 0 && iy0 + PY <= p.Qh && ix0 + PX <= p.Qw && c0 + 32 <= p.C;
#pragma unroll
        for (int k = 0; k < KPW; ++k) {
            if (inner) {
                t.qv[k] = qrel[k];
            } else {
                const int iy = iy0 + (qpk[k] & 255), ix = ix0 + ((qpk[k] >> 8) & 255);
                const bool ok = qpk[k] >= 0 && (unsigned)iy < (unsigned)p.Qh && (unsigned)ix < (unsigned)p.Qw && c0 + (qpk[k] >> 16) * 8 < p.C;
                t.qv[k] = ok ? qrel[k] : OOB;
            }
        }
        const bool pin = ly0 + TY <= p.Lh && lx0 + TX <= p.Lw && m0 + 32 * MP <= p.M;
        const bool pok = pin || (ly0 + (ppk & 255) < p.Lh && lx0 + ((ppk >> 8) & 255) < p.Lw && m0 + (ppk >> 16) < p.M);
        t.pv = pok ? prel : OOB;
        return t;
    };
    auto advance = [&](const Tile& c, int tile_next) {         // next tile: same column -> a few adds (no divisions)
        if (c.lz + 1 < p.Ld) {
            Tile t = c;
            t.lz = c.lz + 1;
            t.qorg = c.qorg + 2 * qplane;
            t.porg = c.porg + pplane;
            return t;
        }
        return prep(tile_next);
    };
    auto wrap = [](int v) { return v >= NSLOT ? v - NSLOT : v; };
    auto dma_q = [&](const Tile& t, int rel, int slot, int k) {    // piece run k of Q plane `rel` (0 .. EXT-1) of tile t -> ring slot
        if (k * 8 + wave >= NPI) return;
        const int iz = 2 * t.lz - p.pad_lo + rel;
        const uint4v rs = wg_rsrc_n(t.qorg + rel * qplane, (unsigned)iz < (unsigned)p.Qd ? 0x7fffffffu : 0u);
        wg_dma16(rs, lds0 + slot * SLOTB + (k * 8 + wave) * 1024, t.qv[k]);
    };
    auto dma_p = [&](const Tile& t, int img) { wg_dma16(wg_rsrc_n(t.porg, 0x7fffffffu), lds0 + QB + img * PB + wave * 1024, t.pv); };

    Tile cur = prep(t_begin);
    int base = 0, img = 0;
#pragma unroll
    for (int rel = 0; rel < EXT; ++rel)
#pragma unroll
        for (int k = 0; k < KPW; ++k) dma_q(cur, rel, rel, k);
    dma_p(cur, 0);
    wg_wait_all();
    __syncthreads();

    // per-wave tap constants: tap = wave + 8 ti -> (dz, byte offset inside a plane slab)
    int tapz[TPW], tapc[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int tap = min(wave + 8 * ti, NTAP - 1);
        const int dz = EXT == 3 ? tap / 9 : tap >> 2, dy = EXT == 3 ? (tap / 3) % 3 : (tap >> 1) & 1, dx = EXT == 3 ? tap % 3 : tap & 1;
        tapz[ti] = dz;
        tapc[ti] = q_lane + ((dy & 1) * 2 + (dx & 1)) * SUB + ((dy >> 1) * QX + (dx >> 1)) * 64;
    }

#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const bool more = tile + 1 < t_end;
        Tile nxt = cur;
        if (more) nxt = advance(cur, tile + 1);
        const bool same_col = more && nxt.lz != 0;          // z runs fastest: the next tile continues this column
        const int rel0 = same_col ? EXT - 2 : 0;
        const int nq = more ? (EXT - rel0) * KPW : -1;      // plane DMA instructions of this wave, then one for the P tile
        const int nbase = wrap(base + (same_col ? 2 : EXT));
        int tapaddr[TPW];
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) tapaddr[ti] = wrap(base + tapz[ti]) * SLOTB + tapc[ti];
        const char* const pl = smem + QB + img * PB;
        // MFMA groups g = (16-voxel chunk, tap slot); the operands of group g + 1 are read from LDS before the MFMAs of group g
        // are issued (register double buffer), the next tile's DMA instructions go between the groups (spread: issuing them in
        // one burst at the start of the tile measured 3-8 % slower)
        constexpr int NG = (TV / 16) * TPW;
        half8 a[2][MP];
        half4 bq[2][2];
        auto rdA = [&](int ch, int buf) {
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                const half4 a0 = lds_tr16(pl + mp * TV * 64 + ch * 1024 + p_addr), a1 = lds_tr16(pl + mp * TV * 64 + ch * 1024 + 256 + p_addr);
                a[buf][mp] = half8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            }
        };
        auto rdB = [&](int g, int buf) {
            const int ch = g / TPW, ti = g % TPW;
            if (wave + 8 * ti < NTAP) {
                const char* qa = smem + 2 * ch * QX * 64 + tapaddr[ti];      // chunk rows 2ch, 2ch+1 (+hk in the lane address)
                bq[buf][0] = lds_tr16(qa);
                bq[buf][1] = lds_tr16(qa + 256);
            }
        };
        rdA(0, 0);
        rdB(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ch = g / TPW, ti = g % TPW;
            if (g + 1 < NG) {
                rdB(g + 1, (g + 1) & 1);
                if ((g + 1) % TPW == 0) rdA((g + 1) / TPW, ((g + 1) / TPW) & 1);
            }
            if (wave + 8 * ti < NTAP) {
                const half4 b0 = bq[g & 1][0], b1 = bq[g & 1][1];
                const half8 b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                for (int mp = 0; mp < MP; ++mp) acc[ti][mp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ch & 1][mp], b, acc[ti][mp], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < DPP; ++e) {
                const int d = g * DPP + e;
                if (d < nq) dma_q(nxt, rel0 + d / KPW, wrap(nbase + rel0 + d / KPW), d % KPW);
                else if (d == nq) dma_p(nxt, img ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wg_wait_all();                   // the next tile's planes / P tile are complete ...
        __syncthreads();                 // ... and every wave is done reading this tile
        base = nbase;
        img ^= 1;
        cur = nxt;
    }
    const int c = c0 + (lane & 31);
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int tap = wave + 8 * ti;
        if (tap < NTAP) {
            const long pbase_ = (long)tap * p.Mpad * p.Cpad;
#pragma unroll
            for (int mp = 0; mp < MP; ++mp) {
                if (m0 + mp * 32 >= p.Mpad) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + mp * 32 + 8 * (r >> 2) + 4 * hk + (r & 3);
                    wg_out(p, blk.chunk, 0, pbase_ + (long)m * p.Cpad + c, acc[ti][mp][r]);
                }
            }
        }
    }
}

int launch_wgrad_s1_v5(hipStream_t s, WgradParams& p) {
    constexpr int TZ = 4, TY = 8, TX = 8;
    p.tiles_z = lnn_cdiv(p.Ld, TZ); p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    p.tiles_total = p.N * p.tiles_z * p.tiles_y * p.tiles_x;
    const int panels = (p.Mpad / 32) * (p.Cpad / 32);
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_wgrad_s1_v5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    static int dbg4 = -1;
    static unsigned long long* dbgbuf4 = nullptr;
    if (dbg4 < 0) {
        const char* e = getenv("LNN_WGRAD_DEBUG"); dbg4 = e ? atoi(e) : 0;
        const char* b = getenv("LNN_WGRAD_PHASEBUF"); if ((dbg4 & 4) && b) dbgbuf4 = (unsigned long long*)strtoull(b, nullptr, 0);
    }
    p.debug = dbg4; p.dbgbuf = dbgbuf4;
    // one block per CU: spread tiles x panels over ~num_cu blocks, >= 1 tile per block
    int tpb = lnn_cdiv((long)p.tiles_total * panels, lnn_cu_budget(num_cu));
    if (tpb < 1) tpb = 1;
    if (tpb > p.tiles_total) tpb = p.tiles_total;
    p.tiles_per_block = tpb;
    const unsigned chunks = (unsigned)lnn_cdiv(p.tiles_total, tpb);
    const dim3 grid = wg_grid(p, chunks, (unsigned)panels);
    const long slot_elems = 27L * p.Mpad * p.Cpad;
    if (int e = wg_prepare_parts(p, chunks, 2, slot_elems, "lnn_conv3d_wgrad(s1)")) return e;       // writers: the two tile halves
    hipLaunchKernelGGL(igemm_wgrad_s1_v5_kernel, grid, dim3(512), (size_t)(12 * 7 * 1024 + 2 * 2 * 512 * 16), s, p);
    LNN_CHECK_LAUNCH("lnn_conv3d_wgrad(s1,v5)");
    return wg_reduce_parts(s, p, chunks, slot_elems, "lnn_conv3d_wgrad(s1,v5,reduce)");
}

template <int EXT>
int launch_wgrad_s2_v2(hipStream_t s, WgradParams& p, const char* name) {
    constexpr int TY = 8, TX = 8;
    constexpr int QZ = 1 + (EXT == 3), QY = TY + (EXT == 3), QX = TX + (EXT == 3);
    p.tiles_z = p.Ld; p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    p.tiles_total = p.N * p.tiles_z * p.tiles_y * p.tiles_x;
    const int panels = lnn_cdiv(p.Mpad, 64) * (p.Cpad / 32);
    // one 8-wave block per CU: ~2 rounds of blocks over the chip, >= 1 tile per block
    int tpb = lnn_cdiv((long)p.tiles_total * panels, 512);
    if (tpb < 1) tpb = 1;
    if (tpb > p.tiles_total) tpb = p.tiles_total;
    p.tiles_per_block = tpb;
    const size_t lds = (size_t)8 * QZ * QY * QX * 64 + (size_t)2 * 64 * 64;
    auto kern = igemm_wgrad_s2_v2_kernel<EXT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    const unsigned chunks = (unsigned)lnn_cdiv(p.tiles_total, tpb);
    const dim3 grid = wg_grid(p, chunks, (unsigned)panels);
    const long slot_elems = (long)(EXT * EXT * EXT) * p.Mpad * p.Cpad;
    if (int e = wg_prepare_parts(p, chunks, 1, slot_elems, name)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
    LNN_CHECK_LAUNCH(name);
    return wg_reduce_parts(s, p, chunks, slot_elems, name);
}

// z-streaming stride-2 gathers (igemm_wgrad_s2s_kernel); LNN_WGRAD_S2S=0 keeps the register-prefetch kernel (A/B measurements)
template <int EXT>
int launch_wgrad_s2(hipStream_t s, WgradParams& p, const char* name) {
    static int s2s = -1;
    if (s2s < 0) { const char* e = getenv("LNN_WGRAD_S2S"); s2s = (e && e[0] == '0') ? 0 : 1; }
    constexpr int TY = 8, TX = 8, PY = 2 * (TY - 1) + EXT;
    // the DMA descriptors address a plane / a P tile with 32-bit byte offsets relative to the tile origin
    const bool dma_ok = (long)PY * p.Qw * p.ld_q * 2 < 0x7fffffffL && (long)TY * p.Lw * p.ld_p * 2 < 0x7fffffffL;
    // the level-0 transposed conv (one panel, no plane reuse with EXT = 2) already streams at 4.9 TB/s in the register-prefetch kernel
    const bool one_panel_ext2 = EXT == 2 && lnn_cdiv(p.Mpad, 64) * (p.Cpad / 32) < 2;
    if (!s2s || !dma_ok || one_panel_ext2) return launch_wgrad_s2_v2<EXT>(s, p, name);
    constexpr int QY = TY + (EXT == 3), QX = TX + (EXT == 3), NPI = (4 * QY * QX * 64 + 1023) / 1024;
    p.tiles_z = p.Ld; p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    p.tiles_total = p.N * p.tiles_z * p.tiles_y * p.tiles_x;
    const int panels = lnn_cdiv(p.Mpad, 64) * (p.Cpad / 32);
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    // one 8-wave block per CU: tiles x panels over ~num_cu blocks, whole tiles per block
    int tpb = lnn_cdiv((long)p.tiles_total * panels, lnn_cu_budget(num_cu));
    if (tpb < 1) tpb = 1;
    if (tpb > p.tiles_total) tpb = p.tiles_total;
    p.tiles_per_block = tpb;
    const size_t lds = (size_t)2 * EXT * NPI * 1024 + (size_t)2 * 2 * 64 * 64;
    auto kern = igemm_wgrad_s2s_kernel<EXT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned chunks = (unsigned)lnn_cdiv(p.tiles_total, tpb);
    const dim3 grid = wg_grid(p, chunks, (unsigned)panels);
    const long slot_elems = (long)(EXT * EXT * EXT) * p.Mpad * p.Cpad;
    if (int e = wg_prepare_parts(p, chunks, 1, slot_elems, name)) return e;
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);
    LNN_CHECK_LAUNCH(name);
    return wg_reduce_parts(s, p, chunks, slot_elems, name);
}

int check_act_w(const void* ptr, int ld, int C, const char* what) {
    LNN_REQUIRE(ptr != nullptr, "%s: null pointer", what);
    LNN_REQUIRE(lnn_aligned16(ptr), "%s: pointer not 16-byte aligned", what);
    LNN_REQUIRE(C > 0 && C % 8 == 0, "%s: channel count %d must be a positive multiple of 8", what, C);
    LNN_REQUIRE(ld >= C && ld % 8 == 0, "%s: ld %d must be >= C (%d) and a multiple of 8", what, ld, C);
    return LNN_OK;
}

}  // namespace

namespace {
__global__ void tr16_probe_kernel(float* out) {
    __shared__ __attribute__((aligned(16))) half_t img[256];
    for (int i = threadIdx.x; i < 256; i += 64) img[i] = (half_t)(float)i;
    __syncthreads();
    const half4 r = lds_tr16(reinterpret_cast<const char*>(img) + threadIdx.x * 8);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)r[j];
}
}  // namespace

extern "C" int lnn_debug_tr16_probe(lnn_stream_t s_, float* out256) {
    LNN_REQUIRE(out256 != nullptr, "lnn_debug_tr16_probe: null pointer");
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)s_, out256);
    LNN_CHECK_LAUNCH("lnn_debug_tr16_probe");
    return LNN_OK;
}

extern "C" size_t lnn_wgrad_panel_elems(int ntaps, int M, int KC) {
    return (size_t)ntaps * lnn_round_up(M, 32) * lnn_round_up(KC, 32);
}

namespace {
struct C1Fused {          // lnn_conv3d_wgrad_c1_in_bwd: see WgradParams::gz
    const void* gz; int ld_gz; const float *mean, *rstd, *gamma, *beta; float slope; const double* sums;
};
int conv3d_wgrad_impl(lnn_stream_t s_, const void* x, const void* x2, int c_a, int ld_x, const void* dy, int ld_dy, float* dwp, int N,
                      int Di, int Hi, int Wi, int C, int K, int stride, float* parts = nullptr, long parts_elems = 0,
                      const C1Fused* fused = nullptr) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(stride == 1 || stride == 2, "lnn_conv3d_wgrad: stride %d unsupported", stride);
    LNN_REQUIRE(dwp != nullptr, "lnn_conv3d_wgrad: null panel");
    if (int e = check_act_w(dy, ld_dy, K, "lnn_conv3d_wgrad(dy)")) return e;
    WgradParams p{};
    p.p = (const half_t*)dy; p.q = (const half_t*)x; p.dwp = dwp; p.ld_p = ld_dy; p.ld_q = ld_x;
    p.parts = parts; p.parts_elems = parts_elems;
    if (x2) { p.q2 = (const half_t*)x2; p.csplit = c_a; }
    p.N = N; p.Qd = Di; p.Qh = Hi; p.Qw = Wi;
    p.Ld = (Di - 1) / stride + 1; p.Lh = (Hi - 1) / stride + 1; p.Lw = (Wi - 1) / stride + 1;
    p.M = K; p.C = C; p.Mpad = lnn_round_up(K, 32); p.Cpad = lnn_round_up(C, 32); p.pad_lo = 1;
    if (fused) {
        LNN_REQUIRE(C == 1, "lnn_conv3d_wgrad_c1_in_bwd: first layer (C == 1) only");
        p.gz = (const half_t*)fused->gz; p.ld_gz = fused->ld_gz; p.in_mean = fused->mean; p.in_rstd = fused->rstd;
        p.in_gamma = fused->gamma; p.in_beta = fused->beta; p.in_slope = fused->slope; p.in_sums = fused->sums;
    }
    if (C == 1) {
        LNN_REQUIRE(stride == 1 && x != nullptr, "lnn_conv3d_wgrad: C == 1 path needs stride 1");
        constexpr int TZ = 4, TY = 8;
        p.Cpad = 32;
        p.tiles_z = lnn_cdiv(p.Ld, TZ); p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, 8);
        p.tiles_total = N * p.tiles_z * p.tiles_y * p.tiles_x;
        int tpb = lnn_cdiv(p.tiles_total, 2048);
        if (tpb < 1) tpb = 1;
        p.tiles_per_block = tpb;
        dim3 grid((unsigned)lnn_cdiv(p.tiles_total, tpb), (unsigned)(p.Mpad / 32));
        const long slot_elems = (long)p.Mpad * 32;
        if (int e = wg_prepare_parts(p, grid.x, 4, slot_elems, "lnn_conv3d_wgrad(C=1)")) return e;            // writers: the 4 waves
        if (p.gz) hipLaunchKernelGGL((wgrad_c1_kernel<TZ, TY, true>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((wgrad_c1_kernel<TZ, TY, false>), grid, dim3(256), 0, s, p);
        LNN_CHECK_LAUNCH("lnn_conv3d_wgrad(C=1)");
        return wg_reduce_parts(s, p, grid.x, slot_elems, "lnn_conv3d_wgrad(C=1,reduce)");
    }
    if (int e = check_act_w(x, ld_x, x2 ? c_a : C, "lnn_conv3d_wgrad(x)")) return e;
    if (!x2 && lnn_gen_prefers(LNN_GEN_OP_WGRAD, (long)N * p.Ld * p.Lh * p.Lw)) {
        const int k3[3] = {3, 3, 3}, st3[3] = {stride, stride, stride};
        return lnn_gen_conv3d_wgrad(s, x, ld_x, dy, ld_dy, dwp, N, Di, Hi, Wi, C, K, k3, st3, parts, parts_elems);
    }
    p.taps.ntaps = 27;
    if (stride == 1) {
        constexpr int PY = 10, PX = 10;
        for (int t = 0; t < 27; ++t) {
            p.taps.pos_off[t] = (unsigned short)(((t / 9) * PY + (t / 3) % 3) * PX + t % 3);
            p.taps.slot[t] = (unsigned char)t;
        }
        // the DMA descriptors address a tile with 32-bit offsets relative to its origin: 6 input planes must stay below 2 GB (a
        // z-plane of 350 MB -- no 3-D patch comes near; the register-prefetch kernel that used to take such shapes, 24 spilled
        // registers in its loop, was deleted in round 5)
        LNN_REQUIRE(6L * p.Qh * p.Qw * p.ld_q * 2 < 0x7fffffffL && 4L * p.Lh * p.Lw * p.ld_p * 2 < 0x7fffffffL,
                    "lnn_conv3d_wgrad: a z-plane of %ld bytes is beyond the 2 GB window of the tile descriptors", (long)p.Qh * p.Qw * p.ld_q * 2);
        return launch_wgrad_s1_v5(s, p);
    }
    return launch_wgrad_s2<3>(s, p, "lnn_conv3d_wgrad(s2)");
}
}  // namespace

extern "C" int lnn_conv3d_wgrad(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N,
                                int Di, int Hi, int Wi, int C, int K, int stride) {
    return conv3d_wgrad_impl(s, x, nullptr, 0, ld_x, dy, ld_dy, dwp, N, Di, Hi, Wi, C, K, stride);
}

extern "C" int lnn_conv3d_wgrad_c1_in_bwd(lnn_stream_t s, const void* x, const void* y, const void* dz, int ld_dz, float* dwp, int N,
                                          int D, int H, int W, int K, const float* mean, const float* rstd, const float* gamma,
                                          const float* beta, float slope, const double* ws, float* parts, long parts_elems) {
    LNN_REQUIRE(y && dz && lnn_aligned16(y) && lnn_aligned16(dz) && ld_dz >= K && ld_dz % 8 == 0, "lnn_conv3d_wgrad_c1_in_bwd: bad y / dz / ld_dz");
    LNN_REQUIRE(mean && rstd && gamma && beta && ws, "lnn_conv3d_wgrad_c1_in_bwd: null parameter");
    LNN_REQUIRE(parts == nullptr || lnn_aligned16(parts), "lnn_conv3d_wgrad_c1_in_bwd: scratch misaligned");
    const C1Fused f{dz, ld_dz, mean, rstd, gamma, beta, slope, ws};
    return conv3d_wgrad_impl(s, x, nullptr, 0, 1, y, K, dwp, N, D, H, W, 1, K, 1, parts, parts_elems, &f);
}

extern "C" int lnn_conv3d_wgrad_cat(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* dy, int ld_dy,
                                    float* dwp, int N, int Di, int Hi, int Wi, int C, int K) {
    LNN_REQUIRE(x_b != nullptr && lnn_aligned16(x_b), "lnn_conv3d_wgrad_cat: second tensor null/misaligned");
    LNN_REQUIRE(c_a > 0 && c_a < C && c_a % 32 == 0 && (C - c_a) % 8 == 0, "lnn_conv3d_wgrad_cat: split %d of %d channels must be a multiple of 32", c_a, C);
    LNN_REQUIRE(ld_x >= c_a && ld_x >= C - c_a, "lnn_conv3d_wgrad_cat: ld_x %d smaller than a part (%d / %d)", ld_x, c_a, C - c_a);
    return conv3d_wgrad_impl(s, x_a, x_b, c_a, ld_x, dy, ld_dy, dwp, N, Di, Hi, Wi, C, K, 1);
}

namespace {
int convT3d_k2s2_wgrad_impl(lnn_stream_t s_, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D, int H, int W,
                            int C, int K, float* parts, long parts_elems);
}
extern "C" int lnn_convT3d_k2s2_wgrad(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp,
                                      int N, int D, int H, int W, int C, int K) {
    return convT3d_k2s2_wgrad_impl(s, x, ld_x, dy, ld_dy, dwp, N, D, H, W, C, K, nullptr, 0);
}
// Deterministic variants (VERDICT r1 / SURVEY 7: "deterministic split-K reduce"): the same kernels, but every writer of a
// block stores its partial sums into its own copy of the panel inside `parts` (fp32 scratch, contents irrelevant) and an
// ordered reduction adds the copies to dwp -- no atomics, bit-reproducible run to run.  parts_elems >= blocks.x * writers *
// panel elements (<= 64 M floats for every layer of the BASELINE configs); too small a scratch is an error, not a fallback.
extern "C" int lnn_conv3d_wgrad_det(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int Di, int Hi,
                                    int Wi, int C, int K, int stride, float* parts, long parts_elems) {
    LNN_REQUIRE(parts != nullptr && lnn_aligned16(parts), "lnn_conv3d_wgrad_det: scratch null/misaligned");
    return conv3d_wgrad_impl(s, x, nullptr, 0, ld_x, dy, ld_dy, dwp, N, Di, Hi, Wi, C, K, stride, parts, parts_elems);
}
extern "C" int lnn_conv3d_wgrad_cat_det(lnn_stream_t s, const void* x_a, const void* x_b, int ld_x, int c_a, const void* dy, int ld_dy,
                                        float* dwp, int N, int Di, int Hi, int Wi, int C, int K, float* parts, long parts_elems) {
    LNN_REQUIRE(parts != nullptr && lnn_aligned16(parts), "lnn_conv3d_wgrad_cat_det: scratch null/misaligned");
    LNN_REQUIRE(x_b != nullptr && lnn_aligned16(x_b), "lnn_conv3d_wgrad_cat_det: second tensor null/misaligned");
    LNN_REQUIRE(c_a > 0 && c_a < C && c_a % 32 == 0 && (C - c_a) % 8 == 0, "lnn_conv3d_wgrad_cat_det: split %d of %d channels must be a multiple of 32", c_a, C);
    LNN_REQUIRE(ld_x >= c_a && ld_x >= C - c_a, "lnn_conv3d_wgrad_cat_det: ld_x %d smaller than a part (%d / %d)", ld_x, c_a, C - c_a);
    return conv3d_wgrad_impl(s, x_a, x_b, c_a, ld_x, dy, ld_dy, dwp, N, Di, Hi, Wi, C, K, 1, parts, parts_elems);
}
extern "C" int lnn_convT3d_k2s2_wgrad_det(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D,
                                          int H, int W, int C, int K, float* parts, long parts_elems) {
    LNN_REQUIRE(parts != nullptr && lnn_aligned16(parts), "lnn_convT3d_k2s2_wgrad_det: scratch null/misaligned");
    return convT3d_k2s2_wgrad_impl(s, x, ld_x, dy, ld_dy, dwp, N, D, H, W, C, K, parts, parts_elems);
}
namespace {
int convT3d_k2s2_wgrad_impl(lnn_stream_t s_, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D, int H, int W,
                            int C, int K, float* parts, long parts_elems) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(dwp != nullptr, "lnn_convT3d_k2s2_wgrad: null panel");
    if (int e = check_act_w(x, ld_x, C, "lnn_convT3d_k2s2_wgrad(x)")) return e;
    if (int e = check_act_w(dy, ld_dy, K, "lnn_convT3d_k2s2_wgrad(dy)")) return e;
    if (lnn_gen_prefers(LNN_GEN_OP_WGRAD, (long)N * D * H * W)) {
        const int st3[3] = {2, 2, 2};
        return lnn_gen_convT3d_wgrad(s, x, ld_x, dy, ld_dy, dwp, N, D, H, W, C, K, st3, parts, parts_elems);
    }
    WgradParams p{};
    // dW[c,k,d] = sum_l x[l,c] dy[2l+d,k]:  P = x (rows c), Q = dy gathered with stride 2 (cols k)
    p.p = (const half_t*)x; p.q = (const half_t*)dy; p.dwp = dwp; p.ld_p = ld_x; p.ld_q = ld_dy;
    p.parts = parts; p.parts_elems = parts_elems;
    p.N = N; p.Ld = D; p.Lh = H; p.Lw = W; p.Qd = 2 * D; p.Qh = 2 * H; p.Qw = 2 * W;
    p.M = C; p.C = K; p.Mpad = lnn_round_up(C, 32); p.Cpad = lnn_round_up(K, 32); p.pad_lo = 0;
    return launch_wgrad_s2<2>(s, p, "lnn_convT3d_k2s2_wgrad");
}
}  // namespace
