// Resolution-halving implicit GEMMs: forward of the stride-2 3x3x3 conv (EXT = 3, 27 taps, pad 1) and data gradient
// of the 2x2x2 stride-2 transposed conv (EXT = 2, 8 taps, pad 0):
//     out[n, l, m] = bias[m] + sum_{d in EXT^3} sum_c  W[d][m][c] . in[n, 2l + d - pad, c]
// The generic kernel (igemm_conv.hip) ran these with a 2x8x8 output tile: 1.6x halo over-read, weights re-staged
// for every 128 output voxels and an LDS tile read with 64-byte lane stride (2..4-way bank conflicts).  They are
// bound by the input read (4x the output), so this version is organised around the loads:
//   * persistent 512-thread blocks (8 waves) walk (4x8x8 output tile, up to 64 output channels) units; a step is
//     one 16-channel chunk = 27 (8) taps x MT MFMAs per wave on a 9x17x17 (8x16x16) input tile staged ONCE for all
//     output channels of the block;
//   * the next step's global loads are issued before the MFMAs and parked in registers (the 83 KB tile is single
//     buffered: LDS cannot hold two), written after a barrier;
//   * the input tile is de-interleaved by x parity in LDS: tap dx reads plane (dx & 1) at consecutive positions, so
//     the fragment reads have the same conflict-free 32-byte lane stride / parity-keyed halves as the stride-1 kernel;
//   * blocked weight panels: each (32-row block, chunk) is one contiguous 27 KB (8 KB) run.
#include "igemm_common.h"

namespace {

constexpr int TZ = 4, TY = 8, TX = 8;
constexpr int CK = 16, ROWB = 32, NT = 512;

template <int EXT, int MT>
struct DnCfg {
    static constexpr int PZ = 2 * (TZ - 1) + EXT, PY = 2 * (TY - 1) + EXT, PX = 2 * (TX - 1) + EXT;
    static constexpr int PXH = (PX + 1) / 2;                 // positions per x-parity plane
    static constexpr int P = PZ * PY * PX;
    static constexpr int XBYTES = PZ * PY * 2 * PXH * ROWB;  // LDS tile (planes padded to PXH)
    static constexpr int XCHUNKS = P * 2;
    static constexpr int XN = (XCHUNKS + NT - 1) / NT;
    static constexpr int NTAP = EXT * EXT * EXT;
    static constexpr int MB = 32 * MT;
    static constexpr int WBYTES = NTAP * MB * ROWB;
    static constexpr int WCHUNKS = NTAP * MB * 2;
    static constexpr int WN = (WCHUNKS + NT - 1) / NT;
};

__device__ __forceinline__ void dn_lane_voxel(int v, int& r, int& x) {   // see igemm_conv_tile.hip
    if (v < 4) { r = 0; x = v; }
    else if (v < 12) { r = 2; x = v - 4; }
    else if (v < 16) { r = 0; x = v - 8; }
    else if (v < 20) { r = 3; x = v - 16; }
    else if (v < 28) { r = 1; x = v - 20; }
    else { r = 3; x = v - 24; }
}

struct DnStep {
    int n, lz0, ly0, lx0, m0, c0;
    bool valid, first_chunk, last_chunk, interior;
};

template <int EXT, int MT>
__global__ __launch_bounds__(NT, 2) void igemm_down2_kernel(const ConvParams p, int units_total, int tiles_total,
                                                            int units_per_block) {
    using Cfg = DnCfg<EXT, MT>;
    constexpr int PY = Cfg::PY, PX = Cfg::PX, PXH = Cfg::PXH, XCHUNKS = Cfg::XCHUNKS, XN = Cfg::XN;
    constexpr int NTAP = Cfg::NTAP, MB = Cfg::MB, WCHUNKS = Cfg::WCHUNKS, WN = Cfg::WN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xl = smem;
    char* const wl = smem + Cfg::XBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = lane & 31, hk = lane >> 5;
    const int u_begin = blockIdx.x * units_per_block;
    const int u_end = min(u_begin + units_per_block, units_total);
    if (u_begin >= u_end) return;
    const int nchunks = (p.C + CK - 1) / CK;
    const int nq = (u_end - u_begin) * nchunks;
    const int pad = p.pad_lo;

    auto decode = [&](int q) {
        DnStep r;
        r.valid = q < nq;
        const int u = u_begin + q / nchunks;
        const int ch = q % nchunks;
        int t = u % tiles_total;
        r.m0 = (u / tiles_total) * MB;
        r.c0 = ch * CK;
        r.first_chunk = ch == 0;
        r.last_chunk = ch == nchunks - 1;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int tz = t % p.tiles_z; t /= p.tiles_z;
        r.n = t; r.lz0 = tz * TZ; r.ly0 = ty * TY; r.lx0 = tx * TX;
        const int iz0 = 2 * r.lz0 - pad, iy0 = 2 * r.ly0 - pad, ix0 = 2 * r.lx0 - pad;
        r.interior = iz0 >= 0 && iy0 >= 0 && ix0 >= 0 && iz0 + Cfg::PZ <= p.Di && iy0 + PY <= p.Hi && ix0 + PX <= p.Wi &&
                     r.c0 + CK <= p.C;
        return r;
    };

    // ---- per-thread staging constants ---------------------------------------------------------------------
    int xrel[XN], xlds[XN];
#pragma unroll
    for (int i = 0; i < XN; ++i) {
        const int idx = min(i * NT + tid, XCHUNKS - 1);
        const int pos = idx >> 1, c2 = idx & 1;
        const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
        xrel[i] = ((pz * p.Hi + py) * p.Wi + px) * p.ld_x + c2 * 8;
        xlds[i] = (((pz * PY + py) * 2 + (px & 1)) * PXH + (px >> 1)) * ROWB + ((c2 ^ ((py >> 1) & 1)) << 4);
    }
    int wrel[WN], wlds[WN];
    const int nck16 = p.KCpad >> 4;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int idx = min(i * NT + tid, WCHUNKS - 1);
        const int c2 = idx & 1, row = idx >> 1, r = row % MB, tl = row / MB;
        // row block (r >> 5) of the unit's MT blocks: each (row block, chunk) is a contiguous NTAP x 1 KB run
        wrel[i] = ((r >> 5) * nck16 * NTAP + tl) * 512 + (r & 31) * 16 + c2 * 8;
        wlds[i] = row * ROWB + ((c2 ^ ((r >> 3) & 1)) << 4);
    }

    half8 xr[XN], wr[WN];
    unsigned xok = 0;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_x = [&](const DnStep& t) {
        const int iz0 = 2 * t.lz0 - pad, iy0 = 2 * t.ly0 - pad, ix0 = 2 * t.lx0 - pad;
        const long base = ((((long)t.n * p.Di + iz0) * p.Hi + iy0) * p.Wi + ix0) * p.ld_x + t.c0;
        if (t.interior) {
            const half_t* bp = p.x + base;
#pragma unroll
            for (int i = 0; i < XN; ++i) xr[i] = *reinterpret_cast<const half8*>(bp + xrel[i]);
            xok = 0xFFFFFFFFu;
        } else {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < XN; ++i) {
                const int idx = min(i * NT + tid, XCHUNKS - 1);
                const int pos = idx >> 1, c2 = idx & 1;
                const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
                const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
                const bool ok = (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi &&
                                t.c0 + c2 * 8 < p.C;
                xr[i] = *reinterpret_cast<const half8*>(p.x + (ok ? base + xrel[i] : 0));
                m |= (ok ? 1u : 0u) << i;
            }
            xok = m;
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < XN; ++i)
            if (i * NT + tid < XCHUNKS) *reinterpret_cast<half8*>(xl + xlds[i]) = ((xok >> i) & 1u) ? xr[i] : zero8;
    };
    auto load_w = [&](int m0, int c0) {
        const half_t* bp = p.wp + ((long)(m0 >> 5) * nck16 + (c0 >> 4)) * NTAP * 512;
        // row blocks past Mpad (MT = 2 with an odd number of 32-row blocks) are masked: read block 0 instead
        const bool full = m0 + MB <= p.Mpad;
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int idx = min(i * NT + tid, WCHUNKS - 1);
            const int r = (idx >> 1) % MB;
            const bool ok = full || r < 32;
            wr[i] = *reinterpret_cast<const half8*>(bp + (ok ? wrel[i] : wrel[i] - (r >> 5) * nck16 * NTAP * 512));
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WN; ++i)
            if (i * NT + tid < WCHUNKS) *reinterpret_cast<half8*>(wl + wlds[i]) = wr[i];
    };

    // ---- per-lane fragment addressing: wave -> (z, y half); lane -> (row, x) --------------------------------
    int vr, vx;
    dn_lane_voxel(v, vr, vx);
    const int wz = wave >> 1, wy = (wave & 1) * 4 + vr;
    int lterm[2];   // input position (2 wz, 2 wy, 2 vx) in plane 0; half keyed with the parity of output row wy + par
#pragma unroll
    for (int par = 0; par < 2; ++par)
        lterm[par] = (((2 * wz * PY + 2 * wy) * 2) * PXH + vx) * ROWB + ((hk ^ ((wy + par) & 1)) << 4);
    const int a_lane = v * ROWB + ((hk ^ ((v >> 3) & 1)) << 4);

    floatx16 acc[MT];

    DnStep cur = decode(0);
    load_x(cur);
    load_w(cur.m0, cur.c0);
    store_x();
    store_w();
    __syncthreads();

#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        const DnStep nxt = decode(q + 1);
        if (cur.first_chunk) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
        }
        if (nxt.valid) {
            load_x(nxt);
            load_w(nxt.m0, nxt.c0);
        }
#pragma unroll
        for (int tl = 0; tl < NTAP; ++tl) {
            const int dz = tl / (EXT * EXT), dy = (tl / EXT) % EXT, dx = tl % EXT;
            // input row 2 wy + dy has key ((2 wy + dy) >> 1) & 1 = (wy + (dy >> 1)) & 1
            const int ximm = (((dz * PY + dy) * 2 + (dx & 1)) * PXH + (dx >> 1)) * ROWB;
            const half8 b = *reinterpret_cast<const half8*>(xl + ximm + lterm[dy >> 1]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const half8 a = *reinterpret_cast<const half8*>(wl + (tl * MB + mt * 32) * ROWB + a_lane);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[mt], 0, 0, 0);
            }
        }
        __syncthreads();          // every wave is done reading this step's tile / weights
        if (nxt.valid) {
            store_x();
            store_w();
        }
        if (cur.last_chunk) {
            const int lz = cur.lz0 + wz, ly = cur.ly0 + wy, lx = cur.lx0 + vx;
            if (lz < p.Do && ly < p.Ho && lx < p.Wo) {
                half_t* yrow = p.y + ((((long)cur.n * p.Do + lz) * p.Ho + ly) * p.Wo + lx) * p.ld_y;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int m = cur.m0 + mt * 32 + qq * 8 + hk * 4;
                        if (m >= p.M) continue;
                        float r0 = acc[mt][qq * 4 + 0], r1 = acc[mt][qq * 4 + 1], r2 = acc[mt][qq * 4 + 2], r3 = acc[mt][qq * 4 + 3];
                        if (p.bias) {
                            const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + m);
                            r0 += bv[0]; r1 += bv[1]; r2 += bv[2]; r3 += bv[3];
                        }
                        half4* dst = reinterpret_cast<half4*>(yrow + m);
                        if (p.accumulate) {
                            const half4 old = *dst;
                            r0 += (float)old[0]; r1 += (float)old[1]; r2 += (float)old[2]; r3 += (float)old[3];
                        }
                        half4 o = {(half_t)r0, (half_t)r1, (half_t)r2, (half_t)r3};
                        *dst = o;
                    }
            }
        }
        __syncthreads();          // the next step's tile / weights are in place
        cur = nxt;
    }
}

template <int EXT, int MT>
int launch_down2(hipStream_t s, ConvParams& p, const char* name) {
    using Cfg = DnCfg<EXT, MT>;
    p.tiles_z = lnn_cdiv(p.Do, TZ); p.tiles_y = lnn_cdiv(p.Ho, TY); p.tiles_x = lnn_cdiv(p.Wo, TX);
    const int mblocks = lnn_cdiv(p.M, Cfg::MB);
    const int tiles = p.N * p.tiles_z * p.tiles_y * p.tiles_x;
    const long units = (long)tiles * mblocks;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int upb = lnn_cdiv(units, num_cu);
    if (upb < 1) upb = 1;
    const int grid = lnn_cdiv(units, upb);
    const size_t lds = Cfg::XBYTES + Cfg::WBYTES;
    auto kern = igemm_down2_kernel<EXT, MT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, p, (int)units, tiles, upb);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}

}  // namespace

// p.x: input (C channels, extents Di/Hi/Wi), p.y: output (M channels, extents Do/Ho/Wo), p.pad_lo = 1 | 0
int lnn_launch_down2_conv(hipStream_t s, ConvParams& p, const char* name) {
    return p.M > 32 ? launch_down2<3, 2>(s, p, name) : launch_down2<3, 1>(s, p, name);
}
int lnn_launch_down2_convT_dgrad(hipStream_t s, ConvParams& p, const char* name) {
    return p.M > 32 ? launch_down2<2, 2>(s, p, name) : launch_down2<2, 1>(s, p, name);
}
