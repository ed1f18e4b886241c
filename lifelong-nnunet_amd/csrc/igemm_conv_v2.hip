// Stride-1 3x3x3 implicit-GEMM convolution: persistent blocks + software-pipelined staging (v3 of the kernel).
//
// This is the kernel behind ~90 % of the forward + dgrad FLOPs of the U-Net (every stride-1 conv and its
// data gradient).  Same GEMM mapping as igemm_conv.hip (MFMA rows = output channels, columns = 32 voxels,
// v_mfma_f32_32x32x16_f16); what differs is everything around the MFMAs, shaped by PMC counters of the
// previous version (12.6 VALU instructions per MFMA, 51 % of LDS cycles lost to bank conflicts):
//
//   * TWO blocks (4 waves each) per CU, each walking a contiguous range of work units (4x8x8-voxel tile x 32
//     output channels); the pipeline runs ACROSS unit boundaries, and the co-resident block's MFMAs cover this
//     block's LDS / barrier latencies;
//   * a step = (input-channel chunk of 32, dz plane of 3x3 taps) = 36 MFMAs per wave.  While the MFMAs of step s
//     run, the global loads of step s+1 (9 weight panels) and one third of the NEXT halo tile are in flight into
//     registers; weights land in LDS between two barriers after the MFMAs, the halo tile when its pair ends;
//   * LDS halo tile: 64-byte rows (32 channels), x pitch padded 10 -> 12 positions, 16-byte slot XOR key
//     ((px >> 2) & 1) | ((py & 1) << 1).  Together with the lane -> voxel map below (each ds_read_b128 16-lane
//     group covers 2 rows x 8 consecutive x) every fragment read is bank-conflict free for all 27 taps;
//   * addressing is strength-reduced: fragment reads are (per-lane base register [dx][row parity]) + compile-time
//     immediate, prefetch loads are (uniform tile base) + (per-thread precomputed 32-bit offset) with the
//     bounds checks only on tiles that touch the volume border.
#include "igemm_common.h"

namespace {

constexpr int TZ = 4, TY = 8, TX = 8, PZ = 6, PY = 10, PX = 10, PXP = 12;
constexpr int P = PZ * PY * PX;             // 600 real halo positions
constexpr int XBYTES = PZ * PY * PXP * 64;  // 46080 (padded pitch)
constexpr int XCHUNKS = P * 4;              // 16-byte chunks per halo tile
constexpr int XTHIRD = XCHUNKS / 3;         // 800, staged per dz step
constexpr int XN = (XTHIRD + 255) / 256;    // 4 loads per thread per step
constexpr int MB = 32, VT = 2;
constexpr int WBYTES = 9 * MB * 64, WCHUNKS = 9 * MB * 4, WN = (WCHUNKS + 255) / 256;  // 18432 B, 5 loads

__device__ __forceinline__ int wswz(int row, int c16) { return row * 64 + ((c16 ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ int xkey(int py, int px) { return ((px >> 2) & 1) | ((py & 1) << 1); }
__device__ __forceinline__ int xaddr(int pz, int py, int px, int c16) {
    return ((pz * PY + py) * PXP + px) * 64 + ((c16 ^ xkey(py, px)) << 4);
}

// lane (0..31) -> (row 0..3, x 0..7) inside a 32-voxel MFMA tile.  ds_read_b128 is serviced in the 16-lane groups
// {0-3,12-15,20-27} and {4-11,16-19,28-31}: each group gets two full 8-voxel rows.
__device__ __forceinline__ void lane_voxel(int v, int& r, int& x) {
    if (v < 4) { r = 0; x = v; }
    else if (v < 12) { r = 2; x = v - 4; }
    else if (v < 16) { r = 0; x = v - 8; }
    else if (v < 20) { r = 3; x = v - 16; }
    else if (v < 28) { r = 1; x = v - 20; }
    else { r = 3; x = v - 24; }
}

struct Pair {   // one (work unit, channel chunk)
    int n, lz0, ly0, lx0, m0, c0;
    bool valid, first_chunk, last_chunk, interior;
};

__global__ __launch_bounds__(256, 2) void igemm_conv_s1_v3_kernel(const ConvParams p, int units_total, int mblocks,
                                                                  int units_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xb = smem;
    char* const wb = smem + XBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = lane & 31, hk = lane >> 5;
    const int u_begin = blockIdx.x * units_per_block;
    const int u_end = min(u_begin + units_per_block, units_total);
    if (u_begin >= u_end) return;
    const int nchunks = (p.C + 31) / 32;
    const bool flip = p.taps.slot[0] != 0;   // dgrad: tap offset d' uses weight slot 26 - d'
    const int nq = (u_end - u_begin) * nchunks;

    auto decode = [&](int q) {
        Pair r;
        r.valid = q < nq;
        const int u = u_begin + q / nchunks, ch = q % nchunks;
        int t = u / mblocks;
        r.m0 = (u % mblocks) * MB;
        r.c0 = ch * 32;
        r.first_chunk = ch == 0;
        r.last_chunk = ch == nchunks - 1;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int tz = t % p.tiles_z; t /= p.tiles_z;
        r.n = t; r.lz0 = tz * TZ; r.ly0 = ty * TY; r.lx0 = tx * TX;
        // whole halo inside the volume and a full 32-channel chunk -> no per-element checks needed
        r.interior = r.lz0 >= 1 && r.ly0 >= 1 && r.lx0 >= 1 && r.lz0 + TZ + 1 <= p.Di && r.ly0 + TY + 1 <= p.Hi &&
                     r.lx0 + TX + 1 <= p.Wi && r.c0 + 32 <= p.C;
        return r;
    };

    // ---- per-thread staging constants (tile independent) -------------------------------------------------
    int xrel[3][XN], xlds[3][XN], xcrd[3][XN];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int li = min(i * 256 + tid, XTHIRD - 1);
            const int idx = j * XTHIRD + li;
            const int pos = idx >> 2, c4 = idx & 3;
            const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
            xrel[j][i] = ((pz * p.Hi + py) * p.Wi + px) * p.ld_x + c4 * 8;
            xlds[j][i] = xaddr(pz, py, px, c4);
            xcrd[j][i] = pz | (py << 8) | (px << 16) | (c4 << 24);
        }
    int wrel[WN], wlds[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int idx = min(i * 256 + tid, WCHUNKS - 1);
        const int c4 = idx & 3, r = (idx >> 2) % MB, tl = idx / (4 * MB);
        wrel[i] = ((flip ? -tl : tl) * p.Mpad + r) * p.KCpad + c4 * 8;
        wlds[i] = wswz(idx >> 2, c4) | (c4 << 28);     // c4 kept in the top bits for the chunk-tail mask
    }

    // Prefetch registers.  The global loads are UNCONDITIONAL (out-of-range lanes read element 0 of the tensor and
    // are zeroed when the value is written to LDS): a predicated load makes hipcc wrap each one in an exec-mask
    // branch with s_waitcnt vmcnt(0) in front, which serialises the whole prefetch.
    half8 xr[3][XN], wr[WN];
    unsigned xok[3] = {0xFu, 0xFu, 0xFu}, wok = 0;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_x = [&](const Pair& t, const int (&rel)[XN], const int (&crd)[XN], half8 (&dst)[XN], unsigned& okmask) {
        const long base = ((((long)t.n * p.Di + (t.lz0 - 1)) * p.Hi + (t.ly0 - 1)) * p.Wi + (t.lx0 - 1)) * p.ld_x + t.c0;
        if (t.interior) {
            const half_t* bp = p.x + base;
#pragma unroll
            for (int i = 0; i < XN; ++i) dst[i] = *reinterpret_cast<const half8*>(bp + rel[i]);
            okmask = 0xFu;
        } else {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < XN; ++i) {
                const int pz = crd[i] & 255, py = (crd[i] >> 8) & 255, px = (crd[i] >> 16) & 255, c4 = crd[i] >> 24;
                const int iz = t.lz0 - 1 + pz, iy = t.ly0 - 1 + py, ix = t.lx0 - 1 + px;
                const bool ok = (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi &&
                                t.c0 + c4 * 8 < p.C;
                const long off = ok ? base + rel[i] : 0;
                dst[i] = *reinterpret_cast<const half8*>(p.x + off);
                m |= (ok ? 1u : 0u) << i;
            }
            okmask = m;
        }
    };
    auto store_x = [&](const int (&lds)[XN], const half8 (&src)[XN], unsigned okmask) {
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            if (i * 256 + tid < XTHIRD)
                *reinterpret_cast<half8*>(xb + lds[i]) = ((okmask >> i) & 1u) ? src[i] : zero8;
        }
    };
    auto load_w = [&](const Pair& t, int j) {
        const int s0 = flip ? 26 - j * 9 : j * 9;
        const half_t* bp = p.wp + ((long)s0 * p.Mpad + t.m0) * p.KCpad + t.c0;
        const int cleft = p.KCpad - t.c0;      // channels left in the panel row (>= 16)
        unsigned m = 0;
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const bool ok = (int)((unsigned)wlds[i] >> 28) * 8 < cleft;
            wr[i] = *reinterpret_cast<const half8*>(bp + (ok ? wrel[i] : 0));
            m |= (ok ? 1u : 0u) << i;
        }
        wok = m;
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            if (i * 256 + tid < WCHUNKS)
                *reinterpret_cast<half8*>(wb + (wlds[i] & 0x0FFFFFFF)) = ((wok >> i) & 1u) ? wr[i] : zero8;
        }
    };

    // ---- per-lane fragment addressing -------------------------------------------------------------------------
    int vr, vx;
    lane_voxel(v, vr, vx);
    // lterm[vt][dx][par]: byte address of (z, y, x+dx) with the slot key of row parity (y+par)&1, chunk hk
    int lterm[VT][3][2];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int tile = wave * VT + vt;
        const int z = tile / (TY / 4), y = (tile % (TY / 4)) * 4 + vr;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int par = 0; par < 2; ++par)
                lterm[vt][dx][par] = ((z * PY + y) * PXP + vx + dx) * 64 + ((hk ^ xkey(y + par, vx + dx)) << 4);
    }
    const int a_lane = v * 64 + ((hk ^ ((v >> 2) & 3)) << 4);

    floatx16 acc[VT];

    // ---- prologue: first halo tile + first weight group, synchronously --------------------------------------
    Pair cur = decode(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        load_x(cur, xrel[j], xcrd[j], xr[j], xok[j]);
        store_x(xlds[j], xr[j], xok[j]);
    }
    load_w(cur, 0);
    store_w();
    __syncthreads();

#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        const Pair nxt = decode(q + 1);
        if (cur.first_chunk) {
#pragma unroll
            for (int b = 0; b < VT; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool more = !(q == nq - 1 && j == 2);
            // ---- issue next step's global loads (in flight during the MFMAs below) ----
            if (nxt.valid) load_x(nxt, xrel[j], xcrd[j], xr[j], xok[j]);
            if (more) {
                if (j < 2) load_w(cur, j + 1); else load_w(nxt, 0);
            }
            // ---- MFMAs of this step: 9 taps x 2 k-slices, all addresses = register + immediate ----
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tl = dy * 3 + dx;
                    const int ximm = ((j * PY + dy) * PXP) * 64;
#pragma unroll
                    for (int k16 = 0; k16 < 2; ++k16) {
                        const half8 a = *reinterpret_cast<const half8*>(wb + tl * MB * 64 + (a_lane ^ (k16 << 5)));
                        half8 b[VT];
#pragma unroll
                        for (int vt = 0; vt < VT; ++vt)
                            b[vt] = *reinterpret_cast<const half8*>(xb + ximm + (lterm[vt][dx][dy & 1] ^ (k16 << 5)));
#pragma unroll
                        for (int vt = 0; vt < VT; ++vt)
                            acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[vt], acc[vt], 0, 0, 0);
                    }
                }
            // ---- land the prefetched data: barrier (readers done) -> LDS writes -> barrier ----
            __syncthreads();
            if (more) store_w();
            if (j == 2 && nxt.valid) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) store_x(xlds[jj], xr[jj], xok[jj]);
            }
            __syncthreads();
        }
        if (cur.last_chunk) {
            // ---- epilogue: lane holds voxel (vr, vx) x channels {8*q + 4*hk + 0..3} per accumulator quad ----
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int tile = wave * VT + vt;
                const int lz = cur.lz0 + tile / (TY / 4), ly = cur.ly0 + (tile % (TY / 4)) * 4 + vr, lx = cur.lx0 + vx;
                if (lz >= p.Ld || ly >= p.Lh || lx >= p.Lw) continue;
                half_t* yrow = p.y + ((((long)cur.n * p.Do + lz) * p.Ho + ly) * p.Wo + lx) * p.ld_y;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int m = cur.m0 + qq * 8 + hk * 4;
                    if (m >= p.M) continue;
                    float r0 = acc[vt][qq * 4 + 0], r1 = acc[vt][qq * 4 + 1], r2 = acc[vt][qq * 4 + 2], r3 = acc[vt][qq * 4 + 3];
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
        cur = nxt;
    }
}

}  // namespace

int lnn_launch_conv_s1_v2(hipStream_t s, ConvParams& p, const char* name) {
    p.tiles_z = lnn_cdiv(p.Ld, TZ); p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    const int mblocks = lnn_cdiv(p.M, MB);
    const long units = (long)p.N * p.tiles_z * p.tiles_y * p.tiles_x * mblocks;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int upb = lnn_cdiv(units, 2 * num_cu);     // two resident blocks per CU
    if (upb < 1) upb = 1;
    const int grid = lnn_cdiv(units, upb);
    const size_t lds = XBYTES + WBYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_conv_s1_v3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm_conv_s1_v3_kernel, dim3(grid), dim3(256), lds, s, p, (int)units, mblocks, upb);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}
