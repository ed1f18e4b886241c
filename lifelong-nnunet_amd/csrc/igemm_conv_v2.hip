// Stride-1 3x3x3 implicit-GEMM convolution, v2: persistent blocks + software-pipelined staging.
//
// This is the kernel behind ~90 % of the forward + dgrad FLOPs of the U-Net (every stride-1 conv and its
// data gradient).  Same GEMM mapping as igemm_conv.hip (MFMA rows = output channels, columns = 32 voxels,
// v_mfma_f32_32x32x16_f16) but the staging is restructured around the LDS capacity of a gfx950 CU:
//
//   * TWO blocks (4 waves each) per CU, each walking a contiguous range of work units (spatial tile x 32*MT
//     output channels); the pipeline runs ACROSS unit boundaries, so there is no per-tile ramp-up, and the
//     co-resident block's MFMAs cover this block's LDS / VALU / barrier latencies;
//   * a step = (input-channel chunk of 32, dz plane of 3x3 taps).  While the MFMAs of step s run, the global
//     loads of step s+1 (9 weight panels) and one third of the NEXT halo tile are in flight into registers;
//     weights are written to LDS between two barriers after the MFMAs, the halo tile when its pair ends;
//   * LDS rows are 64 B (32 channels) with an XOR swizzle of the 16-B slot ((row >> 2) & 3) instead of
//     padding: conflict-free ds_read_b128 fragment reads, 37.5 KB halo + 18 KB*MT weights <= 73.5 KB / block.
#include "igemm_common.h"

namespace {

constexpr int TZ = 4, TY = 8, TX = 8, PZ = 6, PY = 10, PX = 10, P = PZ * PY * PX;
constexpr int XBYTES = P * 64;            // 38400
constexpr int XCHUNKS = P * 4;            // 16-byte chunks per halo tile
constexpr int XTHIRD = XCHUNKS / 3;       // 800, staged per dz step
constexpr int XN = (XTHIRD + 255) / 256;  // 4 loads per thread per step

__device__ __forceinline__ int swz(int row, int c16) { return row * 64 + ((c16 ^ ((row >> 2) & 3)) << 4); }

struct Pair {   // one (work unit, channel chunk)
    int n, lz0, ly0, lx0, m0, c0;
    bool valid, first_chunk, last_chunk;
};

template <int MT>
__global__ __launch_bounds__(256, 2) void igemm_conv_s1_v2_kernel(const ConvParams p, int units_total, int mblocks,
                                                                  int units_per_block) {
    constexpr int VT = 2, MB = 32 * MT;
    constexpr int WBYTES = 9 * MB * 64, WCHUNKS = 9 * MB * 4, WN = (WCHUNKS + 255) / 256;
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
        return r;
    };

    // Prefetch registers.  The global loads are UNCONDITIONAL (out-of-range lanes read element 0 of the tensor
    // and are zeroed when the value is written to LDS): a predicated load makes hipcc wrap each one in an
    // exec-mask branch with s_waitcnt vmcnt(0) in front, which serialises the whole prefetch.
    half8 xr[3][XN], wr[WN];
    unsigned xok[3] = {0, 0, 0}, wok = 0;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_x = [&](const Pair& t, int j, half8 (&dst)[XN], unsigned& okmask) {
        const long base_n = (long)t.n * p.Di * p.Hi * p.Wi;
        unsigned xok = 0;
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int li = min(i * 256 + tid, XTHIRD - 1);
            const int idx = j * XTHIRD + li;
            const int pos = idx >> 2, c4 = idx & 3;
            const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
            const int iz = t.lz0 - 1 + pz, iy = t.ly0 - 1 + py, ix = t.lx0 - 1 + px;
            const bool ok = (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi &&
                            t.c0 + c4 * 8 < p.C;
            const long off = ok ? (base_n + ((long)iz * p.Hi + iy) * p.Wi + ix) * p.ld_x + t.c0 + c4 * 8 : 0;
            dst[i] = *reinterpret_cast<const half8*>(p.x + off);
            xok |= (ok ? 1u : 0u) << i;
        }
        okmask = xok;
    };
    auto store_x = [&](char* buf, int j, const half8 (&src)[XN], unsigned xok) {
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int li = i * 256 + tid;
            if (li < XTHIRD) {
                const int idx = j * XTHIRD + li;
                *reinterpret_cast<half8*>(buf + swz(idx >> 2, idx & 3)) = ((xok >> i) & 1u) ? src[i] : zero8;
            }
        }
    };
    auto load_w = [&](const Pair& t, int j) {
        wok = 0;
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int idx = min(i * 256 + tid, WCHUNKS - 1);
            const int c4 = idx & 3, r = (idx >> 2) % MB, tl = idx / (4 * MB);
            const int slot = flip ? 26 - (j * 9 + tl) : j * 9 + tl;   // arithmetic, not a per-lane table gather
            const bool ok = t.m0 + r < p.Mpad && t.c0 + c4 * 8 < p.KCpad;
            const long off = ok ? ((long)slot * p.Mpad + t.m0 + r) * p.KCpad + t.c0 + c4 * 8 : 0;
            wr[i] = *reinterpret_cast<const half8*>(p.wp + off);
            wok |= (ok ? 1u : 0u) << i;
        }
    };
    auto store_w = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int idx = i * 256 + tid;
            if (idx < WCHUNKS) *reinterpret_cast<half8*>(buf + swz(idx >> 2, idx & 3)) = ((wok >> i) & 1u) ? wr[i] : zero8;
        }
    };

    // per-lane constants
    int basepos[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int tile = wave * VT + vt;
        const int z = tile / (TY / 4), y = (tile % (TY / 4)) * 4 + (v >> 3), x = v & 7;
        basepos[vt] = (z * PY + y) * PX + x;
    }
    const int a_swz = (v >> 2) & 3;
    const int a_lane = v * 64;

    floatx16 acc[MT][VT];

    // ---- prologue: first halo tile + first weight group, synchronously --------------------------------
    Pair cur = decode(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        load_x(cur, j, xr[j], xok[j]);
        store_x(xb, j, xr[j], xok[j]);
    }
    load_w(cur, 0);
    store_w(wb);
    __syncthreads();

#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        const Pair nxt = decode(q + 1);
        if (cur.first_chunk) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < VT; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
        }
        const char* xl = xb;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const bool more = !(q == nq - 1 && j == 2);
            // ---- issue next step's global loads (in flight during the MFMAs below) ----
            if (nxt.valid) load_x(nxt, j, xr[j], xok[j]);
            if (more) {
                if (j < 2) load_w(cur, j + 1); else load_w(nxt, 0);
            }
            // ---- MFMAs of this step: 9 taps x 2 k-slices ----
            const char* wl = wb;
            const int joff = j * PY * PX;
#pragma unroll 1
            for (int dyy = 0; dyy < 3; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 3; ++dxx) {
                const int tl = dyy * 3 + dxx;
                const int toff = joff + dyy * PX + dxx;              // arithmetic: no table lookup in the hot loop
                int bpos[VT];
#pragma unroll
                for (int vt = 0; vt < VT; ++vt) bpos[vt] = basepos[vt] + toff;
#pragma unroll
                for (int k16 = 0; k16 < 2; ++k16) {
                    half8 a[MT], b[VT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        a[mt] = *reinterpret_cast<const half8*>(wl + (tl * MB + mt * 32) * 64 + a_lane + (((k16 * 2 + hk) ^ a_swz) << 4));
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt)
                        b[vt] = *reinterpret_cast<const half8*>(xl + swz(bpos[vt], k16 * 2 + hk));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int vt = 0; vt < VT; ++vt)
                            acc[mt][vt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], b[vt], acc[mt][vt], 0, 0, 0);
                }
            }
            // ---- land the prefetched data: barrier (readers done) -> LDS writes -> barrier ----
            __syncthreads();
            if (more) store_w(wb);
            if (j == 2 && nxt.valid) {
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) store_x(xb, jj, xr[jj], xok[jj]);
            }
            __syncthreads();
        }
        if (cur.last_chunk) {
            // ---- epilogue: lane holds voxel (lane&31) x channels {8*q + 4*hk + 0..3} per accumulator quad ----
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int tile = wave * VT + vt;
                const int lz = cur.lz0 + tile / (TY / 4), ly = cur.ly0 + (tile % (TY / 4)) * 4 + (v >> 3), lx = cur.lx0 + (v & 7);
                if (lz >= p.Ld || ly >= p.Lh || lx >= p.Lw) continue;
                half_t* yrow = p.y + ((((long)cur.n * p.Do + lz) * p.Ho + ly) * p.Wo + lx) * p.ld_y;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int m = cur.m0 + mt * 32 + qq * 8 + hk * 4;
                        if (m >= p.M) continue;
                        float r0 = acc[mt][vt][qq * 4 + 0], r1 = acc[mt][vt][qq * 4 + 1], r2 = acc[mt][vt][qq * 4 + 2],
                              r3 = acc[mt][vt][qq * 4 + 3];
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
        }
        cur = nxt;
    }
}

template <int MT>
int launch(hipStream_t s, ConvParams& p, const char* name) {
    constexpr int MB = 32 * MT;
    p.tiles_z = lnn_cdiv(p.Ld, TZ); p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    const int mblocks = lnn_cdiv(p.M, MB);
    const long units = (long)p.N * p.tiles_z * p.tiles_y * p.tiles_x * mblocks;
    int num_cu = 256;
    static int cached_cu = 0;
    if (!cached_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached_cu = prop.multiProcessorCount;
        else cached_cu = 256;
    }
    num_cu = cached_cu;
    int upb = lnn_cdiv(units, 2 * num_cu);     // two resident blocks per CU
    if (upb < 1) upb = 1;
    const int grid = lnn_cdiv(units, upb);
    const size_t lds = XBYTES + (size_t)(9 * MB * 64);
    auto kern = igemm_conv_s1_v2_kernel<MT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, p, (int)units, mblocks, upb);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}

}  // namespace

int lnn_launch_conv_s1_v2(hipStream_t s, ConvParams& p, const char* name) {
    // MT = 1 everywhere: with two resident blocks per CU (<= 256 VGPRs per lane) the MT = 2 instance spills its
    // prefetch registers; output channels beyond 32 become additional work units instead (halo tile re-read from L2).
    return launch<1>(s, p, name);
}
