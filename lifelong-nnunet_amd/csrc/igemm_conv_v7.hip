// Stride-1 3x3x3 implicit-GEMM convolution, v7: TWO 8-wave blocks per CU (experiment, LNN_CONV_V7=1).
//
// v5 (igemm_conv_v2.hip) keeps one block per CU busy ~36 % of the matrix-pipe cycles: its eight waves run in lockstep,
// so the load-issue, LDS-store and barrier phases of a step leave the pipe idle.  v7 gives up the in-block double
// buffering (LDS: one 31 KB halo tile + one 27 KB weight chunk = 58 KB, registers <= 128) so that two blocks are
// resident per CU and the hardware interleaves one block's staging phase with the other block's MFMA phase.
// Same GEMM mapping, LDS layout, lane -> voxel map and blocked weight panels as v5.
#include "igemm_common.h"

namespace {

constexpr int TZ = 8, TY = 8, TX = 8, PZ = 10, PY = 10, PX = 10;
constexpr int P = PZ * PY * PX;             // 1000 halo positions
constexpr int CK = 16;                      // input channels per step
constexpr int ROWB = CK * 2;                // 32-byte LDS rows
constexpr int XBYTES = P * ROWB;            // 32000
constexpr int XCHUNKS = P * 2;              // 16-byte chunks per halo tile
constexpr int NT = 512;
constexpr int XN = (XCHUNKS + NT - 1) / NT; // 4 loads per thread per step
constexpr int MB = 32, VT = 2;
constexpr int WBYTES = 27 * MB * ROWB, WCHUNKS = 27 * MB * 2, WN = (WCHUNKS + NT - 1) / NT;  // 27648 B, 4 loads

__device__ __forceinline__ int xaddr(int pz, int py, int px, int c2) {
    return ((pz * PY + py) * PX + px) * ROWB + ((c2 ^ (py & 1)) << 4);
}
__device__ __forceinline__ int waddr(int row, int c2) { return row * ROWB + ((c2 ^ ((row >> 3) & 1)) << 4); }

// lane (0..31) -> (row 0..3, x 0..7) inside a 32-voxel MFMA tile.  ds_read_b128 is serviced in the 16-lane groups
// {0-3,12-15,20-27} and {4-11,16-19,28-31}: each group gets two full 8-voxel rows (= one 256-byte bank row each,
// the second shifted by 64 B and separated by the row-parity key).
__device__ __forceinline__ void lane_voxel(int v, int& r, int& x) {
    if (v < 4) { r = 0; x = v; }
    else if (v < 12) { r = 2; x = v - 4; }
    else if (v < 16) { r = 0; x = v - 8; }
    else if (v < 20) { r = 3; x = v - 16; }
    else if (v < 28) { r = 1; x = v - 20; }
    else { r = 3; x = v - 24; }
}

struct Step {   // (work unit, 16-channel chunk)
    int n, lz0, ly0, lx0, m0, c0, ch, part, zlim;
    bool valid, first_chunk, last_chunk, interior;
};

__global__ __launch_bounds__(NT, 4) void igemm_conv_s1_v7_kernel(const ConvParams p, int units_total, int tiles_total,
                                                                 int units_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xb = smem;                   // halo tile
    char* const wb = smem + XBYTES;          // weight chunk

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = lane & 31, hk = lane >> 5;
    const int u_begin = blockIdx.x * units_per_block;
    const int u_end = min(u_begin + units_per_block, units_total);
    if (u_begin >= u_end) return;
    const int nchunks = (p.C + CK - 1) / CK;
    const bool flip = p.taps.slot[0] != 0;   // dgrad: tap offset d' uses weight slot 26 - d'
    // split-K (small layers: fewer units than resident blocks, a long serial chunk loop per unit): unit index u' = u * ksplit +
    // part, part owns nsteps = nchunks / ksplit consecutive chunks and writes its partial sums into ITS slice of the fp32
    // scratch tensor [ksplit][voxel][Mpad] (plain 16-byte stores: float atomics from 8 XCDs onto one tensor measured slower
    // than no split at all); lnn_launch_splitk_finalize adds the slices in a fixed order
    const int ks = p.ksplit, nsteps = nchunks / ks;
    const int nq = (u_end - u_begin) * nsteps;

    // units are ordered output-channel-block major, tile minor: a block's consecutive units share their weights
    auto decode = [&](int q) {
        Step r;
        r.valid = q < nq;
        const int up = u_begin + q / nsteps, st = q % nsteps;
        const int u = up / ks;
        r.part = up % ks;
        r.ch = r.part * nsteps + st;
        int t = u % tiles_total;
        r.m0 = (u / tiles_total) * MB;
        r.c0 = r.ch * CK;
        r.first_chunk = st == 0;
        r.last_chunk = st == nsteps - 1;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int tz = t % p.tiles_z; t /= p.tiles_z;
        r.n = t; r.lz0 = tz * TZ; r.ly0 = ty * TY; r.lx0 = tx * TX;
        r.zlim = p.Ld - r.lz0;                    // planes of this tile inside the volume (>= TZ: all)
        // whole halo inside the volume and a full 16-channel chunk -> no per-element checks needed
        r.interior = r.lz0 >= 1 && r.ly0 >= 1 && r.lx0 >= 1 && r.lz0 + TZ + 1 <= p.Di && r.ly0 + TY + 1 <= p.Hi &&
                     r.lx0 + TX + 1 <= p.Wi && r.c0 + CK <= p.C;
        return r;
    };

    // ---- per-thread staging constants (tile independent) -------------------------------------------------
    int xrel[XN], xlds[XN];
#pragma unroll
    for (int i = 0; i < XN; ++i) {
        const int idx = min(i * NT + tid, XCHUNKS - 1);
        const int pos = idx >> 1, c2 = idx & 1;
        const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
        xrel[i] = ((pz * p.Hi + py) * p.Wi + px) * p.ld_x + c2 * 8;
        xlds[i] = xaddr(pz, py, px, c2);
    }
    int wrel[WN], wlds[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int idx = min(i * NT + tid, WCHUNKS - 1);
        const int c2 = idx & 1, row = idx >> 1, r = row % MB, tl = row / MB;
        wrel[i] = ((flip ? 26 - tl : tl) * MB + r) * 16 + c2 * 8;   // blocked panel: one contiguous 27 KB run
        wlds[i] = waddr(row, c2);
    }

    // Prefetch registers.  The global loads are UNCONDITIONAL (out-of-range lanes read element 0 of the tensor and
    // are zeroed when the value is written to LDS): a predicated load makes hipcc wrap each one in an exec-mask
    // branch with s_waitcnt vmcnt(0) in front, which serialises the whole prefetch.
    half8 xr[XN], wr[WN];
    unsigned xok = 0;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_x = [&](const Step& t) {
        // channel concatenation of two tensors (lnn_conv3d_*_cat): chunks below csplit come from x, the rest from x2
        const bool part2 = t.c0 >= p.csplit;
        const half_t* const xp = part2 ? p.x2 : p.x;
        const long base = ((((long)t.n * p.Di + (t.lz0 - 1)) * p.Hi + (t.ly0 - 1)) * p.Wi + (t.lx0 - 1)) * p.ld_x + (part2 ? t.c0 - p.csplit : t.c0);
        if (t.interior) {
            const half_t* bp = xp + base;
#pragma unroll
            for (int i = 0; i < XN; ++i) xr[i] = *reinterpret_cast<const half8*>(bp + xrel[i]);
            xok = 0xFFFFu;
        } else {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < XN; ++i) {
                const int idx = min(i * NT + tid, XCHUNKS - 1);
                const int pos = idx >> 1, c2 = idx & 1;
                const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
                const int iz = t.lz0 - 1 + pz, iy = t.ly0 - 1 + py, ix = t.lx0 - 1 + px;
                const bool ok = (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi &&
                                t.c0 + c2 * 8 < p.C;
                xr[i] = *reinterpret_cast<const half8*>(xp + (ok ? base + xrel[i] : 0));
                m |= (ok ? 1u : 0u) << i;
            }
            xok = m;
        }
    };
    auto store_x = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            if (i * NT + tid < XCHUNKS) *reinterpret_cast<half8*>(buf + xlds[i]) = ((xok >> i) & 1u) ? xr[i] : zero8;
        }
    };
    auto load_w = [&](int m0, int c0) {     // KCpad is a multiple of 16: a chunk never leaves the padded panel row
        const half_t* bp = p.wp + lnn_panel_off(0, m0, c0, 27, p.KCpad);
#pragma unroll
        for (int i = 0; i < WN; ++i) wr[i] = *reinterpret_cast<const half8*>(bp + wrel[i]);
    };
    auto store_w = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            if (i * NT + tid < WCHUNKS) *reinterpret_cast<half8*>(buf + wlds[i]) = wr[i];
        }
    };

    // ---- per-lane fragment addressing -------------------------------------------------------------------------
    int vr, vx;
    lane_voxel(v, vr, vx);
    // lterm[vt][par]: byte address of (z = wave, y, x), 16-byte half keyed with the parity of row y+par (dx is an immediate)
    int lterm[VT][2];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int y = vt * 4 + vr;
#pragma unroll
        for (int par = 0; par < 2; ++par)
            lterm[vt][par] = ((wave * PY + y) * PX + vx) * ROWB + ((hk ^ ((y + par) & 1)) << 4);
    }
    const int a_lane = v * ROWB + ((hk ^ ((v >> 3) & 1)) << 4);

    floatx16 acc[VT];

    int wtag = -1;
#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        const Step cur = decode(q);
        const int tag = cur.m0 * 4096 + cur.c0;
        const bool new_w = wtag != tag;
        // ---- stage this step's halo tile / weight chunk: loads first, then wait for the previous step's readers
        load_x(cur);
        if (new_w) load_w(cur.m0, cur.c0);
        __syncthreads();
        store_x(xb);
        if (new_w) { store_w(wb); wtag = tag; }
        __syncthreads();
        const char* xl = xb;
        const char* wl = wb;
        if (cur.first_chunk) {
#pragma unroll
            for (int b = 0; b < VT; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        }
        // A wave owns one z plane of the 8x8x8 tile.  Where the volume ends inside the tile (20 planes = 8 + 8 + 4 at the fourth
        // level of the 160x192x160 plan, 5 planes at the sixth) the waves of the padding planes skip the 54 MFMAs and their 81
        // fragment reads of the step -- they only take part in the staging and its barriers -- and leave the matrix pipe and the
        // LDS port to the second block resident on the CU (round 5: 19.99 / 20.00 vs 20.05 / 20.04 ms per C2 step, alternating runs;
        // enc3.1 dgrad 0.120-0.125 vs 0.122-0.136 ms).  Also measured: "half steps" -- for tiles with <= 4 live planes the two MFMA
        // tiles of a plane on two waves, 27 MFMAs per wave -- with the second tile's read + MFMA behind a uniform branch in this one
        // loop: 13 % SLOWER (enc3.1 fwd 0.120-0.124 vs 0.105-0.107 ms, step 19.98 vs 19.86 ms; 51 branches in the MFMA loop); as two
        // loops: 44 spilled registers at the 128 this kernel may use.  Not kept.
        const bool zlive = wave < cur.zlim;
        half8 fa[3], fb[3][VT];
        auto frag = [&](int tl, half8& a, half8 (&b)[VT]) {     // tl compile-time after unrolling
            const int dz = tl / 9, dy = (tl / 3) % 3, dx = tl % 3;
            const int ximm = ((dz * PY + dy) * PX + dx) * ROWB;
            a = *reinterpret_cast<const half8*>(wl + tl * MB * ROWB + a_lane);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
                b[vt] = *reinterpret_cast<const half8*>(xl + ximm + lterm[vt][dy & 1]);
        };
        if (zlive) {
            frag(0, fa[0], fb[0]);
            frag(1, fa[1], fb[1]);
#pragma unroll
            for (int g = 0; g < 27; ++g) {
                if (g + 2 < 27) frag(g + 2, fa[(g + 2) % 3], fb[(g + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int vt = 0; vt < VT; ++vt)
                    acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g % 3], fb[g % 3][vt], acc[vt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (cur.last_chunk) {
            // ---- epilogue: lane holds voxel (vr, vx) x channels {8*q + 4*hk + 0..3} per accumulator quad ----
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int lz = cur.lz0 + wave, ly = cur.ly0 + vt * 4 + vr, lx = cur.lx0 + vx;
                if (lz >= p.Ld || ly >= p.Lh || lx >= p.Lw) continue;
                const long vox = (((long)cur.n * p.Do + lz) * p.Ho + ly) * p.Wo + lx;
                if (ks > 1) {
                    float* srow = p.scratch + ((long)cur.part * ((long)p.N * p.Do * p.Ho * p.Wo) + vox) * p.Mpad;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int m = cur.m0 + qq * 8 + hk * 4;
                        if (m >= p.Mpad) continue;
                        const floatx4 w4 = {acc[vt][qq * 4 + 0], acc[vt][qq * 4 + 1], acc[vt][qq * 4 + 2], acc[vt][qq * 4 + 3]};
                        *reinterpret_cast<floatx4*>(srow + m) = w4;
                    }
                    continue;
                }
                const long yoff = vox * p.ld_y;
                half_t* yrow = p.y + yoff;
                half_t* yrow2 = p.y2 + yoff - p.msplit;      // output channels >= msplit go to the second tensor
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int m = cur.m0 + qq * 8 + hk * 4;
                    if (m >= p.M) continue;
                    float r0 = acc[vt][qq * 4 + 0], r1 = acc[vt][qq * 4 + 1], r2 = acc[vt][qq * 4 + 2], r3 = acc[vt][qq * 4 + 3];
                    if (p.bias) {
                        const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + m);
                        r0 += bv[0]; r1 += bv[1]; r2 += bv[2]; r3 += bv[3];
                    }
                    half4* dst = reinterpret_cast<half4*>((m < p.msplit ? yrow : yrow2) + m);
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
}

// scratch (fp32, [ksplit][voxel][Mpad]) -> y (fp16, slices added in order, + bias, + old value when accumulating)
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const ConvParams p, long nvox) {
    const int q4 = p.Mpad >> 2;
    const long total = nvox * q4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long vox = i / q4;
        const int m = (int)(i % q4) * 4;
        if (m >= p.M) continue;
        floatx4 r = *reinterpret_cast<const floatx4*>(p.scratch + vox * p.Mpad + m);
        for (int k = 1; k < p.ksplit; ++k) r += *reinterpret_cast<const floatx4*>(p.scratch + ((long)k * nvox + vox) * p.Mpad + m);
        if (p.bias) {
            const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + m);
            r += bv;
        }
        half_t* yrow = (m < p.msplit ? p.y : p.y2 - p.msplit) + vox * p.ld_y;
        half4* dst = reinterpret_cast<half4*>(yrow + m);
        if (p.accumulate) {
            const half4 old = *dst;
            r[0] += (float)old[0]; r[1] += (float)old[1]; r[2] += (float)old[2]; r[3] += (float)old[3];
        }
        const half4 o = {(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3]};
        *dst = o;
    }
}

}  // namespace

int lnn_launch_splitk_finalize(hipStream_t s, const ConvParams& p, const char* name) {
    const long nvox = (long)p.N * p.Do * p.Ho * p.Wo;
    const long total = nvox * (p.Mpad >> 2);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_finalize_kernel, dim3(blocks), dim3(256), 0, s, p, nvox);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}

int lnn_launch_conv_s1_v7(hipStream_t s, ConvParams& p, const char* name) {
    p.tiles_z = lnn_cdiv(p.Ld, TZ); p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    const int mblocks = lnn_cdiv(p.M, MB);
    const int tiles = p.N * p.tiles_z * p.tiles_y * p.tiles_x;
    const long units = (long)tiles * mblocks * p.ksplit;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int upb = lnn_cdiv(units, 2 * num_cu); // two resident 8-wave blocks per CU (58 KB of LDS each)
    if (upb < 1) upb = 1;
    const int grid = lnn_cdiv(units, upb);
    const size_t lds = XBYTES + WBYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_conv_s1_v7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm_conv_s1_v7_kernel, dim3(grid), dim3(NT), lds, s, p, (int)units, tiles, upb);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}
