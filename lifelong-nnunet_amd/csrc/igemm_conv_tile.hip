// Stride-1 3x3x3 implicit-GEMM convolution, TILE kernel ("v5"): persistent blocks, double-buffered LDS, resident weights.
//
// Since round 2 the big layers run on the z-streaming kernel (igemm_conv_v9.hip) and since round 6 the deep ones on the macro-tile
// kernel (igemm_conv_mt.hip); this one serves what is left: narrow layers (< 128 input channels) on short volumes -- the levels of
// Hippocampus-sized plans, the toy shapes of the parity tests -- and odd channel counts.  It was the kernel behind ~90 % of the
// forward + dgrad FLOPs in round 1.  Same GEMM mapping as igemm_conv.hip (MFMA rows = output channels, columns = 32 voxels,
// v_mfma_f32_32x32x16_f16); everything around the MFMAs was shaped by measurements of the previous versions:
//   v2: PMC showed 12.6 VALU per MFMA and 51 % of LDS cycles lost to bank conflicts;
//   v3: s_memtime phase split: re-fetching the 9-tap weight panels every 36 MFMAs kept the waves in load issue;
//   v4: hipcc emitted read -> s_waitcnt lgkmcnt(0) -> MFMA (no read-ahead), and the single-buffered halo tile
//       made all 8 waves idle through barrier / LDS-store / barrier after every step.
// Structure now:
//   * ONE 512-thread block (8 waves, two per SIMD) per CU walks a contiguous range of work units
//     (8x8x8-voxel tile x 32 output channels); a step = (unit, 16-channel chunk) = 27 taps x 2 voxel tiles =
//     54 MFMAs per wave;
//   * everything in LDS is double buffered (halo tile 2 x 31.25 KB, weights 2 x 27 KB): the next step's global
//     loads are issued before the MFMAs, parked in registers, and written to the OTHER buffers right after the
//     wave's own MFMAs -- ONE barrier per step, no wave waits for another wave's stores;
//   * a weight slot is reloaded only when its (output-channel block, chunk) tag changes: layers with <= 32 input
//     channels load their weights once per output-channel block;
//   * LDS rows are 32 B (16 channels); the 16-byte half is XOR-keyed with the halo row parity (B operand) /
//     bit 3 of the output channel (A operand), and the lane -> voxel map below gives every ds_read_b128 16-lane
//     group two full 8-voxel rows: all fragment reads are bank-conflict free for all 27 taps;
//   * fragment reads run two MFMA groups ahead (explicit register triple buffering pinned with sched_barrier);
//   * addressing is strength-reduced: fragment reads are (per-lane base register [dx][row parity]) + immediate,
//     prefetch loads are (uniform tile base) + (per-thread precomputed 32-bit offset), bounds checks only on
//     tiles that touch the volume border.
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
    int n, lz0, ly0, lx0, m0, c0, ch;
    bool valid, first_chunk, last_chunk, interior;
};

__global__ __launch_bounds__(NT, 2) void igemm_conv_s1_v5_kernel(const ConvParams p, int units_total, int tiles_total,
                                                                 int units_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xb = smem;                   // 2 halo buffers
    char* const wb = smem + 2 * XBYTES;      // 2 weight slots

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v = lane & 31, hk = lane >> 5;
    const int u_begin = blockIdx.x * units_per_block;
    const int u_end = min(u_begin + units_per_block, units_total);
    if (u_begin >= u_end) return;
    const int nchunks = (p.C + CK - 1) / CK;
    // weight slot of a step: its chunk index when the layer has exactly two chunks (both stay resident), else the
    // step parity; a slot is refilled only when its tag (output-channel block, chunk) changes
    const bool by_chunk = nchunks == 2;
    const bool flip = p.taps.slot[0] != 0;   // dgrad: tap offset d' uses weight slot 26 - d'
    const int nq = (u_end - u_begin) * nchunks;

    // units are ordered output-channel-block major, tile minor: a block's consecutive units share their weights
    auto decode = [&](int q) {
        Step r;
        r.valid = q < nq;
        const int u = u_begin + q / nchunks;
        r.ch = q % nchunks;
        int t = u % tiles_total;
        r.m0 = (u / tiles_total) * MB;
        r.c0 = r.ch * CK;
        r.first_chunk = r.ch == 0;
        r.last_chunk = r.ch == nchunks - 1;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int tz = t % p.tiles_z; t /= p.tiles_z;
        r.n = t; r.lz0 = tz * TZ; r.ly0 = ty * TY; r.lx0 = tx * TX;
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
    // lterm[vt][dx][par]: byte address of (z = wave, y, x+dx), 16-byte half keyed with the parity of row y+par
    int lterm[VT][3][2];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int y = vt * 4 + vr;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int par = 0; par < 2; ++par)
                lterm[vt][dx][par] = ((wave * PY + y) * PX + vx + dx) * ROWB + ((hk ^ ((y + par) & 1)) << 4);
    }
    const int a_lane = v * ROWB + ((hk ^ ((v >> 3) & 1)) << 4);

    floatx16 acc[VT];
    unsigned long long ph[4] = {0, 0, 0, 0};   // debug: cycles in {issue, mfma, store, barrier}
    const bool prof = p.dbg != nullptr;

    // ---- prologue: first halo tile + first weight chunk, synchronously ---------------------------------------
    Step cur = decode(0);
    int wtag[2] = {-1, -1};
    load_x(cur);
    store_x(xb);
    {
        const int s0 = by_chunk ? cur.ch : 0;
        load_w(cur.m0, cur.c0);
        store_w(wb + s0 * WBYTES);
        if (s0) wtag[1] = cur.m0 * 4096 + cur.c0; else wtag[0] = cur.m0 * 4096 + cur.c0;
    }
    __syncthreads();

#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        const Step nxt = decode(q + 1);
        const int wslot = by_chunk ? cur.ch : (q & 1), nslot = by_chunk ? nxt.ch : ((q + 1) & 1);   // nslot != wslot
        const int ntag = nxt.m0 * 4096 + nxt.c0;
        const bool new_w = nxt.valid && (nslot ? wtag[1] : wtag[0]) != ntag;
        const char* xl = xb + (q & 1) * XBYTES;
        const char* wl = wb + wslot * WBYTES;
        if (cur.first_chunk) {
#pragma unroll
            for (int b = 0; b < VT; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        }
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
        if (prof) t0 = __builtin_amdgcn_s_memtime();
        // ---- issue the next step's global loads (in flight during the 54 MFMAs below) ----
        if (nxt.valid) load_x(nxt);
        if (new_w) load_w(nxt.m0, nxt.c0);
        if (prof) t1 = __builtin_amdgcn_s_memtime();
        // ---- MFMAs of this step.  The fragment reads of tap g+2 are issued BEFORE the MFMAs of tap g (explicit
        // register triple buffering): left to itself hipcc emits read -> s_waitcnt lgkmcnt(0) -> MFMA with no
        // read-ahead, which exposes the full LDS latency on every MFMA.
        half8 fa[3], fb[3][VT];
        auto frag = [&](int tl, half8& a, half8 (&b)[VT]) {     // tl compile-time after unrolling
            const int dz = tl / 9, dy = (tl / 3) % 3, dx = tl % 3;
            const int ximm = ((dz * PY + dy) * PX) * ROWB;
            a = *reinterpret_cast<const half8*>(wl + tl * MB * ROWB + a_lane);
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
                b[vt] = *reinterpret_cast<const half8*>(xl + ximm + lterm[vt][dx][dy & 1]);
        };
        frag(0, fa[0], fb[0]);
        frag(1, fa[1], fb[1]);
#pragma unroll
        for (int g = 0; g < 27; ++g) {
            if (g + 2 < 27) frag(g + 2, fa[(g + 2) % 3], fb[(g + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);      // keep the reads of g+2 above the MFMAs of g
#pragma unroll
            for (int vt = 0; vt < VT; ++vt)
                acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g % 3], fb[g % 3][vt], acc[vt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- land the prefetched data in the OTHER buffers (last read before the previous barrier) ----
        if (prof) t2 = __builtin_amdgcn_s_memtime();
        if (nxt.valid) store_x(xb + ((q + 1) & 1) * XBYTES);
        if (new_w) {
            store_w(wb + nslot * WBYTES);
            if (nslot) wtag[1] = ntag; else wtag[0] = ntag;
        }
        if (prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t3 = __builtin_amdgcn_s_memtime(); }
        __syncthreads();
        if (prof) {
            const unsigned long long t4 = __builtin_amdgcn_s_memtime();
            ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3;
        }
        if (cur.last_chunk) {
            // ---- epilogue: lane holds voxel (vr, vx) x channels {8*q + 4*hk + 0..3} per accumulator quad ----
#pragma unroll
            for (int vt = 0; vt < VT; ++vt) {
                const int lz = cur.lz0 + wave, ly = cur.ly0 + vt * 4 + vr, lx = cur.lx0 + vx;
                if (lz >= p.Ld || ly >= p.Lh || lx >= p.Lw) continue;
                const long yoff = ((((long)cur.n * p.Do + lz) * p.Ho + ly) * p.Wo + lx) * p.ld_y;
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
        cur = nxt;
    }
    if (prof && lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) atomicAdd(p.dbg + i, ph[i]);
        atomicAdd(p.dbg + 5, (unsigned long long)nq);
    }
}

}  // namespace

int lnn_launch_conv_s1_tile(hipStream_t s, ConvParams& p, const char* name) {
    p.tiles_z = lnn_cdiv(p.Ld, TZ); p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    const int mblocks = lnn_cdiv(p.M, MB);
    const int tiles = p.N * p.tiles_z * p.tiles_y * p.tiles_x;
    const long units = (long)tiles * mblocks;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int upb = lnn_cdiv(units, num_cu);     // one resident 8-wave block per CU (116.5 KB of LDS)
    if (upb < 1) upb = 1;
    const int grid = lnn_cdiv(units, upb);
    const size_t lds = 2 * XBYTES + 2 * WBYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_conv_s1_v5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm_conv_s1_v5_kernel, dim3(grid), dim3(NT), lds, s, p, (int)units, tiles, upb);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}
