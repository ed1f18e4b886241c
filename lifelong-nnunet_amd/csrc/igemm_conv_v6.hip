// Stride-1 3x3x3 implicit-GEMM convolution, "ping-pong" schedule (v6).
//
// Measurements of v5 (igemm_conv_v2.hip) with the s_memtime phase hook: only ~47 % of a wave's time is its MFMA
// phase; load issue, LDS stores and the step barrier make up the rest, and because all 8 waves of the block run
// in lockstep the matrix pipes idle during those phases (36 % MFMA pipe utilisation in cycles).
//
// v6 splits the 8 waves of the block into two groups of 4 (one wave per SIMD each) that run HALF A STEP OUT OF
// PHASE: in every slot one group computes (54 MFMAs per wave on its own 4x8x8-voxel tile) while the other group
// does all of its memory work (LDS stores of the tile it loaded one slot earlier, global-load issue for the tile
// after that, output stores); a block-wide barrier separates the slots and the roles swap.  The matrix pipe of
// every SIMD therefore always has exactly one wave in an MFMA phase.
//   * per group: one halo buffer (600 positions x 32 B) -- it is written only in the group's memory slots and
//     read only in its compute slots, so no double buffering is needed;
//   * weights: layers with <= 32 input channels keep both 16-channel chunks resident (loaded once, a block never
//     crosses an output-channel-block boundary); wider layers stream one chunk per step into the group's own slot;
//   * LDS layout, lane -> voxel map, strength-reduced addressing and pinned fragment read-ahead as in v5.
#include "igemm_common.h"

namespace {

constexpr int TZ = 4, TY = 8, TX = 8, PZ = 6, PY = 10, PX = 10;
constexpr int P = PZ * PY * PX;             // 600 halo positions
constexpr int CK = 16, ROWB = CK * 2;       // 16 channels = 32-byte LDS rows
constexpr int XBYTES = P * ROWB;            // 19200
constexpr int XCHUNKS = P * 2;              // 1200 16-byte chunks
constexpr int GT = 256, NT = 512;           // threads per group / block
constexpr int XN = (XCHUNKS + GT - 1) / GT; // 5 loads per thread
constexpr int MB = 32, VT = 2;
constexpr int WBYTES = 27 * MB * ROWB, WCHUNKS = 27 * MB * 2, WN = (WCHUNKS + GT - 1) / GT;  // 27648 B, 7 loads

__device__ __forceinline__ int xaddr(int pz, int py, int px, int c2) {
    return ((pz * PY + py) * PX + px) * ROWB + ((c2 ^ (py & 1)) << 4);
}
__device__ __forceinline__ int waddr(int row, int c2) { return row * ROWB + ((c2 ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ void lane_voxel(int v, int& r, int& x) {   // see igemm_conv_v2.hip
    if (v < 4) { r = 0; x = v; }
    else if (v < 12) { r = 2; x = v - 4; }
    else if (v < 16) { r = 0; x = v - 8; }
    else if (v < 20) { r = 3; x = v - 16; }
    else if (v < 28) { r = 1; x = v - 20; }
    else { r = 3; x = v - 24; }
}

struct Step {
    int n, lz0, ly0, lx0, c0, ch;
    bool valid, first_chunk, last_chunk, interior;
};

__global__ __launch_bounds__(NT, 2) void igemm_conv_s1_v6_kernel(const ConvParams p, int tiles_total, int blocks_per_mb,
                                                                 int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, gtid = tid & (GT - 1), gw = wave & 3;
    char* const xg = smem + grp * XBYTES;        // this group's halo buffer
    char* const wb = smem + 2 * XBYTES;          // 2 weight slots
    const int v = lane & 31, hk = lane >> 5;

    const int mb = blockIdx.x / blocks_per_mb, bi = blockIdx.x % blocks_per_mb;
    const int m0 = mb * MB;
    const int t_begin = bi * tiles_per_block, t_end = min(t_begin + tiles_per_block, tiles_total);
    if (t_begin >= t_end) return;
    const int nchunks = (p.C + CK - 1) / CK;
    const bool resident = nchunks <= 2;
    const bool flip = p.taps.slot[0] != 0;
    // group g takes tiles t_begin + g, t_begin + g + 2, ...
    const int ntiles_g = (t_end - t_begin - grp + 1) / 2;
    const int nsteps = ntiles_g * nchunks;                                  // this group's steps
    const int nsteps_max = ((t_end - t_begin + 1) / 2) * nchunks;           // group 0's (>= group 1's)

    auto decode = [&](int k) {
        Step r;
        r.valid = k < nsteps;
        int t = t_begin + 2 * (k / nchunks) + grp;
        r.ch = k % nchunks;
        r.c0 = r.ch * CK;
        r.first_chunk = r.ch == 0;
        r.last_chunk = r.ch == nchunks - 1;
        const int tx = t % p.tiles_x; t /= p.tiles_x;
        const int ty = t % p.tiles_y; t /= p.tiles_y;
        const int tz = t % p.tiles_z; t /= p.tiles_z;
        r.n = t; r.lz0 = tz * TZ; r.ly0 = ty * TY; r.lx0 = tx * TX;
        r.interior = r.lz0 >= 1 && r.ly0 >= 1 && r.lx0 >= 1 && r.lz0 + TZ + 1 <= p.Di && r.ly0 + TY + 1 <= p.Hi &&
                     r.lx0 + TX + 1 <= p.Wi && r.c0 + CK <= p.C;
        return r;
    };

    // ---- per-thread staging constants ----
    int xrel[XN], xlds[XN];
#pragma unroll
    for (int i = 0; i < XN; ++i) {
        const int idx = min(i * GT + gtid, XCHUNKS - 1);
        const int pos = idx >> 1, c2 = idx & 1;
        const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
        xrel[i] = ((pz * p.Hi + py) * p.Wi + px) * p.ld_x + c2 * 8;
        xlds[i] = xaddr(pz, py, px, c2);
    }
    int wrel[WN], wlds[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int idx = min(i * GT + gtid, WCHUNKS - 1);
        const int c2 = idx & 1, row = idx >> 1, r = row % MB, tl = row / MB;
        wrel[i] = ((flip ? 26 - tl : tl) * MB + r) * 16 + c2 * 8;   // blocked panel: one contiguous 27 KB run
        wlds[i] = waddr(row, c2);
    }

    half8 xr[XN], wr[WN];
    unsigned xok = 0;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_x = [&](const Step& t) {     // unconditional loads + validity mask (see igemm_conv_v2.hip)
        const long base = ((((long)t.n * p.Di + (t.lz0 - 1)) * p.Hi + (t.ly0 - 1)) * p.Wi + (t.lx0 - 1)) * p.ld_x + t.c0;
        if (t.interior) {
            const half_t* bp = p.x + base;
#pragma unroll
            for (int i = 0; i < XN; ++i) xr[i] = *reinterpret_cast<const half8*>(bp + xrel[i]);
            xok = 0xFFFFu;
        } else {
            unsigned m = 0;
#pragma unroll
            for (int i = 0; i < XN; ++i) {
                const int idx = min(i * GT + gtid, XCHUNKS - 1);
                const int pos = idx >> 1, c2 = idx & 1;
                const int px = pos % PX, py = (pos / PX) % PY, pz = pos / (PX * PY);
                const int iz = t.lz0 - 1 + pz, iy = t.ly0 - 1 + py, ix = t.lx0 - 1 + px;
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
        for (int i = 0; i < XN; ++i) {
            if (i * GT + gtid < XCHUNKS) *reinterpret_cast<half8*>(xg + xlds[i]) = ((xok >> i) & 1u) ? xr[i] : zero8;
        }
    };
    auto load_w = [&](int c0) {
        const half_t* bp = p.wp + lnn_panel_off(0, m0, c0, 27, p.KCpad);
#pragma unroll
        for (int i = 0; i < WN; ++i) wr[i] = *reinterpret_cast<const half8*>(bp + wrel[i]);
    };
    auto store_w = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            if (i * GT + gtid < WCHUNKS) *reinterpret_cast<half8*>(buf + wlds[i]) = wr[i];
        }
    };

    // ---- per-lane fragment addressing ----
    int vr, vx;
    lane_voxel(v, vr, vx);
    int lterm[VT][3][2];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int y = vt * 4 + vr;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int par = 0; par < 2; ++par)
                lterm[vt][dx][par] = ((gw * PY + y) * PX + vx + dx) * ROWB + ((hk ^ ((y + par) & 1)) << 4);
    }
    const int a_lane = v * ROWB + ((hk ^ ((v >> 3) & 1)) << 4);
    floatx16 acc[VT];

    // ---- prologue ----
    // resident weights: chunk c lives in slot c for the whole kernel (group g loads chunk g)
    if (resident && grp < nchunks) {
        load_w(grp * CK);
        store_w(wb + grp * WBYTES);
    }
    // group 0 computes in even slots: its step 0 must be in LDS now, step 1 is put in flight.
    // group 1 computes in odd slots: its step 0 is put in flight and lands in its first memory slot (slot 0).
    int kp = 0;                 // step whose data sits in the prefetch registers
    bool wp_valid = false;      // prefetch registers also hold that step's (streamed) weights
    Step pend = decode(0);
    if (grp == 0) {
        load_x(pend);
        store_x();
        if (!resident) { load_w(pend.c0); store_w(wb); }
        kp = 1;
        pend = decode(1);
    }
    if (pend.valid) {
        load_x(pend);
        if (!resident) { load_w(pend.c0); wp_valid = true; }
    }
    __syncthreads();

    const int nslots = 2 * nsteps_max + 1;
#pragma unroll 1
    for (int t = 0; t < nslots; ++t) {
        if ((t & 1) == grp) {
            // =============== compute slot: step k of this group ===============
            const int k = t >> 1;
            const Step cur = decode(k);
            if (cur.valid) {
                const char* wl = wb + (resident ? cur.ch : grp) * WBYTES;
                if (cur.first_chunk) {
#pragma unroll
                    for (int b = 0; b < VT; ++b)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
                }
                constexpr int RA = 4;                 // fragment read-ahead depth (MFMA groups)
                half8 fa[RA + 1], fb[RA + 1][VT];
                auto frag = [&](int tl, half8& a, half8 (&b)[VT]) {
                    const int dz = tl / 9, dy = (tl / 3) % 3, dx = tl % 3;
                    const int ximm = ((dz * PY + dy) * PX) * ROWB;
                    a = *reinterpret_cast<const half8*>(wl + tl * MB * ROWB + a_lane);
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt)
                        b[vt] = *reinterpret_cast<const half8*>(xg + ximm + lterm[vt][dx][dy & 1]);
                };
#pragma unroll
                for (int g = 0; g < RA; ++g) frag(g, fa[g], fb[g]);
#pragma unroll
                for (int g = 0; g < 27; ++g) {
                    if (g + RA < 27) frag(g + RA, fa[(g + RA) % (RA + 1)], fb[(g + RA) % (RA + 1)]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt)
                        acc[vt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[g % (RA + 1)], fb[g % (RA + 1)][vt], acc[vt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (cur.last_chunk) {
#pragma unroll
                    for (int vt = 0; vt < VT; ++vt) {
                        const int lz = cur.lz0 + gw, ly = cur.ly0 + vt * 4 + vr, lx = cur.lx0 + vx;
                        if (lz >= p.Ld || ly >= p.Lh || lx >= p.Lw) continue;
                        half_t* yrow = p.y + ((((long)cur.n * p.Do + lz) * p.Ho + ly) * p.Wo + lx) * p.ld_y;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int m = m0 + qq * 8 + hk * 4;
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
            }
        } else {
            // =============== memory slot: land step kp (computed next slot), put step kp+1 in flight ===============
            if (pend.valid) {
                store_x();
                if (wp_valid) store_w(wb + grp * WBYTES);
            }
            ++kp;
            pend = decode(kp);
            wp_valid = false;
            if (pend.valid) {
                load_x(pend);
                if (!resident) { load_w(pend.c0); wp_valid = true; }
            }
        }
        __syncthreads();
    }
}

}  // namespace

int lnn_launch_conv_s1_v6(hipStream_t s, ConvParams& p, const char* name) {
    p.tiles_z = lnn_cdiv(p.Ld, TZ); p.tiles_y = lnn_cdiv(p.Lh, TY); p.tiles_x = lnn_cdiv(p.Lw, TX);
    const int mblocks = lnn_cdiv(p.M, MB);
    const int tiles = p.N * p.tiles_z * p.tiles_y * p.tiles_x;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    // one block per CU; a block never crosses an output-channel-block boundary (resident weights stay valid)
    int bpm = num_cu / mblocks;
    if (bpm < 1) bpm = 1;
    int tpb = lnn_cdiv(tiles, bpm);
    if (tpb < 2) tpb = tiles >= 2 ? 2 : 1;      // both groups want a tile
    bpm = lnn_cdiv(tiles, tpb);
    const size_t lds = 2 * XBYTES + 2 * WBYTES;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_conv_s1_v6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm_conv_s1_v6_kernel, dim3(bpm * mblocks), dim3(NT), lds, s, p, tiles, bpm, tpb);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}
