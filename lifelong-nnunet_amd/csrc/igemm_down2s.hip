// Stride-2 3x3x3 convolution forward (pad 1), z-streaming with register-resident weights: the v9 recipe
// (igemm_conv_v9.hip) applied to the resolution-halving conv of the encoder
// (nn.Conv3d stride 2, test/network_architecture/test_MultiHead_Module.py:346-415) for 32 / 64 input channels.
//
// The tile kernel it replaces (igemm_down2.hip) stages a 9x17x17 input tile per 16-channel chunk through registers: every
// chunk step re-touches all lines of the tile and the set exceeds the XCD's L2 -- 1.58 GB fetched for a 0.63 GB input
// (DESIGN.md section 4).  Here a block owns a column of FY x FX OUTPUT voxels and walks it along z; every input plane
// (2 FY + 1) x (2 FX + 1) positions, ALL channels) enters an LDS ring once by direct-to-LDS buffer loads and is used for
// everything it contributes to:
//   * even input plane 2 zo     : taps dz = 1 of output plane zo              ->  9 B fragments,  9 MFMAs per wave
//   * odd  input plane 2 zo + 1 : taps dz = 2 of zo and dz = 0 of zo + 1      ->  9 B fragments, 18 MFMAs per wave
// (two rolling accumulators); one plane per step, one barrier per step, 3 - 4 planes in flight exactly as in v9.
// The kernel is bound by the input read (4x the output; ~200 FLOP per byte), so what matters is that each input byte
// crosses the fabric once: the in-plane halo (17 / 16 per axis) is an L2 hit between neighbouring columns of one XCD.
// LDS layout of a plane slab: [32-channel group][py][x parity][pxh (row stride PXHS)][4 x 16 bytes]: the input row is
// de-interleaved by x parity, so tap dx of output voxel ox reads parity (dx & 1) at position ox + (dx >> 1) -- consecutive
// positions for consecutive lanes, the stride-1 kernel's conflict-free pattern.  The 16-byte piece index is XOR-keyed with
// ((pxh >> 2) & 1) | (((py >> 1) & 1) << 1); the DMA writes LDS linearly, so the key (and the de-interleave) is applied to
// the SOURCE address each lane fetches.  Wave roles, weight-row rotation, partial-sum exchange, v_permlane32_swap epilogue
// and the InstanceNorm-statistics epilogue are those of v9 (NCK chunks x NMB output blocks x NF footprints = 8 waves).
// EXT = 2: the data gradient of the 2x2x2 stride-2 transposed conv (`tu`, nnViTUNetTrainer.py:122) is the same gather with 8 taps
// and no padding: out[l] = sum_{d in {0,1}^3} W[d] in[2 l + d] -- even plane: taps dz = 0, odd plane: dz = 1, one accumulator,
// no overlap between output planes (4 MFMAs per wave and plane: purely bound by the 4x larger input).
#include "igemm_common.h"

namespace {

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

template <int V>
struct IC { static constexpr int value = V; };

template <int N, bool LGKM>
__device__ __forceinline__ void wait_vm() {           // literal counts only (see igemm_conv_v9.hip)
    static_assert(N >= 0 && N <= 6, "extend the table");
    if constexpr (LGKM) {
        if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    } else {
        if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_wave_base, 16, voffset, 0, 0, 0);
}

__device__ __forceinline__ void lane_voxel_d(int v, int& r, int& x) {   // see igemm_conv_tile.hip
    if (v < 4) { r = 0; x = v; }
    else if (v < 12) { r = 2; x = v - 4; }
    else if (v < 16) { r = 0; x = v - 8; }
    else if (v < 20) { r = 3; x = v - 16; }
    else if (v < 28) { r = 1; x = v - 20; }
    else { r = 3; x = v - 24; }
}

template <int NCK_, int NMB_, int NF_, int EXT_ = 3>
struct D2 {
    static constexpr int NCK = NCK_, NMB = NMB_, NF = NF_, EXT = EXT_;
    static_assert(EXT == 2 || EXT == 3, "3x3x3 pad 1 or 2x2x2 pad 0");
    static constexpr int PAD = EXT == 3 ? 1 : 0;
    static constexpr int LEND = EXT == 3 ? 1 : 0;            // walked outputs in front of a segment that only lend their odd plane
    static constexpr int NIP = EXT * EXT, NTAPS = EXT * EXT * EXT;   // in-plane taps, taps
    static constexpr int NB = NIP % 3 == 0 ? 3 : 2;          // B-fragment register ring (reads run NB - 1 ahead of the MFMAs)
    static_assert(NCK * NMB * NF == 8 && (NCK == 2 || NCK == 4), "8 waves per block");
    static constexpr int NG = NCK / 2;                       // 32-channel groups of the input
    static constexpr int NFX = NF >= 4 ? 2 : 1, NFY = NF / NFX;
    static constexpr int FY = 4 * NFY, FX = 8 * NFX;         // block footprint (output voxels)
    static constexpr int PY = 2 * FY + EXT - 2, PX = 2 * FX + EXT - 2;   // input patch of a plane
    static constexpr int PXH = (PX + 1) / 2, PXHS = (PXH + 1) / 2 * 2;   // positions per x-parity row; even stride: rows start 256-byte aligned
    static constexpr int GRAW = PY * 2 * PXHS * 64;          // bytes of one group slab
    static constexpr int DPW = (NG * ((GRAW + 1023) / 1024) + 7) / 8;   // DMA instructions per wave per plane
    static constexpr int GSLAB = DPW * 8 / NG * 1024;
    static_assert(GSLAB >= GRAW && (DPW * 8) % NG == 0, "slab padding");
    static constexpr int PLANE = NG * GSLAB;
    static constexpr int D = NCK == 4 ? 3 : 4, R = D + 1;    // planes in flight / ring slots (64 channels: 4 x 24 KB + 2 x 24 KB of exchange)
    static constexpr int QN = 4 / NCK;                       // accumulator quads a wave finalises
    static constexpr int EXB = NF * NMB * NCK * (NCK - 1) * QN * 1024;   // one partial-sum exchange buffer
    static constexpr int LDS = R * PLANE + 2 * EXB;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(D * DPW - 2 * DPW <= 6, "wait_vm table");
};

struct D2Launch {
    int items, S, L, tiles_y, tiles_x, nslots, ipx;
};

// p: x / ld_x / Di,Hi,Wi = input, y / ld_y / Do,Ho,Wo = output (= Ld,Lh,Lw), p.C in {32, 64}, p.M % (32 NMB) == 0, pad 1.
template <int NCK_, int NMB_, int NF_, int EXT_, bool STATS>
__global__ __launch_bounds__(512, 2) void igemm_down2s_kernel(const ConvParams p, const D2Launch q) {
    using K = D2<NCK_, NMB_, NF_, EXT_>;
    constexpr int EXT = K::EXT, PAD = K::PAD, LEND = K::LEND, NIP = K::NIP, NTAPS = K::NTAPS, NB = K::NB;
    constexpr int NCK = K::NCK, NMB = K::NMB, NFX = K::NFX, PXHS = K::PXHS, PY = K::PY, PX = K::PX;
    constexpr int GSLAB = K::GSLAB, PLANE = K::PLANE, DPW = K::DPW, D = K::D, R = K::R, QN = K::QN, EXB = K::EXB;
    constexpr int ROWB = 2 * PXHS * 64;                      // bytes of one input row (both parities)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const exch = smem + R * PLANE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ck = wave % NCK, mb = (wave / NCK) % NMB, f = wave / (NCK * NMB);
    const int fxi = f % NFX, fyi = f / NFX, gi = f * NMB + mb;
    const int hk = lane >> 5, v = lane & 31;
    int vr, vx;
    lane_voxel_d(v, vr, vx);

    // ---- B-fragment read addresses: lbase[dx][(dy >> 1)] + dy * ROWB + ring slot ---------------------------------------
    // input row py = 2 (4 fyi + vr) + dy -> key row bit = (4 fyi + vr + (dy >> 1)) & 1
    int lbase[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int oy = 4 * fyi + vr, pxh = 8 * fxi + vx + (dx >> 1);
            const int key = ((pxh >> 2) & 1) | (((oy + h) & 1) << 1);
            lbase[dx][h] = (ck >> 1) * GSLAB + ((2 * oy * 2 + (dx & 1)) * PXHS + pxh) * 64 + (((((ck & 1) << 1) | hk) ^ key) << 4);
        }

    // ---- DMA lane constants (item independent) ----------------------------------------------------------------------
    int drel[DPW], dpk[DPW];
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
        const int j = wave * DPW + k;
        const int gg = (j * 1024) / GSLAB;                   // wave-uniform: a DMA instruction never straddles groups
        const int cc = j * 64 + lane - gg * (GSLAB / 16);
        const int pos = cc >> 2, pc = cc & 3;
        const int py = pos / (2 * PXHS), xpar = (pos / PXHS) & 1, pxh = pos % PXHS, px = 2 * pxh + xpar;
        const int key = ((pxh >> 2) & 1) | (((py >> 1) & 1) << 1);
        drel[k] = ((py * p.Wi + px) * p.ld_x + 32 * gg + (pc ^ key) * 8) * 2;
        dpk[k] = (py < PY && px < PX) ? (py | (px << 8)) : -1;
    }
    const unsigned in_plane_bytes = (unsigned)p.Hi * p.Wi * p.ld_x * 2u;
    const unsigned out_plane_bytes = (unsigned)p.Ho * p.Wo * p.ld_y * 2u;

    // ---- partial-sum exchange addresses (as v9) ------------------------------------------------------------------------
    const int rbase = (gi * NCK + ck) * (NCK - 1) * QN * 1024 + lane * 16;
    int wbase[NCK - 1];
#pragma unroll
    for (int jj = 1; jj < NCK; ++jj)
        wbase[jj - 1] = ((gi * NCK + (ck + jj) % NCK) * (NCK - 1) + (NCK - jj - 1)) * QN * 1024 + lane * 16;

    half8 A[NTAPS];
    float biasv[4 * QN];
    int cur_mg = -1;
    floatx16 acc[2];
    half8 b[NB];
    float own[4 * QN];
#pragma unroll
    for (int i = 0; i < 4 * QN; ++i) own[i] = 0.f;
    const floatx16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int xcd = blockIdx.x & 7, cu_slot = blockIdx.x >> 3;
    if constexpr (STATS) {
        // the block zeroes its own partial rows (the item epilogues below accumulate into them; rows of (sample, channel) pairs the
        // block never visits stay zero for the finalize launch): no memset launch in front of the kernel.  Complete before the first
        // barrier of the item loop lets another wave of the block read-modify-write a row.
        const long astride = (long)p.stats_nblk * p.N * p.M;
        float* const rows = p.stats_pws + (long)blockIdx.x * K::NF * p.N * p.M;
        for (int i = threadIdx.x; i < K::NF * p.N * p.M; i += 512) { rows[i] = 0.f; rows[astride + i] = 0.f; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

#pragma unroll 1
    for (int round = 0;; ++round) {
        const int local = round * q.nslots + cu_slot;
        if (local >= q.ipx) break;
        int it = xcd * q.ipx + local;
        if (it >= q.items) break;
        const int fxb = it % q.tiles_x; it /= q.tiles_x;
        const int fyb = it % q.tiles_y; it /= q.tiles_y;
        const int zs = it % q.S; it /= q.S;
        const int n = it % p.N;
        const int mg = it / p.N;
        const int y0 = fyb * K::FY, x0 = fxb * K::FX;                      // output coordinates of the footprint
        const int zs0 = zs * q.L, zs1 = min(zs0 + q.L, p.Ld);              // output planes [zs0, zs1)
        // walked output planes o = zs0 - LEND .. zs1 - 1 (3x3x3: the first only lends its odd plane to zs0); plane step hs
        // covers input plane zin = 2 (zs0 - LEND) + hs
        const int NO = zs1 - zs0 + LEND;

        const int m0 = 32 * (mg * NMB + mb);
        if (mg != cur_mg) {
            cur_mg = mg;
            const int row = ((lane & 31) + 8 * QN * ck) & 31;
#pragma unroll
            for (int tl = 0; tl < NTAPS; ++tl) {
                const half_t* wp = p.wp + lnn_panel_off(tl, m0, 16 * ck, NTAPS, p.KCpad);
                A[tl] = *reinterpret_cast<const half8*>(wp + row * 16 + hk * 8);
            }
#pragma unroll
            for (int i = 0; i < 4 * QN; ++i) biasv[i] = 0.f;
            if (p.bias) {
#pragma unroll
                for (int i = 0; i < 4 * QN; ++i) biasv[i] = p.bias[m0 + 8 * QN * ck + 8 * (i >> 2) + 4 * hk + (i & 3)];
            }
        }

        // ---- per-item lane offsets ----
        float ssum[4 * QN], ssq[4 * QN];
#pragma unroll
        for (int i = 0; i < 4 * QN; ++i) ssum[i] = ssq[i] = 0.f;
        const int iy0 = 2 * y0 - PAD, ix0 = 2 * x0 - PAD;                  // input coordinates of patch position (0, 0)
        int dvoff[DPW];
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int py = dpk[k] & 255, px = (dpk[k] >> 8) & 255;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = dpk[k] >= 0 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            dvoff[k] = ok ? drel[k] + (iy0 * p.Wi + ix0) * p.ld_x * 2 : (int)0x80000000;
        }
        int svoff;
        {
            const int oy = y0 + 4 * fyi + vr, ox = x0 + 8 * fxi + vx;
            const bool ok = oy < p.Lh && ox < p.Lw;
            const int ch = m0 + (NCK == 2 ? 16 * ck + 8 * hk : 8 * ck + 4 * hk);
            svoff = ok ? ((oy * p.Wo + ox) * p.ld_y + ch) * 2 : (int)0x80000000;
        }

        const long in_plane = (long)p.Hi * p.Wi * p.ld_x, out_plane = (long)p.Ho * p.Wo * p.ld_y;
        const int zin0 = 2 * (zs0 - LEND);                                 // input plane of step 0 (may be negative)
        const half_t* din = p.x + ((long)n * p.Di + zin0) * in_plane;
        int dtp = 0, dslot_off = 0;
        const int nsteps_in = 2 * NO;                                      // plane steps that carry data
        auto dma = [&]() {                  // next input plane of this item -> next ring slot
            const int zin = zin0 + dtp;
            // the even plane of the lending output (step 0) contributes nothing: fetch zeros instead of touching memory
            const bool zok = dtp >= LEND && dtp < nsteps_in && zin >= 0 && zin < p.Di;
            const int nrec = zok ? (int)in_plane_bytes : 0;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)din, 0, nrec, 0x00020000);
#pragma unroll
            for (int k = 0; k < DPW; ++k) dma16(rs, smem + dslot_off + (wave * DPW + k) * 1024, dvoff[k]);
            din += in_plane;
            ++dtp;
            dslot_off = dslot_off + PLANE == R * PLANE ? 0 : dslot_off + PLANE;
        };

        // output plane finalised after the odd step of walked output w (z = zs0 - 1 + w): own quads + partial sums -> y
        half_t* optr = p.y + ((long)n * p.Do + (zs0 - LEND - 1)) * out_plane;   // plane of the first call, fin_store(-1); never dereferenced below zs0
        floatx4 pv[(NCK - 1) * QN];
        auto fin_load = [&](int w) {
            const char* eb = exch + (w & 1) * EXB + rbase;
#pragma unroll
            for (int s = 0; s < (NCK - 1) * QN; ++s) pv[s] = *reinterpret_cast<const floatx4*>(eb + s * 1024);
        };
        auto fin_store = [&](int w) {
            const bool ov = w >= LEND && w < NO;
            float fin[4 * QN];
#pragma unroll
            for (int i = 0; i < 4 * QN; ++i) fin[i] = own[i];
#pragma unroll
            for (int s = 0; s < (NCK - 1) * QN; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) fin[(s % QN) * 4 + i] += pv[s][i];
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)optr, 0, ov ? (int)out_plane_bytes : 0, 0x00020000);
            optr += out_plane;
            if constexpr (NCK == 2) {
                const half2v a0 = {(half_t)(fin[0] + biasv[0]), (half_t)(fin[1] + biasv[1])};
                const half2v a1 = {(half_t)(fin[2] + biasv[2]), (half_t)(fin[3] + biasv[3])};
                const half2v b0 = {(half_t)(fin[4] + biasv[4]), (half_t)(fin[5] + biasv[5])};
                const half2v b1 = {(half_t)(fin[6] + biasv[6]), (half_t)(fin[7] + biasv[7])};
                if constexpr (STATS) {
                    const float r[8] = {(float)a0[0], (float)a0[1], (float)a1[0], (float)a1[1],
                                        (float)b0[0], (float)b0[1], (float)b1[0], (float)b1[1]};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float m = ov ? r[i] : 0.f;
                        ssum[i] += m;
                        ssq[i] = __builtin_fmaf(m, m, ssq[i]);
                    }
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, b0), false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a1), __builtin_bit_cast(unsigned, b1), false, false);
                const uint4v o = {s0[0], s1[0], s0[1], s1[1]};
                __builtin_amdgcn_raw_buffer_store_b128(o, rs, svoff, 0, 0);
            } else {
                half4 o4;
#pragma unroll
                for (int i = 0; i < 4; ++i) o4[i] = (half_t)(fin[i] + biasv[i]);
                if constexpr (STATS) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float m = ov ? (float)o4[i] : 0.f;
                        ssum[i] += m;
                        ssq[i] = __builtin_fmaf(m, m, ssq[i]);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v, o4), rs, svoff, 0, 0);
            }
        };

        auto ldb = [&](int slot_off, int i) -> half8 {      // i = dy * EXT + dx (compile time after unrolling)
            const int dy = i / EXT, dx = i % EXT;
            return *reinterpret_cast<const half8*>(smem + slot_off + lbase[dx][dy >> 1] + dy * ROWB);
        };

        // ---- prologue: first D planes in flight, planes 0 and 1 landed ----
#pragma unroll
        for (int tp = 0; tp < D; ++tp) dma();
        wait_vm<(D - 2) * DPW, false>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) b[i] = ldb(0, i);
        int ro = 0;

        // one input plane.  ODD: taps dz = 2 into acc[CUR] (completing it) and dz = 0 into acc[1 - CUR] (starting the next
        // output plane); even: taps dz = 1 into acc[CUR].  w = walked output the plane belongs to.
        auto step = [&](auto ODD_, auto CUR_, int w) {
            constexpr bool ODD = decltype(ODD_)::value != 0;
            constexpr int CUR = decltype(CUR_)::value;
            if constexpr (!ODD) fin_load(w - 1);            // the output completed by the previous (odd) step
            dma();
            const int rn = ro + PLANE == R * PLANE ? 0 : ro + PLANE;
#pragma unroll
            for (int i = 0; i < NIP; ++i) {
                if constexpr (!ODD) { if (i == (NIP > 4 ? 3 : 1)) fin_store(w - 1); }
                const int ii = i + NB - 1;
                b[ii % NB] = ii < NIP ? ldb(ro, ii) : ldb(rn, ii - NIP);
                if constexpr (EXT == 3) {
                    if constexpr (ODD) {
                        acc[CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[18 + i], b[i % NB], acc[CUR], 0, 0, 0);
                        acc[1 - CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], b[i % NB], i == 0 ? zero16 : acc[1 - CUR], 0, 0, 0);
                    } else {
                        acc[CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[9 + i], b[i % NB], acc[CUR], 0, 0, 0);
                    }
                } else {
                    // 2x2x2: even plane = taps dz 0 (starting the output plane), odd plane = taps dz 1 (completing it)
                    acc[CUR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(ODD ? NIP : 0) + i], b[i % NB],
                                                                      (!ODD && i == 0) ? zero16 : acc[CUR], 0, 0, 0);
                }
            }
            ro = rn;
            if constexpr (ODD) {
                // publish the completed accumulator: own quads stay in registers, the rest goes to the waves that finalise them
#pragma unroll
                for (int i = 0; i < 4 * QN; ++i) own[i] = acc[CUR][i];
                char* eb = exch + (w & 1) * EXB;
#pragma unroll
                for (int jj = 1; jj < NCK; ++jj)
#pragma unroll
                    for (int qi = 0; qi < QN; ++qi) {
                        const int r0 = (jj * QN + qi) * 4;
                        const floatx4 wv = {acc[CUR][r0], acc[CUR][r0 + 1], acc[CUR][r0 + 2], acc[CUR][r0 + 3]};
                        *reinterpret_cast<floatx4*>(eb + wbase[jj - 1] + qi * 1024) = wv;
                    }
            }
            wait_vm<(D - 2) * DPW, true>();
            __builtin_amdgcn_s_barrier();
        };

        // walked outputs w = 0 .. NO - 1 (two plane steps each) + one more even step that stores the last output, in rounds
        // of two outputs (static accumulator roles); the padding steps run on zero planes and store nothing
        acc[0] = zero16;                                     // (3x3x3) the lending output's accumulator, never stored
        const int NW = (NO + 1 + 1) / 2 * 2;
#pragma unroll 1
        for (int w = 0; w < NW; w += 2) {
            step(IC<0>{}, IC<0>{}, w);
            step(IC<1>{}, IC<0>{}, w);
            step(IC<0>{}, IC<1>{}, w + 1);
            step(IC<1>{}, IC<1>{}, w + 1);
        }
        if constexpr (STATS) {
            const bool lane_ok = svoff != (int)0x80000000;
            float* const prow0 = p.stats_pws + (((long)(blockIdx.x * K::NF + f)) * p.N + n) * p.M;
            const long astride = (long)p.stats_nblk * p.N * p.M;
#pragma unroll
            for (int i = 0; i < 4 * QN; ++i) {
                float a = lane_ok ? ssum[i] : 0.f, b2 = lane_ok ? ssq[i] : 0.f;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b2 += __shfl_xor(b2, o, 64); }
                if ((lane & 31) == 0) {
                    const int ch = m0 + 8 * QN * ck + 8 * (i >> 2) + 4 * hk + (i & 3);
                    prow0[ch] += a;
                    prow0[astride + ch] += b2;
                }
            }
        }
        wait_vm<0, true>();
        __builtin_amdgcn_s_barrier();
    }
}

template <class K, bool STATS>
int launch_d2(hipStream_t s, ConvParams& p, int num_cu, const char* name) {
    const int tiles_y = lnn_cdiv(p.Lh, K::FY), tiles_x = lnn_cdiv(p.Lw, K::FX);
    const int mgroups = p.M / (32 * K::NMB);
    const long cols = (long)mgroups * p.N * tiles_y * tiles_x;
    int G = num_cu / 8 * 8;
    if (G < 8) G = 8;
    int bestS = 1;
    double best = 1e30;
    for (int S = 1; S <= 16 && S <= p.Ld; ++S) {
        const int L = lnn_cdiv(p.Ld, S);
        if ((long)(S - 1) * L >= p.Ld) continue;                    // empty last segment
        const long items = cols * S;
        const double cost = (double)lnn_cdiv(lnn_cdiv(items, 8), G / 8) * (2 * (L + 1 + K::LEND) + 4);
        if (cost < best * 0.97) { best = cost; bestS = S; }
    }
    D2Launch q;
    q.S = bestS; q.L = lnn_cdiv(p.Ld, bestS); q.tiles_y = tiles_y; q.tiles_x = tiles_x;
    q.items = (int)(cols * bestS);
    q.ipx = lnn_cdiv(q.items, 8);
    q.nslots = G / 8;
    if (q.nslots > q.ipx) q.nslots = q.ipx;
    const int grid = q.nslots * 8;
    static bool attr_set = false;     // per instantiation; idempotent attribute of the code object
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_down2s_kernel<K::NCK, K::NMB, K::NF, K::EXT, STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS);
        attr_set = true;
    }
    if (STATS) {
        p.stats_nblk = grid * K::NF;          // (every launched block zeroes its own rows)
    }
    hipLaunchKernelGGL((igemm_down2s_kernel<K::NCK, K::NMB, K::NF, K::EXT, STATS>), dim3(grid), dim3(512), K::LDS, s, p, q);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}

int d2_num_cu() {
    static int num_cu = 0;        // device property, read once (immutable for the process)
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return lnn_cu_budget(num_cu);
}

}  // namespace

bool lnn_down2s_supported(const ConvParams& p) {
    if (p.accumulate) return false;
    if (!((p.pad_lo == 1 && p.wtaps == 27) || (p.pad_lo == 0 && p.wtaps == 8))) return false;
    if (p.C != 32 && p.C != 64) return false;
    if (p.M % 64 != 0) return false;
    if (p.ld_x % 8 != 0 || p.ld_y % 8 != 0) return false;
    if (p.Di != 2 * p.Do || p.Hi != 2 * p.Ho || p.Wi != 2 * p.Wo) return false;        // even extents (the U-Net's pooling levels)
    if ((double)p.Hi * p.Wi * p.ld_x * 2.0 >= 2147483648.0 || (double)p.Ho * p.Wo * p.ld_y * 2.0 >= 2147483648.0) return false;
    if ((double)(2 * p.Wo + 2) * p.ld_x * 2.0 * (2 * p.Ho + 2) >= 2147483648.0) return false;
    return true;
}

int lnn_down2s_stats_slots(const ConvParams& p) {
    const int nf = p.C == 32 ? 2 : 1;
    return (d2_num_cu() / 8 * 8 < 8 ? 8 : d2_num_cu() / 8 * 8) * nf;
}

int lnn_launch_down2s(hipStream_t s, ConvParams& p, const char* name) {
    const int num_cu = d2_num_cu();
    if (p.wtaps == 8) {                 // transposed-conv data gradient: never feeds an InstanceNorm, no statistics instance
        if (p.C == 32) return launch_d2<D2<2, 2, 2, 2>, false>(s, p, num_cu, name);
        return launch_d2<D2<4, 2, 1, 2>, false>(s, p, num_cu, name);
    }
    if (p.C == 32) {
        if (p.stats_pws) return launch_d2<D2<2, 2, 2>, true>(s, p, num_cu, name);
        return launch_d2<D2<2, 2, 2>, false>(s, p, num_cu, name);
    }
    if (p.stats_pws) return launch_d2<D2<4, 2, 1>, true>(s, p, num_cu, name);
    return launch_d2<D2<4, 2, 1>, false>(s, p, num_cu, name);
}
