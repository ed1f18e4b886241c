// Stride-1 3x3x3 implicit-GEMM convolution, z-streaming with register-resident weights (v9).
//
// Replaces nn.Conv3d forward / data gradient (test/network_architecture/test_MultiHead_Module.py:346-415) for the
// layers of the two highest resolutions, where the input has 32 or 64 channels.  Same GEMM mapping as the other
// kernels (v_mfma_f32_32x32x16_f16: rows = 32 output channels, columns = 32 voxels, contraction = 16 input channels
// of one tap); what changed is WHERE the operands live.  v5 / v8 read both MFMA operands from LDS for every MFMA
// (1.0 - 1.5 KB of ds_read_b128 per MFMA: LDS-feed-limited at ~30 % of the matrix peak).  Here:
//   * a wave owns ONE 16-channel chunk of the input channels and ONE 32-row block of output channels and keeps the 27
//     weight fragments of that (block, chunk) in 108 VGPRs for the whole launch: no A-operand traffic at all;
//   * a wave owns a 4 (y) x 8 (x) footprint and walks it along z.  The B fragment of input plane zi at in-plane shift
//     (dy, dx) is read ONCE from LDS and feeds three MFMAs: output planes zi+1, zi, zi-1 with taps dz = 0, 1, 2
//     (three rolling accumulators): 9 ds_read_b128 per 27 MFMAs = 0.33 KB of LDS reads per MFMA;
//   * the channel chunks of one footprint are spread over NCK waves; their partial sums meet in LDS once per output
//     plane (each wave finalises 32 / NCK of the channels: balanced, and the NCK = 2 case ends in 16-byte stores
//     after a v_permlane32_swap);
//   * input planes (footprint + 1 halo, all channel groups) are brought into an LDS ring by direct-to-LDS buffer
//     loads (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass; out-of-volume positions use the
//     buffer descriptor's range check (offset 0x80000000 / num_records = 0 -> zeros land in LDS), and the epilogue
//     stores drop out-of-volume lanes the same way, so the steady-state loop has no branches;
//   * the z halo is free (a column is walked continuously); the in-plane halo is shared between neighbouring blocks
//     through the XCD's L2: blocks are placed so that the 32 CUs of an XCD work on adjacent columns.
// 128 input channels (NCK = 8, round 3): every wave is a chunk, four of them finalise one accumulator quad each (a lane's
// 4 consecutive channels -> 8-byte stores) and receive 7 partial quads per plane; the ring shrinks to 4 slots (3 planes in
// flight) so that 4 x 24 KB of planes + 2 x 28 KB of exchange fit the 160 KB.
// LDS layout of a plane slab: [32-channel group][py][px (row stride PXS)][4 x 16 bytes]; the 16-byte piece index is
// XOR-keyed with ((px >> 2) & 1) | ((py & 1) << 1): with the lane -> voxel map below every ds_read_b128 16-lane
// group reads 16 distinct 16-byte bank slots for all 9 in-plane shifts (conflict free).  The DMA writes LDS linearly
// (wave-uniform base + lane * 16), so the key is applied to the SOURCE piece each lane fetches.
#include "igemm_common.h"

namespace {

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;

template <int V>
struct IC { static constexpr int value = V; };

// s_waitcnt vmcnt(N) [lgkmcnt(0)] with a literal count (an "n" operand that depends on a template parameter makes the
// host pass drop the kernel's instantiation without a diagnostic).  The "memory" clobber is what keeps the compiler from
// moving LDS accesses across the wait / the raw s_barrier that follows it.
template <int N, bool LGKM>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N <= 8, "extend the table");
    if constexpr (LGKM) {
        if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
        if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    } else {
        if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
}

// 16 bytes per lane, global -> LDS (wave-uniform LDS base + lane * 16), range-checked by the buffer descriptor.
// A plain (non-template) device function: inside the kernel template the host pass drops the instantiation silently.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, int voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_wave_base, 16, voffset, 0, 0, 0);
}


__device__ __forceinline__ void lane_voxel9(int v, int& r, int& x) {   // see igemm_conv_tile.hip
    if (v < 4) { r = 0; x = v; }
    else if (v < 12) { r = 2; x = v - 4; }
    else if (v < 16) { r = 0; x = v - 8; }
    else if (v < 20) { r = 3; x = v - 16; }
    else if (v < 28) { r = 1; x = v - 20; }
    else { r = 3; x = v - 24; }
}

// NCK input-channel chunks (C = 16 NCK) x NMB output-channel blocks (32 NMB per item) x NF footprints = 8 waves
template <int NCK_, int NMB_, int NF_>
struct V9 {
    static constexpr int NCK = NCK_, NMB = NMB_, NF = NF_;
    static_assert(NCK * NMB * NF == 8 && (NCK == 2 || NCK == 4 || NCK == 8), "8 waves per block");
    static constexpr int NG = NCK / 2;                       // 32-channel groups of the input
    static constexpr int NFX = NF >= 4 ? 2 : 1, NFY = NF / NFX;
    static constexpr int FY = 4 * NFY, FX = 8 * NFX;         // block footprint
    static constexpr int PY = FY + 2, PX = FX + 2, PXS = (PX + 3) / 4 * 4;
    static constexpr int GRAW = PY * PXS * 64;               // bytes of one group slab
    static constexpr int DPW = (NG * ((GRAW + 1023) / 1024) + 7) / 8;   // DMA instructions per wave per plane
    static constexpr int GSLAB = DPW * 8 / NG * 1024;
    static_assert(GSLAB >= GRAW && (DPW * 8) % NG == 0, "slab padding");
    static constexpr int PLANE = NG * GSLAB;
    static constexpr int D = NCK == 8 ? 3 : 4, R = D + 1;    // planes in flight / ring slots (R >= D + 1)
    static constexpr int NFIN = NCK < 4 ? NCK : 4;           // waves of a (footprint, output block) group that finalise outputs
    static constexpr int QN = 4 / NFIN;                      // accumulator quads a finalising wave owns
    static constexpr int EXB = NF * NMB * NFIN * (NCK - 1) * QN * 1024;   // one partial-sum exchange buffer
    static constexpr int LDS = R * PLANE + 2 * EXB;
    // fused normalisation-backward reduce (EPI = 2, NCK = 2): per plane every wave brings 16 bytes per lane of u
    static constexpr int UD = 1, UW = 1024, UB = 8 * UW;
    static constexpr int LDS_RED = LDS + R * UB + 2 * 32 * NMB * 4;
};

struct V9Launch {
    int items, S, L, tiles_y, tiles_x, nslots, ipx;
};

// EPI = 1 (statistics): the epilogue also accumulates sum / sum of squares of the STORED (fp16-rounded) outputs per (sample,
// channel) -- the InstanceNorm statistics pass of the next op (norm_act.hip:in_stats_kernel) without re-reading the tensor.
// Partials go to p.stats_pws[a][slot][n][c] (a = 0 sum, 1 sum of squares; slot = block * NF + footprint: every slot row is owned
// by one wave, so the accumulation is a plain read-modify-write and the result is deterministic).
// EPI = 2 (data gradient only; round 4): the output is dL/dz of an InstanceNorm + LeakyReLU block whose convolution output u
// (p.red_u) is still in memory, and the epilogue takes pass 1 of that block's backward (norm_act.hip:in_lrelu_bwd_reduce_kernel)
// with it: g = dz * lrelu'(a u + b) (a = gamma rstd, b = beta - a mean; dz = the STORED fp16 value), partials of sum g and
// sum g u in the same slot rows (norm_act.hip:in_lrelu_bwd_sums_kernel with raw_mean turns sum g u into sum g xhat in fp64).  u's
// tile of an output plane rides the input ring: each wave fetches the 16 bytes per lane it will need by ONE direct-to-LDS load next
// to the input plane that is consumed in the step that stores this output plane -- no registers held across steps, the same
// counted waits.  No bias in this mode (a data gradient has none).  Instantiated for 32 input channels only (NCK = 2, where a lane
// stores 8 consecutive channels): +2.5 % on the data gradient against a 0.145 ms reduce pass; the 64-channel variant (8 bytes per
// lane by two 4-byte direct loads, 32 cache lines each) cost the kernel +28 % -- as much as the pass it removes -- and was dropped
// (profiles/r04_fused_in_bwd_reduce.txt).
// KY = 1, GS (round 6): a [1,3,3] convolution (the first stages of anisotropic plans, nnUNetTrainerMultiHead.py:348-369) as the SAME
// column walk with permuted axes: the walk runs along H (the three ky taps are the three rolling accumulators), the footprint spans
// (D, W), and of the nine in-plane shifts only the three of the centre row exist (the kernel has extent 1 along D): 3 fragment reads
// feed 9 MFMAs per plane, 9 resident weight fragments.  GS = the tensor axes come with explicit element strides (p.gs_*) instead of
// the dense (D, H, W) order; p.Di / Hi / Wi (and Do / Ho / Wo, Ld / Lh / Lw) are then the extents of the WALK, footprint-row and
// footprint-column axes.  The dense 3x3x3 instances (KY = 3, GS = false) are compiled from the same text and unchanged.
template <int NCK_, int NMB_, int NF_, int EPI, int KY = 3, bool GS = false>
__global__ __launch_bounds__(512, 2) void igemm_conv_s1_v9_kernel(const ConvParams p, const V9Launch q) {
    using K = V9<NCK_, NMB_, NF_>;
    constexpr bool STATS = EPI == 1, RED = EPI == 2;
    static_assert(KY == 3 || KY == 1, "in-plane rows of the kernel");
    static_assert(!(RED && (GS || KY != 3)), "the fused reduce exists for the dense 3x3x3 data gradient only");
    constexpr int I0 = KY == 3 ? 0 : 3, NI = 3 * KY;          // in-plane shifts i = dy * 3 + dx that exist: [I0, I0 + NI)
    constexpr int NCK = K::NCK, NMB = K::NMB, NFX = K::NFX, PXS = K::PXS, PY = K::PY, PX = K::PX;
    constexpr int GSLAB = K::GSLAB, PLANE = K::PLANE, DPW = K::DPW, D = K::D, R = K::R, QN = K::QN, EXB = K::EXB, NFIN = K::NFIN;
    static_assert(!RED || NCK == 2, "fused normalisation-backward reduce: every wave finalises 8 consecutive channels per lane");
    constexpr int UW = K::UW, UB = K::UB;                     // bytes of u per wave / per ring slot
    constexpr int DPT = DPW + (RED ? K::UD : 0);              // direct-to-LDS loads per wave and plane (the unit of the counted waits)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const exch = smem + R * PLANE;
    char* const uring = exch + 2 * EXB;                       // RED: [R][8 waves][UW]
    float* const abl = reinterpret_cast<float*>(uring + R * UB);   // RED: a[32 NMB], b[32 NMB] of the item's (sample, channel block)

    if constexpr (STATS || RED) {
        // the block zeroes its own partial rows (the item epilogues below accumulate into them; rows of (sample, channel) pairs the
        // block never visits stay zero for the finalize launch): no memset launch in front of the kernel.  Complete before the first
        // barrier of the item loop lets another wave of the block read-modify-write a row.
        const long astride = (long)p.stats_nblk * p.N * p.M;
        float* const rows = p.stats_pws + (long)blockIdx.x * K::NF * p.N * p.M;
        for (int i = threadIdx.x; i < K::NF * p.N * p.M; i += 512) { rows[i] = 0.f; rows[astride + i] = 0.f; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ck = wave % NCK, mb = (wave / NCK) % NMB, f = wave / (NCK * NMB);
    const int fxi = f % NFX, fyi = f / NFX, gi = f * NMB + mb;
    const int hk = lane >> 5, v = lane & 31;
    int vr, vx;
    lane_voxel9(v, vr, vx);

    // ---- B-fragment read addresses: lbase[dx][parity of dy] + dy * PXS * 64 + ring slot ----------------------------
    int lbase[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int py0 = 4 * fyi + vr, px = 8 * fxi + vx + dx;
            const int key = ((px >> 2) & 1) | (((py0 + par) & 1) << 1);
            lbase[dx][par] = (ck >> 1) * GSLAB + (py0 * PXS + px) * 64 + (((((ck & 1) << 1) | hk) ^ key) << 4);
        }

    // ---- DMA lane constants (item independent) ----------------------------------------------------------------------
    int drel[DPW], dpk[DPW];
    const half_t* gten[DPW];
#pragma unroll
    for (int k = 0; k < DPW; ++k) {
        const int j = wave * DPW + k;
        const int gg = (j * 1024) / GSLAB;                   // wave-uniform: a DMA instruction never straddles groups
        const int cc = j * 64 + lane - gg * (GSLAB / 16);
        const int pos = cc >> 2, pc = cc & 3, py = pos / PXS, pxs = pos % PXS;
        const int key = ((pxs >> 2) & 1) | ((py & 1) << 1);
        const int cabs = 32 * gg;
        const bool part2 = cabs >= p.csplit;
        gten[k] = part2 ? p.x2 : p.x;
        if constexpr (GS) drel[k] = ((py - 1) * p.gs_in[1] + (pxs - 1) * p.gs_in[2] + (part2 ? cabs - p.csplit : cabs) + (pc ^ key) * 8) * 2;
        else drel[k] = (((py - 1) * p.Wi + (pxs - 1)) * p.ld_x + (part2 ? cabs - p.csplit : cabs) + (pc ^ key) * 8) * 2;
        // KY == 1: the halo rows of the footprint are never read -> not fetched (zeros land, no traffic)
        dpk[k] = (py < PY && pxs < PX && (KY == 3 || (py >= 1 && py < PY - 1))) ? (py | (pxs << 8)) : -1;
    }
    // GS: a plane's lanes reach across the whole sample (their offsets are checked lane by lane above / below)
    const unsigned in_plane_bytes = GS ? (unsigned)p.gs_nrec_in : (unsigned)p.Hi * p.Wi * p.ld_x * 2u;
    const unsigned out_plane_bytes = GS ? (unsigned)p.gs_nrec_out : (unsigned)p.Ho * p.Wo * p.ld_y * 2u;

    // ---- partial-sum exchange addresses ------------------------------------------------------------------------------
    // Finalising waves (ck < NFIN; all of them unless NCK = 8) rotate their MFMA rows by 8 QN ck, so their own output
    // channels are accumulator quads [0, QN); accumulator quad a of a wave holds channel quad Q = (a + QN ck) & 3 (rotated)
    // or a (not rotated) and goes to finaliser f = Q / QN, slot (ck - f - 1) mod NCK of its (NCK - 1) QN KB region.
    const bool fin = NFIN == NCK || ck < NFIN;
    const int rbase = (gi * NFIN + ck) * (NCK - 1) * QN * 1024 + lane * 16;
    int wb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int Q = fin ? (a + QN * ck) & 3 : a;
        const int f = Q / QN;
        wb[a] = ((gi * NFIN + f) * (NCK - 1) + (ck - f - 1 + NCK) % NCK) * QN * 1024 + (Q % QN) * 1024 + lane * 16;
    }

    half8 A[27];
    float biasv[4 * QN];
    int cur_mg = -1;
    floatx16 acc[3];
    half8 b[3];
    float own[4 * QN];
#pragma unroll
    for (int i = 0; i < 4 * QN; ++i) own[i] = 0.f;
    const floatx16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const int xcd = blockIdx.x & 7, cu_slot = blockIdx.x >> 3;

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
        const int y0 = fyb * K::FY, x0 = fxb * K::FX;
        const int zs0 = zs * q.L, zs1 = min(zs0 + q.L, p.Ld);
        const int T = zs1 - zs0 + 2;

        // ---- weights of this wave's (output block, chunk): 27 fragments, resident until the block changes ----
        const int m0 = 32 * (mg * NMB + mb);
        if (mg != cur_mg) {
            cur_mg = mg;
            // MFMA row rho of a wave with chunk role ck holds output channel m0 + ((rho + 8 QN ck) & 31): the rows a wave
            // finalises are always its accumulator quads [0, QN)
            const int row = ((lane & 31) + (fin ? 8 * QN * ck : 0)) & 31;
#pragma unroll
            for (int tl = 0; tl < 27; ++tl) {
                if (KY == 1 && (tl % 9) / 3 != 1) continue;           // (compile time after unrolling: 9 resident fragments)
                const half_t* wp = p.wp + lnn_panel_off(p.taps.slot[tl], m0, 16 * ck, KY == 3 ? 27 : p.wtaps, p.KCpad);
                A[tl] = *reinterpret_cast<const half8*>(wp + row * 16 + hk * 8);
            }
#pragma unroll
            for (int i = 0; i < 4 * QN; ++i) biasv[i] = 0.f;
            if (!RED && p.bias && fin) {
                // the channels of this lane's own accumulator quads: quad qq, register i -> m0 + 8 QN ck + 8 qq + 4 hk + i
#pragma unroll
                for (int i = 0; i < 4 * QN; ++i) biasv[i] = p.bias[m0 + 8 * QN * ck + 8 * (i >> 2) + 4 * hk + (i & 3)];
            }
        }

        // ---- per-item lane offsets ----
        float ssum[4 * QN], ssq[4 * QN];      // STATS: sum y, sum y^2;  RED: sum g, sum g u (of the lane's 4 QN output channels)
#pragma unroll
        for (int i = 0; i < 4 * QN; ++i) ssum[i] = ssq[i] = 0.f;
        if constexpr (RED) {
            // mask constants of this item's (sample, 32 NMB channels): written after the previous item's closing barrier, read
            // from step 3 on (the prologue's barrier lies between)
            if (tid < 32 * NMB) {
                const int c = 32 * NMB * mg + tid;
                const float a = p.red_gamma[c] * p.red_rstd[n * p.M + c];
                abl[tid] = a;
                abl[32 * NMB + tid] = p.red_beta[c] - a * p.red_mean[n * p.M + c];
            }
        }
        int dvoff[DPW];
#pragma unroll
        for (int k = 0; k < DPW; ++k) {
            const int py = dpk[k] & 255, pxs = (dpk[k] >> 8) & 255;
            const int iy = y0 - 1 + py, ix = x0 - 1 + pxs;
            const bool ok = dpk[k] >= 0 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            if constexpr (GS) dvoff[k] = ok ? drel[k] + (y0 * p.gs_in[1] + x0 * p.gs_in[2]) * 2 : (int)0x80000000;
            else dvoff[k] = ok ? drel[k] + (y0 * p.Wi + x0) * p.ld_x * 2 : (int)0x80000000;
        }
        int svoff, uvoff = 0;
        half_t* yten;
        {
            const int oy = y0 + 4 * fyi + vr, ox = x0 + 8 * fxi + vx;
            const bool ok = oy < p.Lh && ox < p.Lw;
            const bool part2 = m0 >= p.msplit;
            yten = part2 ? p.y2 : p.y;
            const int ch = (part2 ? m0 - p.msplit : m0) + (NCK == 2 ? 16 * ck + 8 * hk : 8 * ck + 4 * hk);
            if constexpr (GS) svoff = ok ? (oy * p.gs_out[1] + ox * p.gs_out[2] + ch) * 2 : (int)0x80000000;
            else svoff = ok ? ((oy * p.Wo + ox) * p.ld_y + ch) * 2 : (int)0x80000000;
            if constexpr (RED) uvoff = ok ? ((oy * p.Wo + ox) * p.red_ld + m0 + 16 * ck + 8 * hk) * 2 : (int)0x80000000;
        }

        // running plane pointers / ring offsets (scalar): no multiplications in the plane loop
        const long in_plane = GS ? (long)p.gs_in[0] : (long)p.Hi * p.Wi * p.ld_x, out_plane = GS ? (long)p.gs_out[0] : (long)p.Ho * p.Wo * p.ld_y;
        const half_t* din[DPW];
#pragma unroll
        for (int k = 0; k < DPW; ++k) din[k] = gten[k] + (GS ? (long)n * p.gs_in_n + (zs0 - 1) * in_plane : ((long)n * p.Di + (zs0 - 1)) * in_plane);
        const int tp_lo = zs0 >= 1 ? 0 : 1;                            // planes tp in [tp_lo, tp_hi) lie inside the volume
        const int tp_hi = min(T, p.Di - (zs0 - 1));
        int dtp = 0, dslot_off = 0, duo = 0;
        // RED: u of the output plane that the step consuming input plane dtp stores (z = zs0 + dtp - 3; real for dtp in [3, T])
        const long u_plane = (long)p.Ho * p.Wo * p.red_ld;
        const half_t* uin = RED ? p.red_u + ((long)n * p.Do + (zs0 - 3)) * u_plane : nullptr;
        auto dma = [&]() {                  // next plane of this item (input z = zs0 - 1 + dtp) -> next ring slot
            const bool zok = dtp >= tp_lo && dtp < tp_hi;
            const int nrec = zok ? (int)in_plane_bytes : 0;
#pragma unroll
            for (int k = 0; k < DPW; ++k) {
                const int j = wave * DPW + k;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)din[k], 0, nrec, 0x00020000);
                dma16(rs, smem + dslot_off + j * 1024, dvoff[k]);
                din[k] += in_plane;
            }
            if constexpr (RED) {
                const int urec = (dtp >= 3 && dtp <= T) ? (int)((unsigned)p.Ho * p.Wo * p.red_ld * 2u) : 0;
                __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)uin, 0, urec, 0x00020000);
                dma16(ru, uring + duo + wave * UW, uvoff);
                uin += u_plane;
                duo = duo + UB == R * UB ? 0 : duo + UB;
            }
            ++dtp;
            dslot_off = dslot_off + PLANE == R * PLANE ? 0 : dslot_off + PLANE;
        };

        // output plane completed at step tprev (z = zs0 + tprev - 2): own quads + the other chunks' partial sums -> y.
        // Split in two so that the LDS reads are in flight while the first MFMAs of the step issue.
        half_t* optr = yten + (GS ? (long)n * p.gs_out_n + (zs0 - 3) * out_plane : ((long)n * p.Do + (zs0 - 3)) * out_plane);      // plane of tprev = -1
        floatx4 pv[(NCK - 1) * QN];
        uint4v ur = {0, 0, 0, 0};            // RED: this lane's channels of u at its voxel of the plane being stored
        int ruo = 0;
        auto fin_load = [&](int tprev) {
            if (!fin) return;
            const char* eb = exch + (tprev & 1) * EXB + rbase;
#pragma unroll
            for (int s = 0; s < (NCK - 1) * QN; ++s) pv[s] = *reinterpret_cast<const floatx4*>(eb + s * 1024);
            if constexpr (RED) ur = *reinterpret_cast<const uint4v*>(uring + ruo + wave * UW + lane * 16);
        };
        // RED: g = dz lrelu'(a u + b) of NE channels (dz, u: packed fp16 pairs, channel order), sums into ssum / ssq
        auto red_acc = [&](auto NE_, const unsigned* dzp, const unsigned* up, int choff) {
            constexpr int NE = decltype(NE_)::value;
            const float* at = abl + choff;
#pragma unroll
            for (int e4 = 0; e4 < NE; e4 += 4) {
                const floatx4 a4 = *reinterpret_cast<const floatx4*>(at + e4);
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(at + 32 * NMB + e4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = e4 + i;
                    const half2v dh = __builtin_bit_cast(half2v, dzp[e >> 1]), uh = __builtin_bit_cast(half2v, up[e >> 1]);
                    const float dzf = (float)dh[e & 1], uf = (float)uh[e & 1];
                    const float pre = __builtin_fmaf(a4[i], uf, b4[i]);
                    const float g = pre > 0.f ? dzf : dzf * p.red_slope;
                    ssum[e] += g;
                    ssq[e] = __builtin_fmaf(g, uf, ssq[e]);
                }
            }
        };
        auto fin_store = [&](int tprev) {
            if (!fin) return;
            const bool ov = tprev >= 2 && tprev < T;
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
                // lane l holds channels {4hk..4hk+3} of both quads, lane l+32 the other four of each: pack to fp16 and
                // exchange halves (vdst = quad 0, src = quad 1) so that every lane ends with 8 consecutive channels
                // (lanes < 32: [own quad 0 | upper's quad 0], lanes >= 32: [lower's quad 1 | own quad 1]) -> ONE 16-byte store
                if constexpr (!RED) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) fin[i] += biasv[i];
                }
                const half2v a0 = {(half_t)fin[0], (half_t)fin[1]};
                const half2v a1 = {(half_t)fin[2], (half_t)fin[3]};
                const half2v b0 = {(half_t)fin[4], (half_t)fin[5]};
                const half2v b1 = {(half_t)fin[6], (half_t)fin[7]};
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
                if constexpr (RED) {
                    if (ov) {                         // (wave-uniform) after the swap a lane holds channels 16 ck + 8 hk + [0, 8) in order
                        const unsigned dzp[4] = {o[0], o[1], o[2], o[3]}, up[4] = {ur[0], ur[1], ur[2], ur[3]};
                        red_acc(IC<8>{}, dzp, up, 32 * mb + 16 * ck + 8 * hk);
                    }
                }
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

        auto ldb = [&](int slot_off, int i) -> half8 {      // i = dy * 3 + dx (compile time after unrolling)
            const int dy = i / 3, dx = i % 3;
            return *reinterpret_cast<const half8*>(smem + slot_off + lbase[dx][dy & 1] + dy * PXS * 64);
        };

        // ---- prologue: first D planes in flight, planes 0 and 1 landed ----
#pragma unroll
        for (int tp = 0; tp < D; ++tp) dma();
        wait_vm<(D - 2) * DPT, false>();
        __builtin_amdgcn_s_barrier();
        b[0] = ldb(0, I0);
        b[1] = ldb(0, I0 + 1);
        int ro = 0;

        auto step = [&](auto U_, int t) {
            constexpr int U = decltype(U_)::value;
            fin_load(t - 1);
            dma();
            const int rn = ro + PLANE == R * PLANE ? 0 : ro + PLANE;
#pragma unroll
            for (int j = 0; j < NI; ++j) {                   // in-plane shift i = I0 + j
                if (j == NI / 3) fin_store(t - 1);
                const int jj = j + 2;
                b[jj % 3] = jj < NI ? ldb(ro, I0 + jj) : ldb(rn, I0 + jj - NI);
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int a = (U + 1 - dz + 3) % 3;
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[dz * 9 + I0 + j], b[j % 3], (j == 0 && dz == 0) ? zero16 : acc[a], 0, 0, 0);
                }
            }
            ro = rn;
            if constexpr (RED) ruo = ruo + UB == R * UB ? 0 : ruo + UB;
            // publish the completed accumulator: own quads stay in registers, the rest goes to the waves that finalise them
            constexpr int c = (U + 2) % 3;
#pragma unroll
            for (int i = 0; i < 4 * QN; ++i) own[i] = acc[c][i];
            char* eb = exch + (t & 1) * EXB;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (a < QN && fin) continue;                 // own quads stay in registers
                const floatx4 w = {acc[c][4 * a], acc[c][4 * a + 1], acc[c][4 * a + 2], acc[c][4 * a + 3]};
                *reinterpret_cast<floatx4*>(eb + wb[a]) = w;
            }
            wait_vm<(D - 2) * DPT, true>();
            __builtin_amdgcn_s_barrier();
        };

        // T plane steps + one more that stores the last output plane, rounded up to the 3-step accumulator rotation
        // (the extra steps run on zero planes and store nothing)
        const int T3 = (T + 1 + 2) / 3 * 3;
#pragma unroll 1
        for (int t = 0; t < T3; t += 3) {
            step(IC<0>{}, t);
            step(IC<1>{}, t + 1);
            step(IC<2>{}, t + 2);
        }
        if constexpr (STATS || RED) {
            // this wave's 32 voxel lanes per half-wave hold the same channels: butterfly over the half, then lanes 0 / 32
            // add the item's sums to the wave's own slot row (out-of-volume voxel columns contribute nothing)
            const bool lane_ok = fin && svoff != (int)0x80000000;
            float* const prow0 = p.stats_pws + (((long)(blockIdx.x * K::NF + f)) * p.N + n) * p.M;
            const long astride = (long)p.stats_nblk * p.N * p.M;
#pragma unroll
            for (int i = 0; i < 4 * QN; ++i) {
                float a = lane_ok ? ssum[i] : 0.f, b2 = lane_ok ? ssq[i] : 0.f;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b2 += __shfl_xor(b2, o, 64); }
                if (fin && (lane & 31) == 0) {
                    // STATS sums the accumulator layout (before the half-wave swap), RED the stored one (after it)
                    const int ch = RED ? m0 + 8 * QN * ck + 4 * QN * hk + i : m0 + 8 * QN * ck + 8 * (i >> 2) + 4 * hk + (i & 3);
                    prow0[ch] += a;
                    prow0[astride + ch] += b2;
                }
            }
        }
        wait_vm<0, true>();
        __builtin_amdgcn_s_barrier();
    }
}

int g_v9_zseg = 0;     // lnn_debug_set_v9_zseg (parity tests): 0 = automatic

template <class K, int EPI, int KY = 3, bool GS = false>
int launch_v9(hipStream_t s, ConvParams& p, int num_cu, const char* name) {
    const int tiles_y = lnn_cdiv(p.Lh, K::FY), tiles_x = lnn_cdiv(p.Lw, K::FX);
    const int mgroups = p.M / (32 * K::NMB);
    const long cols = (long)mgroups * p.N * tiles_y * tiles_x;
    int G = num_cu / 8 * 8;
    if (G < 8) G = 8;
    // z segments: a column is walked continuously (2 extra plane steps per segment), so prefer few segments; split only
    // when the columns alone cannot balance the chip
    int bestS = 1;
    double best = 1e30;
    const int forceS = g_v9_zseg;
    for (int S = 1; S <= 16 && S <= p.Ld; ++S) {
        const int L = lnn_cdiv(p.Ld, S);
        if ((long)(S - 1) * L >= p.Ld) continue;                    // empty last segment
        const long items = cols * S;
        const double cost = (double)lnn_cdiv(lnn_cdiv(items, 8), G / 8) * (L + 2 + 3);
        if (cost < best * 0.97) { best = cost; bestS = S; }
    }
    if (forceS > 0 && forceS <= p.Ld && (long)(forceS - 1) * lnn_cdiv(p.Ld, forceS) < p.Ld) bestS = forceS;
    V9Launch q;
    q.S = bestS; q.L = lnn_cdiv(p.Ld, bestS); q.tiles_y = tiles_y; q.tiles_x = tiles_x;
    q.items = (int)(cols * bestS);
    q.ipx = lnn_cdiv(q.items, 8);
    q.nslots = G / 8;
    if (q.nslots > q.ipx) q.nslots = q.ipx;
    const int grid = q.nslots * 8;
    constexpr int lds = EPI == 2 ? K::LDS_RED : K::LDS;
    static_assert(lds <= 160 * 1024, "LDS of a CU");
    static bool attr_set = false;     // per instantiation; idempotent attribute of the code object
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_conv_s1_v9_kernel<K::NCK, K::NMB, K::NF, EPI, KY, GS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    if (EPI != 0) {
        p.stats_nblk = grid * K::NF;          // (every launched block zeroes its own rows)
    }
    hipLaunchKernelGGL((igemm_conv_s1_v9_kernel<K::NCK, K::NMB, K::NF, EPI, KY, GS>), dim3(grid), dim3(512), lds, s, p, q);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
}

}  // namespace

extern "C" int lnn_debug_set_v9_zseg(int segments) {
    LNN_REQUIRE(segments >= 0 && segments <= 64, "lnn_debug_set_v9_zseg: %d out of range", segments);
    g_v9_zseg = segments;
    return LNN_OK;
}

bool lnn_conv_s1_v9_supported(const ConvParams& p) {
    if (p.accumulate || p.os != 1) return false;
    if (p.C != 32 && p.C != 64 && p.C != 128) return false;
    if (p.M % 32 != 0) return false;
    if (p.ld_x % 8 != 0 || p.ld_y % 8 != 0) return false;
    if (p.csplit != 0x7fffffff && p.csplit % 32 != 0) return false;
    if (p.msplit != 0x7fffffff && p.msplit % 32 != 0) return false;
    if ((double)p.Hi * p.Wi * p.ld_x * 2.0 >= 2147483648.0 || (double)p.Ho * p.Wo * p.ld_y * 2.0 >= 2147483648.0) return false;
    return true;
}

// lnn_debug_set_cu_budget (measurements only): the persistent MFMA kernels (this one, the stride-1 / stride-2 weight gradients, the
// streaming stride-2 conv) size their grids for this many CUs instead of all of them.  0 = all.
static int g_cu_budget = 0;
extern "C" int lnn_debug_set_cu_budget(int cus) {
    LNN_REQUIRE(cus >= 0 && cus <= 4096, "lnn_debug_set_cu_budget: %d out of range", cus);
    g_cu_budget = cus;
    return LNN_OK;
}
int lnn_cu_budget(int device_cus) { return g_cu_budget > 0 && g_cu_budget < device_cus ? g_cu_budget : device_cus; }

static int v9_num_cu() {
    static int num_cu = 0;        // device property, read once (immutable for the process)
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return lnn_cu_budget(num_cu);
}

// slots of the fused-statistics partials (rows of N * M floats per accumulator): grid blocks x footprints per block
int lnn_conv_s1_v9_stats_slots(const ConvParams& p) {
    const int nf = p.C == 128 ? 1 : p.C == 32 ? (p.M % 64 == 0 ? 2 : 4) : (p.M % 64 == 0 ? 1 : 2);
    return (v9_num_cu() / 8 * 8 < 8 ? 8 : v9_num_cu() / 8 * 8) * nf;
}

// the data-gradient shape that feeds an InstanceNorm backward at the highest resolution: 32 -> 32
bool lnn_conv_s1_v9_red_supported(const ConvParams& p) {
    if (!lnn_conv_s1_v9_supported(p) || p.bias || p.msplit != 0x7fffffff) return false;
    if (p.red_ld % 8 != 0 || p.red_ld < p.M || (double)p.Ho * p.Wo * p.red_ld * 2.0 >= 2147483648.0) return false;
    if (lnn_conv_s1_v9_stats_slots(p) > 1024) return false;
    return p.C == 32 && p.M % 64 != 0;
}

int lnn_launch_conv_s1_v9(hipStream_t s, ConvParams& p, const char* name) {
    const int num_cu = v9_num_cu();
    if (p.red_u) {
        LNN_REQUIRE(p.stats_pws && lnn_conv_s1_v9_red_supported(p), "%s: no fused-reduce instance for this shape", name);
        return launch_v9<V9<2, 1, 4>, 2>(s, p, num_cu, name);
    }
    if (p.C == 128) {
        if (p.stats_pws) return launch_v9<V9<8, 1, 1>, 1>(s, p, num_cu, name);
        return launch_v9<V9<8, 1, 1>, 0>(s, p, num_cu, name);
    }
    if (p.stats_pws) {
        // 32 -> 64 (the data-gradient shape of the top decoder conv) never feeds an InstanceNorm: no STATS instance for it
        if (p.C == 32) return launch_v9<V9<2, 1, 4>, 1>(s, p, num_cu, name);
        if (p.M % 64 == 0) return launch_v9<V9<4, 2, 1>, 1>(s, p, num_cu, name);
        return launch_v9<V9<4, 1, 2>, 1>(s, p, num_cu, name);
    }
    if (p.C == 32) {
        if (p.M % 64 == 0) return launch_v9<V9<2, 2, 2>, 0>(s, p, num_cu, name);
        return launch_v9<V9<2, 1, 4>, 0>(s, p, num_cu, name);
    }
    if (p.M % 64 == 0) return launch_v9<V9<4, 2, 1>, 0>(s, p, num_cu, name);
    return launch_v9<V9<4, 1, 2>, 0>(s, p, num_cu, name);
}

// ---- [1,3,3] stride-1 convolutions on the z-streaming kernel with permuted axes (KY = 1, GS instances) ------------------------------
// -1 automatic (LNN_CONV_K133_V9=0 forbids), 0 never, 1 wherever supported (parity tests: lnn_debug_set_k133_v9)
static int g_k133_v9 = -1;
extern "C" int lnn_debug_set_k133_v9(int mode) {
    LNN_REQUIRE(mode >= -1 && mode <= 1, "lnn_debug_set_k133_v9: %d is not one of -1, 0, 1", mode);
    g_k133_v9 = mode;
    return LNN_OK;
}
static int g_k133_last = 0;
extern "C" int lnn_debug_last_k133_on_v9(void) { return g_k133_last; }

// x: gathered tensor (C channels, channel stride ld_x), y: output (M channels, ld_y), both (N, D, H, W, ld) channels-last; wp: panel of
// 9 tap slots [ky][kx]; flip = 0 forward (tap offset (ky, kx) uses slot ky * 3 + kx), 1 data gradient (offset (ky', kx') uses slot
// (2 - ky') * 3 + (2 - kx')).  Returns LNN_OK after launching, -1 when the shape is not covered (the caller runs the generic kernel).
int lnn_conv_k133_on_v9(hipStream_t s, const void* x, int ld_x, const void* wp, const float* bias, void* y, int ld_y, int N, int D, int H,
                        int W, int C, int M, int flip, const char* name) {
    g_k133_last = 0;
    if (g_k133_v9 == 0) return -1;
    if (g_k133_v9 < 0) {
        static int env = -1;
        if (env < 0) { const char* e = getenv("LNN_CONV_K133_V9"); env = (e && e[0] == '0') ? 0 : 1; }
        // the walk runs along H: short walks (and tiny planes) stay on the flattened-voxel kernel
        if (!env || H < 32 || (long)D * W < 256) return -1;
    }
    if (C != 32 && C != 64 && C != 128) return -1;
    if (M % 32 != 0 || ld_x % 8 != 0 || ld_y % 8 != 0 || ld_x < C || ld_y < M) return -1;
    if (!x || !y || !wp || !lnn_aligned16(x) || !lnn_aligned16(y) || !lnn_aligned16(wp)) return -1;
    const double in_bytes = (double)D * H * W * ld_x * 2.0, out_bytes = (double)D * H * W * ld_y * 2.0;
    if (in_bytes >= 2147483648.0 || out_bytes >= 2147483648.0) return -1;
    ConvParams p{};
    p.x = (const half_t*)x; p.wp = (const half_t*)wp; p.bias = bias; p.y = (half_t*)y;
    p.ld_x = ld_x; p.ld_y = ld_y; p.N = N;
    // walk axis = H, footprint rows = D, footprint columns = W
    p.Di = p.Do = p.Ld = H; p.Hi = p.Ho = p.Lh = D; p.Wi = p.Wo = p.Lw = W;
    p.C = C; p.M = M; p.Mpad = lnn_round_up(M, 32); p.KCpad = lnn_round_up(C, 16); p.wtaps = 9;
    p.os = 1; p.pad_lo = 1; p.accumulate = 0; p.dbg = nullptr;
    p.taps.ntaps = 27;
    for (int t = 0; t < 27; ++t) {
        const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;       // the walk's (plane, row, column) shift = (ky, kz, kx) of the conv
        p.taps.pos_off[t] = 0;
        p.taps.slot[t] = (unsigned char)(dy != 1 ? 0 : (flip ? (2 - dz) * 3 + (2 - dx) : dz * 3 + dx));
    }
    p.gs_in[0] = W * ld_x; p.gs_in[1] = H * W * ld_x; p.gs_in[2] = ld_x; p.gs_in_n = (long)D * H * W * ld_x;
    p.gs_out[0] = W * ld_y; p.gs_out[1] = H * W * ld_y; p.gs_out[2] = ld_y; p.gs_out_n = (long)D * H * W * ld_y;
    p.gs_nrec_in = (unsigned)in_bytes; p.gs_nrec_out = (unsigned)out_bytes;
    const int num_cu = v9_num_cu();
    int rc;
    if (C == 128) rc = launch_v9<V9<8, 1, 1>, 0, 1, true>(s, p, num_cu, name);
    else if (C == 32) rc = M % 64 == 0 ? launch_v9<V9<2, 2, 2>, 0, 1, true>(s, p, num_cu, name) : launch_v9<V9<2, 1, 4>, 0, 1, true>(s, p, num_cu, name);
    else rc = M % 64 == 0 ? launch_v9<V9<4, 2, 1>, 0, 1, true>(s, p, num_cu, name) : launch_v9<V9<4, 1, 2>, 0, 1, true>(s, p, num_cu, name);
    if (rc == LNN_OK) g_k133_last = 1;
    return rc;
}
