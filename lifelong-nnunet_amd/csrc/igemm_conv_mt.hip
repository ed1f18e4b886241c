// Stride-1 3x3x3 implicit-GEMM convolution, macro-tile kernel with in-block split-K ("mt", round 6).
//
// Replaces nn.Conv3d forward / data gradient (test/network_architecture/test_MultiHead_Module.py:346-415; forward order
// generic_ViT_UNet.py:261-286) for the DEEP levels of the U: >= 128 channels on volumes of 10..24 planes (levels 3 / 4 of the
// 160x192x160 plan: 256 -> 256 and 512 -> 256 @ 20x24x20, 320 -> 320 and 640 -> 320 @ 10x12x10).  There the GEMM is
// M = 256..640 output channels x 2 400..19 200 voxels x 27 C contraction: only ~19 (level 3) or ~3 (level 4) 32x32 accumulator
// tiles of OUTPUT per CU.  The tile kernels (igemm_conv_v7 / v8: 8x8x8 voxels x 32 / 64 channels per unit, one or two
// accumulators per wave, 1.0 - 1.5 KB of LDS reads per MFMA, the halo re-staged for every 32 output channels) ran them at
// 500 - 670 TFLOP/s, the flattened-voxel kernel (igemm_gen) level 4 at ~200.  What this kernel does instead:
//   * a block owns 64 output channels x one BAND of voxels: 2 planes x TY rows x TX columns (TX = W, all columns, on every level of
//     the 160x192x160 plan; W / 2, W / 3 ... on the wide planes of anisotropic plans, see mt_geometry); a plane of the band is one
//     SUB-TILE of WN x 32 voxels (flattened (y, x): 8 rows x 20 = 160 = 5 MFMA column tiles at level 3, 12 x 10 = 120 -> 4 at
//     level 4, no padding planes);
//   * its 8 waves are 2 sub-tiles x 4 TAP QUARTERS (taps 7 kq .. 7 kq + 6 of the 27): every wave holds the FULL 64 x (32 WN)
//     sub-tile as 2 x WN accumulators (160 registers at WN = 5) and walks only its share of the contraction -- the in-block
//     split-K that gives a wave 10 MFMAs per 7 fragment reads (0.7 KB of LDS reads per MFMA) although the CU owns only 20
//     accumulator tiles of output.  The four partial sums of a sub-tile meet once, after the last chunk, through LDS (each
//     wave finalises one accumulator quad = 8 channels of every tile; MFMA rows are rotated by 8 kq so that this is always
//     quad 0 -- no runtime register indexing);
//   * the band's halo of ONE 16-channel chunk (4 planes x (TY + 2) x (W + 2) positions x 32 B <= 32 KB) is brought into a
//     double-buffered LDS image by direct-to-LDS buffer loads (out-of-volume positions: zero-filled by the descriptor's range
//     check = the convolution's padding), one barrier per chunk (7 x 10 MFMAs per wave between barriers);
//   * weight fragments (32 rows x 16 channels of one tap = 1 KB, contiguous in the blocked panel) are private to a wave (its
//     taps, its chunk): each wave streams them through its OWN 4-slot LDS ring by direct-to-LDS loads, 3 iterations ahead,
//     ordered by counted s_waitcnt vmcnt only -- no barrier, no staging registers, and every memory operation of the kernel is
//     a DMA, so the counts are exact (cdna_hip_programming.md section 5: mixing load kinds de-pipelines);
//   * levels with fewer band x channel-block items than CUs split the chunk range over `ksplit` blocks that write fp32 partial
//     tiles to the caller's workspace ([part][voxel][Mpad], fixed slice order: deterministic), lnn_launch_splitk_finalize adds the slices.
// LDS image of the halo: [position][32 B], 16-byte half index XOR ((position >> 3) & 1): the 16-lane groups of a
// ds_read_b128 touch 16 consecutive positions up to row wraps -> conflict free for every tap shift (two lanes collide only if
// their positions are congruent mod 16).  The key is applied to the SOURCE half each DMA lane fetches (linear LDS writes).
#include "igemm_common.h"

namespace {

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

template <int N>
__device__ __forceinline__ void mt_wait_vm() {     // literal counts only (see igemm_conv_v9.hip)
    static_assert(N == 0 || N == 4 || N == 6 || N == 8 || N == 10 || N == 12, "extend the table");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}

// 16 bytes per lane, global -> LDS (wave-uniform LDS base + lane * 16); address = descriptor base + voffset (per lane; 0x80000000 =
// out of range -> zeros land) + soffset (wave-uniform).  ONE descriptor per tensor for the whole launch: the per-fragment / per-chunk
// part of the address is the scalar offset -- rebuilding a 64-bit descriptor per DMA was ~40 scalar instructions per iteration.
__device__ __forceinline__ void mt_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_wave_base, 16, voffset, soffset, 0, 0);
}

constexpr int MT_NW = 8;                    // waves per block
constexpr int MT_NH = 4;                    // halo DMA instructions per wave and chunk: 8 x 4 x 32 = 1024 positions
constexpr int MT_HALO = MT_NW * MT_NH * 1024;
// weight-fragment ring: MT_AR slots per wave (one slot = the two 1 KB fragments of an iteration), template parameter of the kernel
constexpr int mt_lds(int AR) { return 2 * MT_HALO + MT_NW * AR * 2048; }       // AR = 4: 128 KB, AR = 6: 160 KB (all of a CU's LDS)
constexpr int MT_NIT = 7;                   // tap iterations per wave and chunk (4 x 7 = 28 >= 27: the last quarter's 7th is a zero tap)

struct MTLaunch {
    int mblk;          // 64-channel output blocks
    int zb, yb, xb;    // bands per sample along z / y / x
    int TY, TX, PY, PX, P; // band rows / columns, halo rows / columns / positions
    int cpp;           // 16-channel chunks per K part
    int nbands;        // N * zb * yb
};

// The fragments of iteration it + 1 are read WHILE the MFMAs of iteration it issue: each column tile's B fragment is re-read for the next
// tap as soon as its two MFMAs are out, the next weight pair goes to a second register pair -- the wave does not depend on its SIMD
// partner to cover the LDS latency.  (The plain read-then-multiply order, 210 instead of 252 registers, measured 5-8 % slower on the
// level-3 layers and 0-4 % on level 4, profiles/r06_kbench_mt_first.txt; removed.)
template <int WN, int MT_AR>
__global__ __launch_bounds__(512, 2) void igemm_conv_mt_kernel(const ConvParams p, const MTLaunch q) {
    static_assert(MT_AR >= 3 && MT_AR <= 7, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const aring = smem + 2 * MT_HALO;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave >> 2, kq = wave & 3;
    const int hk = lane >> 5, v = lane & 31;

    // ---- block -> (channel block, band, K part) -------------------------------------------------------------------
    int b = blockIdx.x;
    const int mb = b % q.mblk; b /= q.mblk;
    const int band = b % q.nbands;
    const int part = b / q.nbands;
    const int xbi = band % q.xb, bzy = band / q.xb;
    const int ybi = bzy % q.yb, zbi = (bzy / q.yb) % q.zb, n = bzy / (q.yb * q.zb);
    const int z0 = zbi * 2, y0 = ybi * q.TY, x0 = xbi * q.TX;
    const int m0 = mb * 64;
    const int PX = q.PX, PYX = q.PY * q.PX;
    const bool flip = p.taps.slot[0] != 0;               // data gradient: tap offset d' uses weight slot 26 - d'

    // ---- halo DMA lane constants -------------------------------------------------------------------------------------
    int hvoff[MT_NH];
#pragma unroll
    for (int k = 0; k < MT_NH; ++k) {
        const int pos = (wave * MT_NH + k) * 32 + (lane >> 1);
        const int pz = pos / PYX, rem = pos - pz * PYX, py = rem / PX, px = rem - py * PX;
        const int iz = z0 - 1 + pz, iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = pos < q.P && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const int half = (lane & 1) ^ ((pos >> 3) & 1);
        hvoff[k] = ok ? (((iz * p.Hi + iy) * p.Wi + ix) * p.ld_x + half * 8) * 2 : (int)0x80000000;
    }
    const long sample_elems = (long)p.Di * p.Hi * p.Wi * p.ld_x;
    // ONE descriptor per input tensor (sample n), built where it is used from loop-invariant scalars; a chunk's channel offset is the
    // scalar offset of its loads
    const half_t* const xn = p.x + (long)n * sample_elems;
    const half_t* const xn2 = (p.x2 ? p.x2 : p.x) + (long)n * sample_elems;
    const int xbytes = (int)(sample_elems * 2);
    int hbuf = 0;                                        // byte offset of the halo buffer the NEXT halo DMA fills
    auto dma_halo = [&](int chunk, bool live) {          // chunk = absolute 16-channel chunk index
        const int c0 = chunk * 16;
        const bool part2 = c0 >= p.csplit;
        const int so = live ? (part2 ? c0 - p.csplit : c0) * 2 : 0;
        const int nrec = live ? xbytes : 0;              // dead chunk: zero records -> every lane out of range (zeros land, no traffic)
        if (part2) {                                     // (wave-uniform branch: never a select between descriptor OBJECTS -- hipcc wraps
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xn2, 0, nrec, 0x00020000);   // such a load in a waterfall loop)
#pragma unroll
            for (int k = 0; k < MT_NH; ++k) mt_dma16(rs, smem + hbuf + (wave * MT_NH + k) * 1024, hvoff[k], so);
        } else {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, nrec, 0x00020000);
#pragma unroll
            for (int k = 0; k < MT_NH; ++k) mt_dma16(rs, smem + hbuf + (wave * MT_NH + k) * 1024, hvoff[k], so);
        }
        hbuf ^= MT_HALO;
    };

    // ---- weight-fragment DMA: per iteration the two row blocks of (tap, chunk) -> ring slot -------------------------
    // lane i fetches row ((i & 31) + 8 kq) & 31, half hk of the fragment: the LDS image is then lane-linear (ds_read_b128 at
    // lane * 16, conflict free) and MFMA row rho holds output channel 32 rb + ((rho + 8 kq) & 31): the channels a wave
    // finalises (quad kq) are always its accumulator quad 0
    const int avoff = ((((lane & 31) + 8 * kq) & 31) * 32 + hk * 16);
    const int kc16 = p.KCpad >> 4;
    const long rb_stride = (long)kc16 * 27 * 512;        // halves between the two row blocks of a (chunk, tap)
    const bool rb1_live = m0 + 32 < p.Mpad;
    int tapslot[MT_NIT];                                 // weight slot of this wave's taps (-1: none)
    int toffb[MT_NIT];                                   // halo byte offset of the tap relative to the centre position
#pragma unroll
    for (int it = 0; it < MT_NIT; ++it) {
        const int tap = kq * MT_NIT + it;
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        tapslot[it] = tap < 27 ? (flip ? 26 - tap : tap) : -1;
        toffb[it] = tap < 27 ? (((dz - 1) * q.PY + (dy - 1)) * PX + (dx - 1)) * 32 : 0;
    }
    // ONE descriptor for the whole panel; fragment (row block, chunk, tap slot) = scalar byte offset (the panel is < 2 GB, checked by
    // lnn_conv_s1_mt_supported); a dead fragment (zero tap, second row block beyond Mpad, past the last chunk) has zero records:
    // every lane is out of range and zeros land
    const int wbytes = (int)((long)(p.Mpad >> 5) * rb_stride * 2);
    const int wbase = (int)((long)(m0 >> 5) * rb_stride * 2), rb_bytes = (int)(rb_stride * 2);
    int aslot = 0;                                       // ring slot (bytes) the NEXT weight DMA fills
    auto dma_a = [&](int it, int chunk, bool live) {     // it compile-time after unrolling
        const bool ok = live && tapslot[it] >= 0;
        const int so = ok ? wbase + (chunk * 27 + tapslot[it]) * 1024 : 0;
        const bool ok1 = ok && rb1_live;
        // the two row blocks' descriptors differ in their record count only (a scalar select on one descriptor word)
        __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, ok ? wbytes : 0, 0x00020000);
        __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, ok1 ? wbytes : 0, 0x00020000);
        char* dst = aring + wave * (MT_AR * 2048) + aslot;
        mt_dma16(r0, dst, avoff, so);
        mt_dma16(r1, dst + 1024, avoff, ok1 ? so + rb_bytes : 0);
        aslot = aslot + 2048 == MT_AR * 2048 ? 0 : aslot + 2048;
    };

    // ---- B-fragment lane addresses: centre position of the lane's voxel in every column tile ------------------------
    const int nv = q.TY * q.TX;                          // voxels of a sub-tile
    int lb[WN];
#pragma unroll
    for (int ct = 0; ct < WN; ++ct) {
        const int vv = ct * 32 + v;
        const int vc = vv < nv ? vv : nv - 1;
        const int yy = vc / q.TX, xx = vc - yy * q.TX;
        lb[ct] = (((sub + 1) * q.PY + (yy + 1)) * PX + (xx + 1)) * 32 + hk * 16;
    }

    floatx16 acc[2][WN];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int ct = 0; ct < WN; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][ct][i] = 0.f;

    const int c_begin = part * q.cpp, nc = q.cpp;
    half8 fa[2][2], fb[WN];
    int rslot = 0;                                       // ring slot (bytes) of the iteration whose fragments are read next
    int rbuf = 0;                                        // halo buffer the current chunk reads
    auto load_a = [&](int set) {                         // set compile-time
        const char* ab = aring + wave * (MT_AR * 2048) + rslot + lane * 16;
        fa[set][0] = *reinterpret_cast<const half8*>(ab);
        fa[set][1] = *reinterpret_cast<const half8*>(ab + 1024);
        rslot = rslot + 2048 == MT_AR * 2048 ? 0 : rslot + 2048;
    };
    auto load_b = [&](int it, int ct) {                  // it, ct compile-time
        const int bb = lb[ct] + rbuf + toffb[it];
        fb[ct] = *reinterpret_cast<const half8*>(smem + (bb ^ ((bb >> 4) & 16)));
    };

    // ---- prologue: halo of the first chunk, the first MT_AR weight iterations -----------------------------------------
    // (weight iterations are numbered i = c * 7 + it over the part's chunks; the first MT_AR are (chunk 0, it 0 .. MT_AR - 1))
    dma_halo(c_begin, true);
#pragma unroll
    for (int i = 0; i < MT_AR; ++i) dma_a(i, c_begin, true);
    mt_wait_vm<2 * (MT_AR - 1)>();                       // halo(0) and A(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    dma_halo(c_begin + 1, nc > 1);

#pragma unroll 1
    for (int c = 0; c < nc; ++c) {
        const int chunk = c_begin + c;
        // here: A(c*7) landed; halo(c) landed for every wave (barrier); in flight, oldest first: A(+1), A(+2), A(+3), halo(c+1)
        load_a(0);
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) load_b(0, ct);
#pragma unroll
        for (int it = 0; it < MT_NIT; ++it) {
            const int cur = it & 1, nxt = (it + 1) & 1;
            if (it + 1 < MT_NIT) {
                // weights of iteration it + 1 (issued MT_AR - 1 refills ago): younger than them are the two later iterations (4) and,
                // for it + 1 <= 3 -- issued in the previous chunk -- the next chunk's halo (4)
                if (it + 1 <= MT_AR - 1) mt_wait_vm<2 * (MT_AR - 2) + MT_NH>();
                else mt_wait_vm<2 * (MT_AR - 2)>();
                load_a(nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < WN; ++ct) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
                    acc[rb][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][rb], fb[ct], acc[rb][ct], 0, 0, 0);
                if (it + 1 < MT_NIT) load_b(it + 1, ct);
                __builtin_amdgcn_sched_barrier(0);
            }
            // refill the slot just consumed with iteration i + MT_AR (= it + 4 of this chunk, or it - 3 of the next one)
            if (it + MT_AR < MT_NIT) dma_a(it + MT_AR, chunk, true);
            else dma_a(it + MT_AR - MT_NIT, chunk + 1, c + 1 < nc);
        }
        if (c + 1 < nc) {
            // next chunk: A((c+1)*7) (issued at it = 3) landed, and with it the older halo(c + 1); every wave is done reading
            // halo(c) once it arrives at the barrier (its fragment reads were consumed by MFMAs it has issued)
            mt_wait_vm<2 * (MT_AR - 1)>();
            __builtin_amdgcn_s_barrier();
            rbuf ^= MT_HALO;
            dma_halo(chunk + 2, c + 2 < nc);
        }
    }
    mt_wait_vm<0>();
    __builtin_amdgcn_s_barrier();                        // LDS is free: exchange of the four tap quarters' partial sums

    // ---- reduction over the tap quarters + store ---------------------------------------------------------------------
    // accumulator quad a of wave kq holds channel quad Q = (a + kq) & 3 (rows rotated): quad 0 is its own, quad a > 0 goes to wave
    // Q's region, source slot 3 - a.  Region of (sub, Q): [3 slots][WN tiles][64 lanes x 16 B].
    const long nvox = (long)p.N * p.Do * p.Ho * p.Wo;
    const long vbase = (long)n * p.Do * p.Ho * p.Wo;
    int ooff[WN];                                        // output voxel of the lane in every column tile (-1: outside the volume); computed
#pragma unroll                                           // here, not before the loop: five registers less across it
    for (int ct = 0; ct < WN; ++ct) {
        const int vv = ct * 32 + v;
        const int yy = vv / q.TX, xx = vv - yy * q.TX;
        const int oz = z0 + sub, oy = y0 + yy, ox = x0 + xx;
        ooff[ct] = (vv < nv && oz < p.Ld && oy < p.Lh && ox < p.Lw) ? ((oz * p.Ho + oy) * p.Wo + ox) : -1;
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        if (rb) __builtin_amdgcn_s_barrier();            // round 0's readers are done
#pragma unroll
        for (int a = 1; a < 4; ++a) {
            const int Q = (a + kq) & 3;
            char* dst = smem + (((sub * 4 + Q) * 3 + (3 - a)) * WN) * 1024 + lane * 16;
#pragma unroll
            for (int ct = 0; ct < WN; ++ct) {
                const floatx4 w = {acc[rb][ct][4 * a], acc[rb][ct][4 * a + 1], acc[rb][ct][4 * a + 2], acc[rb][ct][4 * a + 3]};
                *reinterpret_cast<floatx4*>(dst + ct * 1024) = w;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* src = smem + (((sub * 4 + kq) * 3) * WN) * 1024 + lane * 16;
        const int m = m0 + rb * 32 + 8 * kq + 4 * hk;
        const bool m_ok = m < p.M;
        floatx4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && p.ksplit == 1 && m_ok) bv = *reinterpret_cast<const floatx4*>(p.bias + m);
        const bool part2 = m0 + rb * 32 >= p.msplit;
        half_t* const yt = part2 ? p.y2 - p.msplit : p.y;
#pragma unroll
        for (int ct = 0; ct < WN; ++ct) {
            floatx4 r = {acc[rb][ct][0], acc[rb][ct][1], acc[rb][ct][2], acc[rb][ct][3]};
#pragma unroll
            for (int s = 0; s < 3; ++s) r += *reinterpret_cast<const floatx4*>(src + (s * WN + ct) * 1024);
            if (ooff[ct] < 0 || !m_ok) continue;
            const long vox = vbase + ooff[ct];
            if (p.ksplit > 1) {
                *reinterpret_cast<floatx4*>(p.scratch + ((long)part * nvox + vox) * p.Mpad + m) = r;
                continue;
            }
            r += bv;
            half4* dst = reinterpret_cast<half4*>(yt + vox * p.ld_y + m);
            if (p.accumulate) {
                const half4 old = *dst;
                r[0] += (float)old[0]; r[1] += (float)old[1]; r[2] += (float)old[2]; r[3] += (float)old[3];
            }
            const half4 o = {(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3]};
            *dst = o;
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

int mt_num_cu() {
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return lnn_cu_budget(num_cu);
}

// band geometry for a (H, W) plane: the (WN, TY, TX) with the fewest MFMA columns per plane whose halo fits the LDS image.  A band
// covers TY rows x TX columns; TX = W (all columns) wherever that fills the columns best -- every level of the 160x192x160 plan --
// and W / 2, W / 3, ... on wide planes (80x64, 160x128: the middle levels of anisotropic plans, whose 20 planes are too few for the
// z-streaming kernel).  Ties go to the wider band (less halo per voxel).
bool mt_geometry(const ConvParams& p, int& WN, int& TY, int& TX, double& eff) {
    WN = 0; TY = 0; TX = 0; eff = 0.0;
    for (int nx = 1; nx <= 16; ++nx) {
        const int tx = lnn_cdiv(p.Lw, nx);
        if (nx > 1 && tx == lnn_cdiv(p.Lw, nx - 1)) continue;
        for (int wn = 5; wn >= 4; --wn) {               // ties go to the wider sub-tile (10 MFMAs per 7 fragment reads)
            int ty = (32 * wn) / tx;
            if (ty > p.Lh) ty = p.Lh;
            while (ty >= 1 && 4L * (ty + 2) * (tx + 2) > MT_NW * MT_NH * 32) --ty;
            if (ty < 1) continue;
            const double e = (double)p.Lh * p.Lw / ((double)lnn_cdiv(p.Lh, ty) * lnn_cdiv(p.Lw, tx) * 32 * wn);
            if (e > eff + 1e-9) { eff = e; WN = wn; TY = ty; TX = tx; }
        }
    }
    return WN != 0;
}

template <int WN, int AR>
int launch_mt(hipStream_t s, ConvParams& p, const MTLaunch& q, int grid, const char* name) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_conv_mt_kernel<WN, AR>), hipFuncAttributeMaxDynamicSharedMemorySize, mt_lds(AR));
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_conv_mt_kernel<WN, AR>), dim3(grid), dim3(512), mt_lds(AR), s, p, q);
    LNN_CHECK_LAUNCH(name);
    return LNN_OK;
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

bool lnn_conv_s1_mt_supported(const ConvParams& p) {
    if (p.os != 1 || p.pad_lo != 1 || p.taps.ntaps != 27 || p.wtaps != 27) return false;
    if (p.C % 16 != 0 || p.Mpad % 32 != 0 || p.M % 8 != 0) return false;
    if (p.ld_x % 8 != 0 || p.ld_y % 8 != 0) return false;
    if (p.csplit != 0x7fffffff && p.csplit % 16 != 0) return false;
    if (p.msplit != 0x7fffffff && p.msplit % 32 != 0) return false;
    if (p.Di != p.Do || p.Hi != p.Ho || p.Wi != p.Wo) return false;
    if ((double)p.Di * p.Hi * p.Wi * p.ld_x * 2.0 >= 2147483648.0) return false;        // one descriptor per sample
    if ((double)p.Mpad * p.KCpad * 27 * 2.0 >= 2147483648.0) return false;                  // one descriptor for the weight panel
    int wn, ty, tx; double eff;
    return mt_geometry(p, wn, ty, tx, eff);
}

// fraction of the MFMA columns that carry real voxels (band geometry x plane-pair padding): the automatic selection asks for >= 0.7
double lnn_conv_s1_mt_efficiency(const ConvParams& p) {
    int wn, ty, tx; double eff;
    if (!lnn_conv_s1_mt_supported(p) || !mt_geometry(p, wn, ty, tx, eff)) return 0.0;
    return eff * p.Ld / (2.0 * lnn_cdiv(p.Ld, 2));
}

// K parts this launch would use with a workspace of ws_elems floats (1 = no split)
int lnn_conv_s1_mt_ksplit(const ConvParams& p, const float* ws, long ws_elems) {
    int wn, ty, tx; double eff;
    if (!mt_geometry(p, wn, ty, tx, eff)) return 1;
    const long items = (long)lnn_cdiv(p.Mpad, 64) * p.N * lnn_cdiv(p.Ld, 2) * lnn_cdiv(p.Lh, ty) * lnn_cdiv(p.Lw, tx);
    const int nchunks = p.C / 16;
    const long nvox = (long)p.N * p.Do * p.Ho * p.Wo;
    const int cus = mt_num_cu();
    int best = 1;
    if (!ws) return 1;
    for (int k = 2; k <= 16; ++k)
        if (nchunks % k == 0 && nchunks / k >= 2 && items * k <= cus + cus / 16 && ws_elems >= (long)k * nvox * p.Mpad) best = k;
    return items * 2 <= cus ? best : 1;
}

int lnn_launch_conv_s1_mt(hipStream_t s, ConvParams& p, float* ws, long ws_elems, const char* name) {
    LNN_REQUIRE(lnn_conv_s1_mt_supported(p), "%s: shape not supported by the macro-tile kernel", name);
    int WN, TY, TX; double eff;
    mt_geometry(p, WN, TY, TX, eff);
    MTLaunch q;
    q.mblk = lnn_cdiv(p.Mpad, 64);
    q.zb = lnn_cdiv(p.Ld, 2); q.yb = lnn_cdiv(p.Lh, TY); q.xb = lnn_cdiv(p.Lw, TX);
    q.TY = TY; q.TX = TX; q.PY = TY + 2; q.PX = TX + 2; q.P = 4 * q.PY * q.PX;
    q.nbands = p.N * q.zb * q.yb * q.xb;
    const int ks = lnn_conv_s1_mt_ksplit(p, ws, ws_elems);
    p.ksplit = ks;
    p.scratch = ks > 1 ? ws : nullptr;
    q.cpp = (p.C / 16) / ks;
    const int grid = q.mblk * q.nbands * ks;
    // ring depth 4 (3 iterations ahead); 6 -- all 160 KB of LDS -- measured +1-2 % on level 3, +-0 on level 4
    // (profiles/r06_kbench_mt_descriptor_ring_ab.txt); what bounds the loop: profiles/r06_mt_ablation.txt
    const int rc = WN == 5 ? launch_mt<5, 4>(s, p, q, grid, name) : launch_mt<4, 4>(s, p, q, grid, name);
    if (rc != LNN_OK || ks == 1) return rc;
    return lnn_launch_splitk_finalize(s, p, name);
}
