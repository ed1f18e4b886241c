// Generic-geometry implicit-GEMM convolution family on gfx950 MFMA (v_mfma_f32_32x32x16_f16), channels-last fp16.
//
// What it is for (round 4):
//   (a) plans that are not 3x3x3 / 2x2x2 everywhere: nnU-Net builds the network from `net_conv_kernel_sizes` and
//       `net_num_pool_op_kernel_sizes` (nnUNetTrainerMultiHead.py:348-369 -> upstream Generic_UNet): kernel extent 1 or 3 and
//       stride 1 or 2 PER AXIS (Task005_Prostate: [1,3,3] kernels, [1,2,2] poolings), transposed convolutions with
//       kernel = stride = the pooling of their level;
//   (b) the bottom of the U of the isotropic plans: on 10x12x10 / 5x6x5 volumes the tile kernels (8x8x8 halo tiles x 32
//       channels per block) cannot fill 256 CUs and waste 2.6-3.4x of their MFMAs on tile padding (60-110 us per launch for
//       1-13 us of work).  Here the voxels are FLATTENED (a tile is 32 consecutive loop voxels, whatever
//       the extents) and the contraction (tap x 16-channel chunk) is split over `ksplit` waves.
//
//   OUT[n, so*l + par, m] (+)= bias[m] + sum_{t in taps(class)} sum_c IN[n, si*l + d_t, c] * WP[slot_t][m][c]
//
// with per-axis scales so / si in {1, 2}, output parity classes (one launch covers all of them) and per-class tap lists:
//   conv forward            1 class,  so = 1, si = stride, d = tap - pad
//   conv data gradient      prod(stride) classes, so = stride, si = 1, taps with (par + pad - tap) % stride == 0
//   transposed conv forward prod(stride) classes of ONE tap, so = stride, si = 1
//   transposed conv dgrad   1 class,  so = 1, si = stride, d = tap
//
// Kernel: no LDS, no barriers.  A wave owns RB x VB accumulators (32 output channels x 32 voxels each) and walks its slice of
// the (tap, chunk) steps; both MFMA operands come straight from memory as 16-byte-per-lane BUFFER loads, D steps ahead of use
// (register ring): the A fragment of a step is one contiguous 1 KB run of the blocked weight panel, the B fragment a gather of
// 32 voxels x 32 B; out-of-volume taps, ragged tiles, padded steps and the over-run of the prefetch ring all resolve to
// out-of-range offsets, which the buffer descriptor answers with zeros (= the convolution's zero padding) -- there is not one
// branch or predicated load in the loop.  Split-K partial sums go to an fp32 scratch and are folded by lnn_launch_splitk_finalize.
// L1 traffic is 1 KB per MFMA (2 x 2 tile), i.e. the kernel is bound by the CU's load path at ~0.25 of the MFMA peak: right
// for volumes that are latency-bound anyway and for geometry the specialised kernels do not cover, wrong for the big layers.
#include "igemm_common.h"
#include "igemm_gen.h"
#include <algorithm>

namespace {

typedef unsigned uint4v __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;          // every tensor a descriptor covers is < 2 GB (checked by the host)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gen_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ half8 gen_load16(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    const uint4v r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    return __builtin_bit_cast(half8, r);
}

template <int RB, int VB, int D>
__global__ __launch_bounds__(256) void igemm_gen_kernel(const GenParams p) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int v = lane & 31, hk = lane >> 5;
    // ---- block -> (combo = class x row group x contraction part, quad of voxel groups); combos go round-robin over the 8
    // XCDs (consecutive workgroups land on consecutive XCDs), so every block that streams one weight slice shares one L2
    // (only when the combos divide evenly over the XCDs; otherwise combo = block % W: a single combo -- a big volume with few
    // output channels -- must spread over the whole chip, not sit on one XCD)
    const int W = p.nclass * p.mgroups * p.ksplit;
    int w, vq;
    if (p.xcd_order) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        w = xcd + 8 * (j / p.nbpc); vq = j % p.nbpc;
    } else {
        w = blockIdx.x % W; vq = blockIdx.x / W;
    }
    const int g = vq * 4 + wave;
    if (w >= W || g >= p.vgroups) return;
    const int part = w % p.ksplit, mg = (w / p.ksplit) % p.mgroups, cls = w / (p.ksplit * p.mgroups);
    const int tap0 = (int)p.cls_first[cls], ntap = (int)p.cls_first[cls + 1] - tap0;
    const int nck = p.KCpad >> 4;
    const int S = ntap * nck;
    const int s0 = (int)((long)part * S / p.ksplit), s1 = (int)((long)(part + 1) * S / p.ksplit);
    const unsigned cpar = p.cls_par[cls];
    const int parz = cpar & 255, pary = (cpar >> 8) & 255, parx = (cpar >> 16) & 255;

    // ---- this lane's voxel of every tile (decoded once)
    const long total = (long)p.N * p.Ld * p.Lh * p.Lw;
    int vn[VB], vz[VB], vy[VB], vx[VB];
    bool vok[VB];
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        long q = ((long)g * VB + b) * 32 + v;
        vok[b] = q < total;
        if (!vok[b]) q = 0;
        const int lx = (int)(q % p.Lw); q /= p.Lw;
        const int ly = (int)(q % p.Lh); q /= p.Lh;
        const int lz = (int)(q % p.Ld);
        vn[b] = (int)(q / p.Ld); vz[b] = lz; vy[b] = ly; vx[b] = lx;
    }
    const unsigned rowb = (unsigned)p.ld_x * 2u;
    unsigned voff[VB];
    auto set_tap = [&](int ti) {               // ti: wave-uniform tap index inside the class (>= ntap: a padded step)
        const GenTap t = p.taps[tap0 + (ti < ntap ? ti : 0)];
        const int tdz = (int)(t & 255) - 8, tdy = (int)((t >> 8) & 255) - 8, tdx = (int)((t >> 16) & 255) - 8;
#pragma unroll
        for (int b = 0; b < VB; ++b) {
            const int iz = vz[b] * p.siz + tdz, iy = vy[b] * p.siy + tdy, ix = vx[b] * p.six + tdx;
            const bool ok = vok[b] && ti < ntap && (unsigned)iz < (unsigned)p.Di && (unsigned)iy < (unsigned)p.Hi &&
                            (unsigned)ix < (unsigned)p.Wi;
            voff[b] = ok ? ((((unsigned)vn[b] * p.Di + iz) * p.Hi + iy) * p.Wi + ix) * rowb + hk * 16u : OOB;
        }
        return (int)(t >> 24);
    };
    const __amdgpu_buffer_rsrc_t rs_w = gen_rsrc(p.wp, p.wp_bytes);
    unsigned abase[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int mb = mg * RB + r;
        abase[r] = mb < (p.Mpad >> 5) ? (unsigned)mb * nck * p.wtaps * 1024u + v * 32u + hk * 16u : OOB;
    }

    // ---- issue side of the software pipeline
    int si = s0, ti = s0 / nck, ck = s0 % nck;
    int slot = set_tap(ti);
    const __amdgpu_buffer_rsrc_t rs_x = gen_rsrc(p.x, p.x_bytes);
    const int climit = p.C - hk * 8;                                         // first channel this lane's 8-channel half may not read
    auto issue = [&](half8 (&a)[RB], half8 (&b)[VB]) {
        const int c0 = ck * 16;
        // padded step (uniform) / channel half beyond C (last chunk of a ragged C): bit 31 -> the descriptor answers zeros
        const unsigned dead = (si < s1 ? 0u : OOB) | (c0 >= climit ? OOB : 0u);
#pragma unroll
        for (int bb = 0; bb < VB; ++bb) b[bb] = gen_load16(rs_x, (voff[bb] | dead) + (unsigned)c0 * 2u);
        const unsigned wo = ((unsigned)ck * p.wtaps + slot) * 1024u;
#pragma unroll
        for (int r = 0; r < RB; ++r) a[r] = gen_load16(rs_w, abase[r] + wo);
        ++si; ++ck;
        if (ck == nck) { ck = 0; ++ti; slot = set_tap(ti); }
    };

    floatx16 acc[RB][VB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int b = 0; b < VB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][b][i] = 0.f;

    half8 fa[D][RB], fb[D][VB];
#pragma unroll
    for (int d = 0; d < D; ++d) issue(fa[d], fb[d]);
    const int nsteps = (s1 - s0 + D - 1) / D * D;        // padded steps multiply zeros
#pragma unroll 1
    for (int s = 0; s < nsteps; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int b = 0; b < VB; ++b)
                    acc[r][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[d][r], fb[d][b], acc[r][b], 0, 0, 0);
            issue(fa[d], fb[d]);                          // (the last D issues run past the slice: zero-filled, never used)
        }
    }

    // ---- epilogue: lane holds voxel v of tile b x channels {8 q + 4 hk + 0..3} of row block r per accumulator quad
    const long nvox = (long)p.N * p.Do * p.Ho * p.Wo;
#pragma unroll
    for (int b = 0; b < VB; ++b) {
        const int oz = vz[b] * p.soz + parz, oy = vy[b] * p.soy + pary, ox = vx[b] * p.sox + parx;
        if (!vok[b] || oz >= p.Do || oy >= p.Ho || ox >= p.Wo) continue;
        const long vox = (((long)vn[b] * p.Do + oz) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int m0 = (mg * RB + r) * 32;
            if (p.ksplit > 1) {
                float* srow = p.scratch + ((long)part * nvox + vox) * p.Mpad;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int m = m0 + qq * 8 + hk * 4;
                    if (m >= p.Mpad) continue;
                    const floatx4 w4 = {acc[r][b][qq * 4 + 0], acc[r][b][qq * 4 + 1], acc[r][b][qq * 4 + 2], acc[r][b][qq * 4 + 3]};
                    *reinterpret_cast<floatx4*>(srow + m) = w4;
                }
                continue;
            }
            half_t* yrow = p.y + vox * p.ld_y;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int m = m0 + qq * 8 + hk * 4;
                if (m >= p.M) continue;
                float r0 = acc[r][b][qq * 4 + 0], r1 = acc[r][b][qq * 4 + 1], r2 = acc[r][b][qq * 4 + 2], r3 = acc[r][b][qq * 4 + 3];
                if (p.bias) {
                    const floatx4 bv = *reinterpret_cast<const floatx4*>(p.bias + m);
                    r0 += bv[0]; r1 += bv[1]; r2 += bv[2]; r3 += bv[3];
                }
                half4* dst = reinterpret_cast<half4*>(yrow + m);
                if (p.accumulate) {
                    const half4 old = *dst;
                    r0 += (float)old[0]; r1 += (float)old[1]; r2 += (float)old[2]; r3 += (float)old[3];
                }
                const half4 o = {(half_t)r0, (half_t)r1, (half_t)r2, (half_t)r3};
                *dst = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient, generic geometry:  DWP[slot_t][m][c] += sum_{n,l} P[n, l, m] * Q[n, si*l + d_t, c]
//   conv3d: P = dy (at output voxel l), Q = x gathered;   transposed conv: P = x (at input voxel l), Q = dy gathered.
// The contraction runs over voxels, so both operands are transposed relative to channels-last storage: a [16 voxels][32
// channels] image of each operand is laid into LDS by ONE wave-wide LDS-DMA (buffer_load ... lds, 16 B per lane: lane l brings
// channels 8 (l & 3) .. +7 of voxel l >> 2; the gather, the zero padding and the ragged ends are per-lane offsets / the
// descriptor's zero fill) and read back with ds_read_b64_tr_b16 (lane mapping pinned by tests/test_kernels_gpu.py::
// test_tr16_lane_mapping).  A wave owns one 32 x 32 (m, c) panel tile for up to TPW taps over its share of the voxels; waves are
// independent (private LDS images, no barriers), the latency of a 16-voxel group is hidden by the other waves of the CU.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void gen_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_wave_base, unsigned voffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds_wave_base, 16, (int)voffset, 0, 0, 0);
}
__device__ __forceinline__ half4 gen_tr16(const char* addr) {
    fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(addr));
    half4 o;
    __builtin_memcpy(&o, &r, 8);
    return o;
}
__device__ __forceinline__ half8 gen_tr_operand(const char* img, int p_addr) {
    const half4 lo = gen_tr16(img + p_addr), hi = gen_tr16(img + 256 + p_addr);
    half8 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = lo[i]; o[4 + i] = hi[i]; }
    return o;
}
// q / d for 0 <= q < 2^24 (checked by the host) with the reciprocal in float and one correction step either way
__device__ __forceinline__ int gen_div(int q, int d, float rd) {
    int r = (int)((float)q * rd);
    int rem = q - r * d;
    if (rem < 0) { --r; rem += d; }
    if (rem >= d) ++r;
    return r;
}

constexpr int GW_IMG = 1024;                    // one [16 voxels][64 B] image

// TPW = taps (accumulators of 16 VGPRs) per wave: 7 for 27 taps (4 groups), 3 for 9 / 3, 4 for 8 / 4, 2, 1.  A tap slot beyond
// the list (27 = 7 + 7 + 7 + 6) runs on a zero image: the loop has no branches.
template <int TPW>
__global__ __launch_bounds__(256) void igemm_gen_wgrad_kernel(const GenWParams p) {
    constexpr int WAVE_LDS = 2 * (1 + TPW) * GW_IMG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const lds = smem + wave * WAVE_LDS;
    // unit = (panel tile, tap group, voxel part)
    const int panels = (p.Mpad >> 5) * (p.Cpad >> 5);
    const long units = (long)panels * p.tgroups * p.vparts;
    const long unit = (long)blockIdx.x * 4 + wave;
    if (unit >= units) return;
    const int vpart = (int)(unit % p.vparts);
    const int tg = (int)((unit / p.vparts) % p.tgroups);
    const int panel = (int)(unit / ((long)p.vparts * p.tgroups));
    const int m0 = (panel / (p.Cpad >> 5)) * 32, c0 = (panel % (p.Cpad >> 5)) * 32;
    const int t_begin = tg * TPW;
    const __amdgpu_buffer_rsrc_t rs_q = gen_rsrc(p.q, p.q_bytes);
    const __amdgpu_buffer_rsrc_t rs_p = gen_rsrc(p.p, p.p_bytes);

    const int dv = lane >> 2, dc = (lane & 3) * 8;            // DMA role: voxel of the group, first channel of the 16-byte piece
    const unsigned pcol = (m0 + dc < p.M) ? (unsigned)(m0 + dc) * 2u : OOB;
    const unsigned qcol = (c0 + dc < p.C) ? (unsigned)(c0 + dc) * 2u : OOB;
    const int hk = lane >> 5, cb = 16 * ((lane >> 4) & 1), sj = (lane & 15) >> 2, sq = lane & 3;
    const int p_addr = (8 * hk + sj) * 64 + (cb + 4 * sq) * 2;   // transposing read: lane -> channel lane & 31, voxels 8 hk .. +7
    // taps of this wave (uniform): offsets, slot; a slot beyond the list reads zeros
    int tdz[TPW], tdy[TPW], tdx[TPW];
    unsigned tdead[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        const bool has = t_begin + k < p.ntaps;
        const GenTap t = p.taps[has ? t_begin + k : 0];
        tdz[k] = (int)(t & 255) - 8; tdy[k] = (int)((t >> 8) & 255) - 8; tdx[k] = (int)((t >> 16) & 255) - 8;
        tdead[k] = has ? 0u : OOB;
    }

    floatx16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;

    const int total = p.N * p.Ld * p.Lh * p.Lw;
    const int groups = (total + 15) >> 4;
    const int g_begin = (int)((long)vpart * groups / p.vparts), g_end = (int)((long)(vpart + 1) * groups / p.vparts);
    const float rLw = 1.0f / (float)p.Lw, rLh = 1.0f / (float)p.Lh, rLd = 1.0f / (float)p.Ld;
    const unsigned prowb = (unsigned)p.ld_p * 2u, qrowb = (unsigned)p.ld_q * 2u;
#pragma unroll 1
    for (int gi = g_begin; gi < g_end; ++gi) {
        char* const buf = lds + (gi & 1) * (1 + TPW) * GW_IMG;
        // this lane's voxel of the group
        const int q = gi * 16 + dv;
        const unsigned vdead = q < total ? 0u : OOB;
        const int qq = q < total ? q : 0;
        const int t1 = gen_div(qq, p.Lw, rLw), lx = qq - t1 * p.Lw;
        const int t2 = gen_div(t1, p.Lh, rLh), ly = t1 - t2 * p.Lh;
        const int n = gen_div(t2, p.Ld, rLd), lz = t2 - n * p.Ld;
        gen_dma16(rs_p, buf, (((unsigned)qq * prowb) | vdead | (pcol & OOB)) + (pcol & ~OOB));
        const int bz = lz * p.siz, by = ly * p.siy, bx = lx * p.six;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            const int iz = bz + tdz[k], iy = by + tdy[k], ix = bx + tdx[k];
            const bool in = (unsigned)iz < (unsigned)p.Qd && (unsigned)iy < (unsigned)p.Qh && (unsigned)ix < (unsigned)p.Qw;
            const unsigned qrow = ((((unsigned)n * p.Qd + iz) * p.Qh + iy) * p.Qw + ix) * qrowb;
            gen_dma16(rs_q, buf + (1 + k) * GW_IMG, ((in ? qrow : OOB) | vdead | tdead[k] | (qcol & OOB)) + (qcol & ~OOB));
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): the images of this group are in LDS (one wave: no barrier)
        __asm__ volatile("" ::: "memory");           // (the LDS reads below stay below)
        const half8 a = gen_tr_operand(buf, p_addr);
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
            const half8 b = gen_tr_operand(buf + (1 + k) * GW_IMG, p_addr);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
        }
    }
    // ---- epilogue: acc[k][i] = dW[slot][m0 + (i / 4) * 8 + hk * 4 + i % 4][c0 + (lane & 31)]: c is the fastest index -> coalesced
    const int c = c0 + (lane & 31);
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
        if (t_begin + k >= p.ntaps) continue;
        const int slot = (int)(p.taps[t_begin + k] >> 24);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + (i >> 2) * 8 + hk * 4 + (i & 3);
            const long idx = ((long)slot * p.Mpad + m) * p.Cpad + c;
            if (p.parts) p.parts[(long)vpart * p.part_stride + idx] = acc[k][i];
            else atomicAdd(p.dwp + idx, acc[k][i]);
        }
    }
}

__global__ __launch_bounds__(256) void gen_reduce_parts_kernel(const float* __restrict__ parts, int nslots, long slot_elems,
                                                               float* __restrict__ panel) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < slot_elems; i += (long)gridDim.x * 256 * 4) {
        floatx4 a = *reinterpret_cast<const floatx4*>(parts + i);
        for (int k = 1; k < nslots; ++k) a += *reinterpret_cast<const floatx4*>(parts + (long)k * slot_elems + i);
        floatx4* o = reinterpret_cast<floatx4*>(panel + i);
        *o = *o + a;
    }
}

int gen_num_cu() {
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    return num_cu;
}

}  // namespace

// ---- host: tap tables ---------------------------------------------------------------------------------------------------
// kind: 0 conv forward, 1 conv data gradient, 2 transposed-conv forward, 3 transposed-conv data gradient.
// k[a] = kernel extent (transposed conv: = stride), st[a] = stride, pad = k / 2 (transposed conv: 0).
int lnn_gen_geometry(GenParams& p, int kind, const int k[3], const int st[3]) {
    const int kz = k[0], ky = k[1], kx = k[2];
    int nt = 0;
    p.nclass = 0;
    auto add_class = [&](int pz, int py, int px) {
        p.cls_first[p.nclass] = (unsigned)nt;
        p.cls_par[p.nclass] = (unsigned)pz | ((unsigned)py << 8) | ((unsigned)px << 16);
        ++p.nclass;
    };
    auto add_tap = [&](int dz, int dy, int dx, int slot) {
        p.taps[nt++] = gen_tap(dz, dy, dx, slot);
    };
    if (kind == 0 || kind == 3) {
        p.soz = p.soy = p.sox = 1; p.siz = st[0]; p.siy = st[1]; p.six = st[2];
        add_class(0, 0, 0);
        for (int a = 0; a < kz; ++a)
            for (int b = 0; b < ky; ++b)
                for (int c = 0; c < kx; ++c) {
                    const int pz = kind == 0 ? kz / 2 : 0, py = kind == 0 ? ky / 2 : 0, px = kind == 0 ? kx / 2 : 0;
                    add_tap(a - pz, b - py, c - px, (a * ky + b) * kx + c);
                }
    } else {
        p.soz = st[0]; p.soy = st[1]; p.sox = st[2]; p.siz = p.siy = p.six = 1;
        for (int pz = 0; pz < st[0]; ++pz)
            for (int py = 0; py < st[1]; ++py)
                for (int px = 0; px < st[2]; ++px) {
                    add_class(pz, py, px);
                    for (int a = 0; a < kz; ++a)
                        for (int b = 0; b < ky; ++b)
                            for (int c = 0; c < kx; ++c) {
                                if (kind == 2) {          // y[st*l + d] = x[l] W[d]: the class IS the tap
                                    if (a == pz && b == py && c == px) add_tap(0, 0, 0, (a * ky + b) * kx + c);
                                    continue;
                                }
                                // dx[st*l + par] += W[d]^T dy[l + (par + pad - d) / st] for the taps whose numerator divides
                                const int nz = pz + kz / 2 - a, ny = py + ky / 2 - b, nx = px + kx / 2 - c;
                                if (nz % st[0] || ny % st[1] || nx % st[2]) continue;
                                add_tap(nz / st[0], ny / st[1], nx / st[2], (a * ky + b) * kx + c);
                            }
                }
    }
    p.cls_first[p.nclass] = (unsigned)nt;
    p.wtaps = kz * ky * kx;
    return nt;
}

int lnn_launch_gen(hipStream_t s, GenParams& p, float* ws, long ws_elems, const char* name) {
    const long xb = (long)p.N * p.Di * p.Hi * p.Wi * p.ld_x * 2;
    LNN_REQUIRE(xb < 0x7fffffffL, "%s: gathered tensor of %ld bytes exceeds the 2 GB a buffer descriptor of the generic kernel covers", name, xb);
    p.x_bytes = (unsigned)xb;
    p.KCpad = lnn_round_up(p.C, 16); p.Mpad = lnn_round_up(p.M, 32);
    const long wb = (long)(p.Mpad >> 5) * (p.KCpad >> 4) * p.wtaps * 1024;
    LNN_REQUIRE(wb < 0x7fffffffL, "%s: weight panel too large", name);
    p.wp_bytes = (unsigned)wb;
    constexpr int RB = 2, VB = 2, D = 4;
    const long total = (long)p.N * p.Ld * p.Lh * p.Lw;
    p.vgroups = (int)((total + 32 * VB - 1) / (32 * VB));
    p.mgroups = lnn_cdiv(p.Mpad >> 5, RB);
    // split the contraction until ~8 waves per CU are in flight (each part keeps >= 4 steps of the shortest class)
    int smin = 1 << 30;
    for (int c = 0; c < p.nclass; ++c) smin = std::min(smin, (int)(p.cls_first[c + 1] - p.cls_first[c]) * (p.KCpad >> 4));
    const long waves0 = (long)p.nclass * p.mgroups * p.vgroups;
    const long nvox = (long)p.N * p.Do * p.Ho * p.Wo;
    int ks = 1;
    const long want = 8L * gen_num_cu();
    while (ws && waves0 * ks * 2 <= want && smin / (ks * 2) >= 4 && (long)(ks * 2) * nvox * p.Mpad <= ws_elems && ks < 64) ks *= 2;
    p.ksplit = ks; p.scratch = ks > 1 ? ws : nullptr;
    p.nbpc = lnn_cdiv(p.vgroups, 4);
    const int W = p.nclass * p.mgroups * p.ksplit;
    p.xcd_order = W % 8 == 0 ? 1 : 0;
    const unsigned grid = (unsigned)((long)W * p.nbpc);
    hipLaunchKernelGGL((igemm_gen_kernel<RB, VB, D>), dim3(grid), dim3(256), 0, s, p);
    LNN_CHECK_LAUNCH(name);
    if (ks > 1) {
        ConvParams c{};
        c.scratch = ws; c.ksplit = ks; c.Mpad = p.Mpad; c.M = p.M; c.bias = p.bias; c.y = p.y;
        c.ld_y = p.ld_y; c.accumulate = p.accumulate; c.N = p.N; c.Do = p.Do; c.Ho = p.Ho; c.Wo = p.Wo;
        return lnn_launch_splitk_finalize(s, c, name);
    }
    return LNN_OK;
}

template <int TPW>
static int gen_wgrad_launch_t(hipStream_t s, GenWParams& p, const char* name) {
    constexpr int WAVE_LDS = 2 * (1 + TPW) * GW_IMG;
    const long total = (long)p.N * p.Ld * p.Lh * p.Lw;
    p.tgroups = lnn_cdiv(p.ntaps, TPW);
    const long base = (long)(p.Mpad >> 5) * (p.Cpad >> 5) * p.tgroups;
    const int groups = (int)((total + 15) / 16);
    long vp = lnn_cdiv(8L * gen_num_cu(), base);        // ~8 waves per CU
    if (vp > groups) vp = groups;
    if (vp < 1) vp = 1;
    const long slot_elems = (long)p.wtaps * p.Mpad * p.Cpad;
    if (p.parts) {
        while (vp > 1 && vp * slot_elems > p.parts_elems) --vp;
        LNN_REQUIRE(vp * slot_elems <= p.parts_elems, "%s: deterministic scratch too small (%ld floats needed, %ld given)", name,
                    vp * slot_elems, p.parts_elems);
        p.part_stride = slot_elems;       // (every element of a part's copy is written: the panel tiles x tap groups cover it)
    }
    p.vparts = (int)vp;
    const long units = base * vp;
    auto kern = igemm_gen_wgrad_kernel<TPW>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)lnn_cdiv(units, 4)), dim3(256), (size_t)4 * WAVE_LDS, s, p);
    LNN_CHECK_LAUNCH(name);
    if (p.parts) {
        const long v4 = slot_elems / 4;
        const int blocks = (int)((v4 + 255) / 256 < 2048 ? (v4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(gen_reduce_parts_kernel, dim3(blocks), dim3(256), 0, s, p.parts, (int)vp, slot_elems, p.dwp);
        LNN_CHECK_LAUNCH(name);
    }
    return LNN_OK;
}

int lnn_launch_gen_wgrad(hipStream_t s, GenWParams& p, const char* name) {
    const long total = (long)p.N * p.Ld * p.Lh * p.Lw;
    LNN_REQUIRE(total < (1L << 24), "%s: %ld loop voxels exceed the 2^24 the generic weight gradient indexes", name, total);
    const long pb = total * p.ld_p * 2, qb = (long)p.N * p.Qd * p.Qh * p.Qw * p.ld_q * 2;
    LNN_REQUIRE(pb < 0x7fffffffL && qb < 0x7fffffffL, "%s: tensor exceeds the 2 GB a buffer descriptor of the generic kernel covers", name);
    p.p_bytes = (unsigned)pb; p.q_bytes = (unsigned)qb;
    p.Mpad = lnn_round_up(p.M, 32); p.Cpad = lnn_round_up(p.C, 32);
    switch (p.ntaps) {
        case 27: return gen_wgrad_launch_t<7>(s, p, name);
        case 9: case 3: return gen_wgrad_launch_t<3>(s, p, name);
        case 8: case 4: return gen_wgrad_launch_t<4>(s, p, name);
        case 2: return gen_wgrad_launch_t<2>(s, p, name);
        case 1: return gen_wgrad_launch_t<1>(s, p, name);
    }
    lnn_set_error("%s: %d taps unsupported", name, p.ntaps);
    return LNN_ERR_BAD_ARG;
}

// ---- entry points ---------------------------------------------------------------------------------------------------------
namespace {
int gen_check_act(const void* ptr, int ld, int C, const char* what) {
    LNN_REQUIRE(ptr != nullptr, "%s: null pointer", what);
    LNN_REQUIRE(lnn_aligned16(ptr), "%s: pointer not 16-byte aligned", what);
    LNN_REQUIRE(C > 0 && C % 8 == 0, "%s: channel count %d must be a positive multiple of 8", what, C);
    LNN_REQUIRE(ld >= C && ld % 8 == 0, "%s: ld %d must be >= C (%d) and a multiple of 8", what, ld, C);
    return LNN_OK;
}
int gen_check_geom(const int k[3], const int st[3], bool transposed, const char* what) {
    for (int a = 0; a < 3; ++a) {
        LNN_REQUIRE(st[a] == 1 || st[a] == 2, "%s: stride %d unsupported (1 or 2 per axis)", what, st[a]);
        if (transposed) LNN_REQUIRE(k[a] == st[a], "%s: a transposed convolution has kernel == stride per axis", what);
        else LNN_REQUIRE(k[a] == 1 || k[a] == 3, "%s: kernel extent %d unsupported (1 or 3 per axis)", what, k[a]);
    }
    return LNN_OK;
}
}  // namespace

int lnn_gen_conv3d_fwd(hipStream_t s, const void* x, int ld_x, const void* wp, const float* bias, void* y, int ld_y, int N, int Di,
                       int Hi, int Wi, int C, int K, const int k[3], const int st[3], float* ws, long ws_elems) {
    if (int e = gen_check_geom(k, st, false, "lnn_conv3d_fwd_g")) return e;
    if (int e = gen_check_act(x, ld_x, C, "lnn_conv3d_fwd_g(x)")) return e;
    if (int e = gen_check_act(y, ld_y, K, "lnn_conv3d_fwd_g(y)")) return e;
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_conv3d_fwd_g: weight panel null/misaligned");
    LNN_REQUIRE(N > 0 && Di > 0 && Hi > 0 && Wi > 0, "lnn_conv3d_fwd_g: bad dims");
    if (k[0] == 1 && k[1] == 3 && k[2] == 3 && st[0] == 1 && st[1] == 1 && st[2] == 1) {
        // the first stages of anisotropic plans: the z-streaming kernel walks H with the ky taps as its rolling accumulators
        const int rc = lnn_conv_k133_on_v9(s, x, ld_x, wp, bias, y, ld_y, N, Di, Hi, Wi, C, K, 0, "lnn_conv3d_fwd_g(k133,v9)");
        if (rc >= 0) return rc;
    }
    GenParams p{};
    p.x = (const half_t*)x; p.wp = (const half_t*)wp; p.bias = bias; p.y = (half_t*)y; p.ld_x = ld_x; p.ld_y = ld_y;
    p.N = N; p.Di = Di; p.Hi = Hi; p.Wi = Wi;
    p.Do = (Di - 1) / st[0] + 1; p.Ho = (Hi - 1) / st[1] + 1; p.Wo = (Wi - 1) / st[2] + 1;
    p.Ld = p.Do; p.Lh = p.Ho; p.Lw = p.Wo; p.C = C; p.M = K;
    lnn_gen_geometry(p, 0, k, st);
    return lnn_launch_gen(s, p, ws, ws_elems, "lnn_conv3d_fwd_g");
}

int lnn_gen_conv3d_dgrad(hipStream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di, int Hi, int Wi,
                         int C, int K, const int k[3], const int st[3], int accumulate, float* ws, long ws_elems) {
    if (int e = gen_check_geom(k, st, false, "lnn_conv3d_dgrad_g")) return e;
    if (int e = gen_check_act(dy, ld_dy, K, "lnn_conv3d_dgrad_g(dy)")) return e;
    if (int e = gen_check_act(dx, ld_dx, C, "lnn_conv3d_dgrad_g(dx)")) return e;
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_conv3d_dgrad_g: weight panel null/misaligned");
    if (k[0] == 1 && k[1] == 3 && k[2] == 3 && st[0] == 1 && st[1] == 1 && st[2] == 1 && !accumulate) {
        const int rc = lnn_conv_k133_on_v9(s, dy, ld_dy, wp, nullptr, dx, ld_dx, N, Di, Hi, Wi, K, C, 1, "lnn_conv3d_dgrad_g(k133,v9)");
        if (rc >= 0) return rc;
    }
    GenParams p{};
    // roles: gathered input = dy (K channels, the conv's output extents), output = dx (C channels); panel wp[slot][C][K]
    p.x = (const half_t*)dy; p.wp = (const half_t*)wp; p.y = (half_t*)dx; p.ld_x = ld_dy; p.ld_y = ld_dx;
    p.N = N; p.Di = (Di - 1) / st[0] + 1; p.Hi = (Hi - 1) / st[1] + 1; p.Wi = (Wi - 1) / st[2] + 1;
    p.Do = Di; p.Ho = Hi; p.Wo = Wi;
    p.Ld = lnn_cdiv(Di, st[0]); p.Lh = lnn_cdiv(Hi, st[1]); p.Lw = lnn_cdiv(Wi, st[2]);
    p.C = K; p.M = C; p.accumulate = accumulate;
    lnn_gen_geometry(p, 1, k, st);
    return lnn_launch_gen(s, p, ws, ws_elems, "lnn_conv3d_dgrad_g");
}

int lnn_gen_convT3d_fwd(hipStream_t s, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N, int D, int H, int W, int C,
                        int K, const int st[3], float* ws, long ws_elems) {
    if (int e = gen_check_geom(st, st, true, "lnn_convT3d_fwd_g")) return e;
    if (int e = gen_check_act(x, ld_x, C, "lnn_convT3d_fwd_g(x)")) return e;
    if (int e = gen_check_act(y, ld_y, K, "lnn_convT3d_fwd_g(y)")) return e;
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_convT3d_fwd_g: weight panel null/misaligned");
    GenParams p{};
    p.x = (const half_t*)x; p.wp = (const half_t*)wp; p.y = (half_t*)y; p.ld_x = ld_x; p.ld_y = ld_y;
    p.N = N; p.Di = D; p.Hi = H; p.Wi = W; p.Do = D * st[0]; p.Ho = H * st[1]; p.Wo = W * st[2];
    p.Ld = D; p.Lh = H; p.Lw = W; p.C = C; p.M = K;
    lnn_gen_geometry(p, 2, st, st);
    return lnn_launch_gen(s, p, ws, ws_elems, "lnn_convT3d_fwd_g");
}

int lnn_gen_convT3d_dgrad(hipStream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int D, int H, int W,
                          int C, int K, const int st[3], int accumulate, float* ws, long ws_elems) {
    if (int e = gen_check_geom(st, st, true, "lnn_convT3d_dgrad_g")) return e;
    if (int e = gen_check_act(dy, ld_dy, K, "lnn_convT3d_dgrad_g(dy)")) return e;
    if (int e = gen_check_act(dx, ld_dx, C, "lnn_convT3d_dgrad_g(dx)")) return e;
    LNN_REQUIRE(wp != nullptr && lnn_aligned16(wp), "lnn_convT3d_dgrad_g: weight panel null/misaligned");
    GenParams p{};
    // dx[l, c] = sum_d sum_k dy[st*l + d, k] W[c, k, d]: gathered input = dy, one class, kernel-many taps
    p.x = (const half_t*)dy; p.wp = (const half_t*)wp; p.y = (half_t*)dx; p.ld_x = ld_dy; p.ld_y = ld_dx;
    p.N = N; p.Di = D * st[0]; p.Hi = H * st[1]; p.Wi = W * st[2]; p.Do = D; p.Ho = H; p.Wo = W;
    p.Ld = D; p.Lh = H; p.Lw = W; p.C = K; p.M = C; p.accumulate = accumulate;
    lnn_gen_geometry(p, 3, st, st);
    return lnn_launch_gen(s, p, ws, ws_elems, "lnn_convT3d_dgrad_g");
}

int lnn_gen_conv3d_wgrad(hipStream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int Di, int Hi, int Wi,
                         int C, int K, const int k[3], const int st[3], float* parts, long parts_elems) {
    if (int e = gen_check_geom(k, st, false, "lnn_conv3d_wgrad_g")) return e;
    if (int e = gen_check_act(x, ld_x, C, "lnn_conv3d_wgrad_g(x)")) return e;
    if (int e = gen_check_act(dy, ld_dy, K, "lnn_conv3d_wgrad_g(dy)")) return e;
    LNN_REQUIRE(dwp != nullptr, "lnn_conv3d_wgrad_g: null panel");
    GenWParams p{};
    p.p = (const half_t*)dy; p.q = (const half_t*)x; p.dwp = dwp; p.ld_p = ld_dy; p.ld_q = ld_x; p.parts = parts; p.parts_elems = parts_elems;
    p.N = N; p.Qd = Di; p.Qh = Hi; p.Qw = Wi;
    p.Ld = (Di - 1) / st[0] + 1; p.Lh = (Hi - 1) / st[1] + 1; p.Lw = (Wi - 1) / st[2] + 1;
    p.siz = st[0]; p.siy = st[1]; p.six = st[2]; p.M = K; p.C = C;
    GenParams g{};
    p.ntaps = lnn_gen_geometry(g, 0, k, st);            // same tap offsets as the forward
    p.wtaps = g.wtaps;
    for (int t = 0; t < p.ntaps; ++t) p.taps[t] = g.taps[t];
    return lnn_launch_gen_wgrad(s, p, "lnn_conv3d_wgrad_g");
}

int lnn_gen_convT3d_wgrad(hipStream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D, int H, int W,
                          int C, int K, const int st[3], float* parts, long parts_elems) {
    if (int e = gen_check_geom(st, st, true, "lnn_convT3d_wgrad_g")) return e;
    if (int e = gen_check_act(x, ld_x, C, "lnn_convT3d_wgrad_g(x)")) return e;
    if (int e = gen_check_act(dy, ld_dy, K, "lnn_convT3d_wgrad_g(dy)")) return e;
    LNN_REQUIRE(dwp != nullptr, "lnn_convT3d_wgrad_g: null panel");
    GenWParams p{};
    // dW[c, k, d] = sum_l x[l, c] dy[st*l + d, k]:  P = x (rows c), Q = dy gathered (cols k)
    p.p = (const half_t*)x; p.q = (const half_t*)dy; p.dwp = dwp; p.ld_p = ld_x; p.ld_q = ld_dy; p.parts = parts; p.parts_elems = parts_elems;
    p.N = N; p.Ld = D; p.Lh = H; p.Lw = W; p.Qd = D * st[0]; p.Qh = H * st[1]; p.Qw = W * st[2];
    p.siz = st[0]; p.siy = st[1]; p.six = st[2]; p.M = C; p.C = K;
    GenParams g{};
    p.ntaps = lnn_gen_geometry(g, 3, st, st);           // taps d = 0 .. st-1 per axis, offset d
    p.wtaps = g.wtaps;
    for (int t = 0; t < p.ntaps; ++t) p.taps[t] = g.taps[t];
    return lnn_launch_gen_wgrad(s, p, "lnn_convT3d_wgrad_g");
}

extern "C" int lnn_conv3d_fwd_g(lnn_stream_t s, const void* x, int ld_x, const void* wp, const float* bias, void* y, int ld_y, int N,
                                int Di, int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, float* splitk_ws,
                                long splitk_elems) {
    const int k[3] = {kz, ky, kx}, st[3] = {sz, sy, sx};
    return lnn_gen_conv3d_fwd((hipStream_t)s, x, ld_x, wp, bias, y, ld_y, N, Di, Hi, Wi, C, K, k, st, splitk_ws, splitk_elems);
}
extern "C" int lnn_conv3d_dgrad_g(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int Di, int Hi,
                                  int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, int accumulate, float* splitk_ws,
                                  long splitk_elems) {
    const int k[3] = {kz, ky, kx}, st[3] = {sz, sy, sx};
    return lnn_gen_conv3d_dgrad((hipStream_t)s, dy, ld_dy, wp, dx, ld_dx, N, Di, Hi, Wi, C, K, k, st, accumulate, splitk_ws, splitk_elems);
}
extern "C" int lnn_conv3d_wgrad_g(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int Di, int Hi,
                                  int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, float* parts, long parts_elems) {
    const int k[3] = {kz, ky, kx}, st[3] = {sz, sy, sx};
    LNN_REQUIRE(parts == nullptr || lnn_aligned16(parts), "lnn_conv3d_wgrad_g: scratch misaligned");
    return lnn_gen_conv3d_wgrad((hipStream_t)s, x, ld_x, dy, ld_dy, dwp, N, Di, Hi, Wi, C, K, k, st, parts, parts_elems);
}
extern "C" int lnn_convT3d_fwd_g(lnn_stream_t s, const void* x, int ld_x, const void* wp, void* y, int ld_y, int N, int D, int H, int W,
                                 int C, int K, int sz, int sy, int sx, float* splitk_ws, long splitk_elems) {
    const int st[3] = {sz, sy, sx};
    return lnn_gen_convT3d_fwd((hipStream_t)s, x, ld_x, wp, y, ld_y, N, D, H, W, C, K, st, splitk_ws, splitk_elems);
}
extern "C" int lnn_convT3d_dgrad_g(lnn_stream_t s, const void* dy, int ld_dy, const void* wp, void* dx, int ld_dx, int N, int D, int H,
                                   int W, int C, int K, int sz, int sy, int sx, int accumulate, float* splitk_ws, long splitk_elems) {
    const int st[3] = {sz, sy, sx};
    return lnn_gen_convT3d_dgrad((hipStream_t)s, dy, ld_dy, wp, dx, ld_dx, N, D, H, W, C, K, st, accumulate, splitk_ws, splitk_elems);
}
extern "C" int lnn_convT3d_wgrad_g(lnn_stream_t s, const void* x, int ld_x, const void* dy, int ld_dy, float* dwp, int N, int D, int H,
                                   int W, int C, int K, int sz, int sy, int sx, float* parts, long parts_elems) {
    const int st[3] = {sz, sy, sx};
    LNN_REQUIRE(parts == nullptr || lnn_aligned16(parts), "lnn_convT3d_wgrad_g: scratch misaligned");
    return lnn_gen_convT3d_wgrad((hipStream_t)s, x, ld_x, dy, ld_dy, dwp, N, D, H, W, C, K, st, parts, parts_elems);
}
