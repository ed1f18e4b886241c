// InstanceNorm3d(affine) + LeakyReLU, forward and backward, channels-last fp16 (HBM-bound kernels).
//
// Thread layout: a 256-thread block covers VPB = 256 / (C/8) voxels per pass; thread = (voxel lane,
// 8-channel vector) so that consecutive threads touch consecutive 16-byte vectors (fully coalesced)
// and every thread keeps a FIXED channel octet -> per-channel partial sums live in registers.
// Per-(n,c) reductions: fp32 in registers over a short strided run, then ONE fp32 partial per (block, n, c) in the
// workspace, summed in fp64 by a tiny finalize kernel.  (The first version used fp64 atomics across blocks: 1024
// blocks x N x C x 2 atomics onto N*C*2 addresses serialise in L2 -- the statistics pass of a 64-channel layer took
// 1.5x as long as the normalise pass that moves twice the bytes -- and needed a memset per call.)
#include "igemm_common.h"

namespace {

constexpr int NT = 256;
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int MAX_BLOCKS = 1024;   // blocks_for() cap; the workspace holds 2 fp32 partials per (block, n, c)
constexpr int UNR = 4;     // independent vector loads in flight per thread

// streaming accesses of the big passes: non-temporal (nt) loads / stores when LNN_IN_NT=1 (A/B switch, measurements only)
__device__ __forceinline__ half8 ld8(const half_t* p, int nt) {
    return nt ? __builtin_nontemporal_load(reinterpret_cast<const half8*>(p)) : *reinterpret_cast<const half8*>(p);
}
__device__ __forceinline__ void st8(half_t* p, const half8& v, int nt) {
    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<half8*>(p));
    else *reinterpret_cast<half8*>(p) = v;
}
int in_nt_flag() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LNN_IN_NT"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

struct RowMap {
    int C8, VPB, c8, vl;
    bool active;
};
__device__ __forceinline__ RowMap row_map(int C) {
    RowMap r;
    r.C8 = C >> 3;
    r.VPB = NT / r.C8;
    r.c8 = threadIdx.x % r.C8;
    r.vl = threadIdx.x / r.C8;
    r.active = r.vl < r.VPB;
    return r;
}

// Each block streams ONE contiguous range of voxels (multiple of VPB).  (Measured equal to a grid-stride walk on
// MI355X: these kernels sit at 3.5-5 TB/s either way.)
__device__ __forceinline__ long vrange_len(long V, const RowMap& rm) {
    const long per = (V + gridDim.x - 1) / gridDim.x;
    return (per + rm.VPB - 1) / rm.VPB * rm.VPB;
}
__device__ __forceinline__ long vrange_begin(long V, const RowMap& rm) { return (long)blockIdx.x * vrange_len(V, rm); }
__device__ __forceinline__ long vrange_end(long V, const RowMap& rm) {
    const long e = vrange_begin(V, rm) + vrange_len(V, rm);
    return e < V ? e : V;
}

// Sum per-thread partials part[NA][8] over the voxel lanes of the block and hand each (a, channel)
// total to `sink(a, channel, value)`.
template <int NA, typename Sink>
__device__ __forceinline__ void block_channel_reduce(const RowMap& rm, int C, float (&part)[NA][8], float* red, Sink sink) {
    constexpr int W = NA * 8 + 1;
    __syncthreads();
    if (rm.active) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x * W + a * 8 + e] = part[a][e];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < NA * C; o += NT) {
        const int a = o / C, c = o % C, c8 = c >> 3, e = c & 7;
        float s = 0.f;
        for (int vl = 0; vl < rm.VPB; ++vl) s += red[(vl * rm.C8 + c8) * W + a * 8 + e];
        sink(a, c, s);
    }
}

// partial sums: pws[((a * gridDim.x + block) * N + n) * C + c]
__global__ __launch_bounds__(NT) void in_stats_kernel(const half_t* __restrict__ y, long V, int C, float* pws) {
    __shared__ float red[NT * 17];
    const RowMap rm = row_map(C);
    const int n = blockIdx.y;
    const half_t* yn = y + (long)n * V * C;
    float part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) part[0][e] = part[1][e] = 0.f;
    if (rm.active) {
        // UNR independent 16-byte loads in flight per thread: one load per loop trip left every wave waiting a full
        // HBM round trip per 1 KB (the kernels sat at 3.6-4.3 TB/s = resident waves x 1 KB / latency)
        const long end = vrange_end(V, rm), step = rm.VPB;
        long v = vrange_begin(V, rm) + rm.vl;
        const half_t* yp = yn + rm.c8 * 8;
        for (; v + (UNR - 1) * step < end; v += UNR * step) {
            half8 x[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) x[u] = *reinterpret_cast<const half8*>(yp + (v + u * step) * C);
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)x[u][e];
                    part[0][e] += f;
                    part[1][e] += f * f;
                }
        }
        for (; v < end; v += step) {
            const half8 x = *reinterpret_cast<const half8*>(yp + v * C);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)x[e];
                part[0][e] += f;
                part[1][e] += f * f;
            }
        }
    }
    block_channel_reduce<2>(rm, C, part, red, [&](int a, int c, float s) {
        pws[(((long)a * gridDim.x + blockIdx.x) * gridDim.y + n) * C + c] = s;
    });
}

// Finalize kernels: 256 threads = EB entries x SL slices of the block partials (EB = 256 / SL).  They are a handful of blocks
// of pure load latency, so the loads of a thread are all independent and issued together: with 16 slices and 4 loads in flight a
// 1024-partial entry was 16 dependent round trips (15 us for the sums kernel of a 32-channel layer, 18 such launches per step);
// 64 slices x 16 loads in flight make it one.
constexpr int FSL = 64, FEB = 256 / FSL;
// fp64 sum of partials sl, sl + SL, ... of one entry (p = address of partial 0 of the entry, stride in floats)
template <int SL>
__device__ __forceinline__ double slice_sum(const float* p, long stride, int nparts, int sl) {
    double acc = 0;
    int b = sl;
    for (; b + 15 * SL < nparts; b += 16 * SL) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[(long)(b + k * SL) * stride];
#pragma unroll
        for (int k = 0; k < 16; k += 4) acc += ((double)v[k] + (double)v[k + 1]) + ((double)v[k + 2] + (double)v[k + 3]);
    }
    for (; b + 3 * SL < nparts; b += 4 * SL) {
        const float a0 = p[(long)b * stride], a1 = p[(long)(b + SL) * stride], a2 = p[(long)(b + 2 * SL) * stride],
                    a3 = p[(long)(b + 3 * SL) * stride];
        acc += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; b < nparts; b += SL) acc += (double)p[(long)b * stride];
    return acc;
}
// Sum of J values per thread over the SL slices of its entry (thread = slice * EB + entry); totals valid in the threads of
// slice 0.  Fixed order: bit-reproducible.  red: J * 256 doubles.
template <int SL, int J>
__device__ __forceinline__ void reduce_slices(double (&s)[J], double* red) {
    constexpr int EB = 256 / SL;
    const int ii = threadIdx.x % EB, sl = threadIdx.x / EB;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; ++j) red[j * 256 + threadIdx.x] = s[j];
    __syncthreads();
    if constexpr (SL > 16) {              // stage 1: slices 0..15 take slices sl + 16, sl + 32, ...
        if (sl < 16) {
#pragma unroll
            for (int j = 0; j < J; ++j)
                for (int k = sl + 16; k < SL; k += 16) s[j] += red[j * 256 + k * EB + ii];
        }
        __syncthreads();
        if (sl < 16) {
#pragma unroll
            for (int j = 0; j < J; ++j) red[j * 256 + threadIdx.x] = s[j];
        }
        __syncthreads();
    }
    if (sl == 0) {
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int k = 1; k < 16; ++k) s[j] += red[j * 256 + k * EB + ii];
    }
}

// grid = ceil(NC / FEB)
__global__ void in_stats_finalize_kernel(const float* pws, int nblk, int NC, long V, float eps, float* mean, float* rstd) {
    __shared__ double red[2 * 256];
    const int ii = threadIdx.x % FEB, sl = threadIdx.x / FEB;
    const int i = blockIdx.x * FEB + ii;
    double t[2] = {0, 0};
    if (i < NC) {
        t[0] = slice_sum<FSL>(pws + i, NC, nblk, sl);
        t[1] = slice_sum<FSL>(pws + (long)nblk * NC + i, NC, nblk, sl);
    }
    reduce_slices<FSL, 2>(t, red);
    if (i >= NC || sl != 0) return;
    const double m = t[0] / (double)V;
    double var = t[1] / (double)V - m * m;
    if (var < 0) var = 0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(NT) void in_lrelu_fwd_kernel(const half_t* __restrict__ y, half_t* __restrict__ z, int ld_z,
                                                          long V, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float slope, int nt) {
    const RowMap rm = row_map(C);
    if (!rm.active) return;
    const int n = blockIdx.y;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = rm.c8 * 8 + e;
        sc[e] = gamma[c] * rstd[n * C + c];
        sh[e] = beta[c] - mean[n * C + c] * sc[e];
    }
    const half_t* yn = y + (long)n * V * C;
    half_t* zn = z + (long)n * V * ld_z;
    auto apply = [&](const half8& x) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = (float)x[e] * sc[e] + sh[e];
            o[e] = (half_t)(t > 0.f ? t : t * slope);
        }
        return o;
    };
    const long end = vrange_end(V, rm), step = rm.VPB;
    long v = vrange_begin(V, rm) + rm.vl;
    const half_t* yp = yn + rm.c8 * 8;
    half_t* zp = zn + rm.c8 * 8;
    for (; v + (UNR - 1) * step < end; v += UNR * step) {
        half8 x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) x[u] = ld8(yp + (v + u * step) * C, nt);
#pragma unroll
        for (int u = 0; u < UNR; ++u) st8(zp + (v + u * step) * ld_z, apply(x[u]), nt);
    }
    for (; v < end; v += step) *reinterpret_cast<half8*>(zp + v * ld_z) = apply(*reinterpret_cast<const half8*>(yp + v * C));
}

// InstanceNorm + LeakyReLU + the 1x1x1 segmentation head of the SAME activation (decoder blocks that feed a seg head): the head
// reads the fp16-rounded z this kernel has in registers, so the separate seg pass over z (0.63 GB at the top level) disappears.
// A voxel's C/8 threads are adjacent lanes (C/8 a power of two <= 64): butterfly over them, lane c8 == 0 writes the K logits.
// STAGED (C in [32, 512], V % 4 == 0; A/B variant, off by default): the K logits of the UNR * VPB consecutive voxels a block
// handles per trip go through LDS and leave as 16-byte stores of complete runs instead of 64-byte pieces per wave-wide store.
constexpr int SEG_KMAX = 8;
// KT: compile-time K (1..4: the runtime bound made every one of the 8 x SEG_KMAX products of a channel octet a scalar branch, the
// kernel ran at 1.7 TB/s of reads with or without its store of z), 0 = runtime K <= SEG_KMAX.
template <bool STAGED, int KT>
__global__ __launch_bounds__(NT) void in_lrelu_seg_fwd_kernel(const half_t* __restrict__ y, half_t* __restrict__ z, int ld_z,
                                                              long V, int C, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float slope,
                                                              const float* __restrict__ segw, float* __restrict__ logits, int K) {
    constexpr int KB = KT > 0 ? KT : SEG_KMAX;
    __shared__ __attribute__((aligned(16))) float lg[STAGED ? 2 * SEG_KMAX * UNR * 64 : 4];
    const RowMap rm = row_map(C);
    if (!STAGED && !rm.active) return;                 // (STAGED: C / 8 is a power of two, every thread is active)
    const int n = blockIdx.y;
    float sc[8], sh[8], w[KB][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = rm.c8 * 8 + e;
        sc[e] = gamma[c] * rstd[n * C + c];
        sh[e] = beta[c] - mean[n * C + c] * sc[e];
#pragma unroll
        for (int k = 0; k < KB; ++k) w[k][e] = k < K ? segw[k * C + c] : 0.f;
    }
    const half_t* yp = y + (long)n * V * C + rm.c8 * 8;
    half_t* zp = z + (long)n * V * ld_z + rm.c8 * 8;
    float* ln = logits + (long)n * K * V;
    const long end = vrange_end(V, rm), step = rm.VPB;
    // the block's range is a multiple of VPB, so all C8 lanes of a voxel are in or out together
    auto one = [&](const half8& x, long v, float* stage) {
        half8 o;
        float pk[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) pk[k] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = (float)x[e] * sc[e] + sh[e];
            o[e] = (half_t)(t > 0.f ? t : t * slope);
            const float zf = (float)o[e];
#pragma unroll
            for (int k = 0; k < KB; ++k) pk[k] += zf * w[k][e];           // (w is zero beyond K)
        }
        if (z) *reinterpret_cast<half8*>(zp + v * ld_z) = o;        // (kernel-uniform: z == nullptr -> logits only)
        for (int m = 1; m < rm.C8; m <<= 1) {
#pragma unroll
            for (int k = 0; k < KB; ++k) pk[k] += __shfl_xor(pk[k], m, 64);
        }
        if (rm.c8 == 0) {
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (KT > 0 || k < K) {
                    if (stage) stage[k * (UNR * 64)] = pk[k];
                    else ln[(long)k * V + v] = pk[k];
                }
        }
    };
    long v = vrange_begin(V, rm) + rm.vl;
    if constexpr (STAGED) {
        // block-uniform trips over runs of UNR * VPB consecutive voxels [base, base + run): thread (vl, u) handles voxel
        // base + u * VPB + vl; double-buffered staging, one barrier per trip
        const long run = (long)UNR * step;
        long base = vrange_begin(V, rm);
        int it = 0;
        for (; base + run <= end; base += run, ++it) {
            float* buf = lg + (it & 1) * (SEG_KMAX * UNR * 64);
            half8 x[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) x[u] = *reinterpret_cast<const half8*>(yp + (base + rm.vl + u * step) * C);
#pragma unroll
            for (int u = 0; u < UNR; ++u) one(x[u], base + rm.vl + u * step, buf + u * step + rm.vl);
            __syncthreads();
            const int q4 = (int)(run >> 2);                  // 16-byte pieces per class
            for (int t = threadIdx.x; t < K * q4; t += NT) {
                const int k = t / q4, j = t - k * q4;
                *reinterpret_cast<floatx4*>(ln + (long)k * V + base + 4 * j) = *reinterpret_cast<const floatx4*>(buf + k * (UNR * 64) + 4 * j);
            }
        }
        v = base + rm.vl;
    } else {
        for (; v + (UNR - 1) * step < end; v += UNR * step) {      // UNR loads in flight per thread, as the plain forward pass
            half8 x[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) x[u] = *reinterpret_cast<const half8*>(yp + (v + u * step) * C);
#pragma unroll
            for (int u = 0; u < UNR; ++u) one(x[u], v + u * step, nullptr);
        }
    }
    for (; v < end; v += step) one(*reinterpret_cast<const half8*>(yp + v * C), v, nullptr);
}

// pass 1 of backward: s1 = sum g, s2 = sum g*xhat with g = dz * lrelu'(gamma*xhat+beta)
__global__ __launch_bounds__(NT) void in_lrelu_bwd_reduce_kernel(const half_t* __restrict__ y, const half_t* __restrict__ dz,
                                                                 int ld_dz, long V, int C, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float slope, float* pws, int nt) {
    __shared__ float red[NT * 17];
    const RowMap rm = row_map(C);
    const int n = blockIdx.y;
    float part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) part[0][e] = part[1][e] = 0.f;
    if (rm.active) {
        float mu[8], rs[8], ga[8], be[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = rm.c8 * 8 + e;
            mu[e] = mean[n * C + c]; rs[e] = rstd[n * C + c]; ga[e] = gamma[c]; be[e] = beta[c];
        }
        const half_t* yn = y + (long)n * V * C;
        const half_t* dzn = dz + (long)n * V * ld_dz;
        auto accum = [&](const half8& x, const half8& d) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)x[e] - mu[e]) * rs[e];
                const float pre = ga[e] * xh + be[e];
                const float g = (float)d[e] * (pre > 0.f ? 1.f : slope);
                part[0][e] += g;
                part[1][e] += g * xh;
            }
        };
        const long end = vrange_end(V, rm), step = rm.VPB;
        long v = vrange_begin(V, rm) + rm.vl;
        const half_t* yp = yn + rm.c8 * 8;
        const half_t* dp = dzn + rm.c8 * 8;
        constexpr int U2 = UNR / 2;
        for (; v + (U2 - 1) * step < end; v += U2 * step) {
            half8 x[U2], d[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                x[u] = *reinterpret_cast<const half8*>(yp + (v + u * step) * C);       // y and dz are read again by pass 2: cached loads
                d[u] = *reinterpret_cast<const half8*>(dp + (v + u * step) * ld_dz);
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) accum(x[u], d[u]);
        }
        for (; v < end; v += step)
            accum(*reinterpret_cast<const half8*>(yp + v * C), *reinterpret_cast<const half8*>(dp + v * ld_dz));
    }
    block_channel_reduce<2>(rm, C, part, red, [&](int a, int c, float s) {
        pws[(((long)a * gridDim.x + blockIdx.x) * gridDim.y + n) * C + c] = s;
    });
}

// ws[(n*C + c)*3 + {0,1}] = s1, s2 (fp64 totals of the partials)
// also adds the affine-parameter gradients (dbeta = sum g, dgamma = sum g * xhat): they do not depend on pass 2
// One thread group per CHANNEL, samples walked in order: the affine gradients are one ordered fp64 sum over n and ONE add per
// call and channel (bit-reproducible for any batch size; the atomic only serves two sample lanes adding from two streams,
// and two operands commute).  NC = N * C, grid = ceil(C / 16).
// (n, c) totals of pass 1 and the affine gradients of FEB channels per block: shared by the plain and the seg-head variant
// raw_mean / raw_rstd (both or neither): the second partial is sum g * u of the UN-normalised convolution output u (the reduce fused
// into the data gradient that produced dz, igemm_conv_v9.hip EPI = 2): sum g xhat = rstd (sum g u - mean sum g), taken here in fp64
__device__ __forceinline__ void in_bwd_sums_block(const float* pws, int nblk, int N, int C, int blk, double* red, double* ws,
                                                  float* dgamma, float* dbeta, float unscale, const float* raw_mean = nullptr,
                                                  const float* raw_rstd = nullptr) {
    const int ii = threadIdx.x % FEB, sl = threadIdx.x / FEB;
    const int c = blk * FEB + ii, NC = N * C;
    double g0 = 0, g1 = 0;
    for (int n = 0; n < N; n += 2) {                   // two samples per reduction
        const bool two = n + 1 < N;
        double o[4] = {0, 0, 0, 0};
        if (c < C) {
            const float* p0 = pws + n * C + c;
            o[0] = slice_sum<FSL>(p0, NC, nblk, sl);
            o[1] = slice_sum<FSL>(p0 + (long)nblk * NC, NC, nblk, sl);
            if (two) {
                o[2] = slice_sum<FSL>(p0 + C, NC, nblk, sl);
                o[3] = slice_sum<FSL>(p0 + C + (long)nblk * NC, NC, nblk, sl);
            }
        }
        reduce_slices<FSL, 4>(o, red);
        if (c < C && sl == 0) {
            const long i0 = (long)n * C + c;
            if (raw_mean) {
                o[1] = (double)raw_rstd[i0] * (o[1] - (double)raw_mean[i0] * o[0]);
                if (two) o[3] = (double)raw_rstd[i0 + C] * (o[3] - (double)raw_mean[i0 + C] * o[2]);
            }
            ws[i0 * 3 + 0] = o[0]; ws[i0 * 3 + 1] = o[1];
            g0 += o[0]; g1 += o[1];
            if (two) {
                ws[(i0 + C) * 3 + 0] = o[2]; ws[(i0 + C) * 3 + 1] = o[3];
                g0 += o[2]; g1 += o[3];
            }
        }
    }
    if (c >= C || sl != 0) return;
    if (dgamma) atomicAdd(dgamma + c, (float)(g1 * unscale));
    if (dbeta) atomicAdd(dbeta + c, (float)(g0 * unscale));
}
__global__ void in_lrelu_bwd_sums_kernel(const float* pws, int nblk, int NC, int C, double* ws, float* dgamma, float* dbeta,
                                         float unscale, const float* raw_mean, const float* raw_rstd) {
    __shared__ double red[4 * 256];
    in_bwd_sums_block(pws, nblk, NC / C, C, blockIdx.x, red, ws, dgamma, dbeta, unscale, raw_mean, raw_rstd);
}

// pass 2: dy = gamma*rstd*(g - s1/V - xhat*s2/V), in place over y; DBIAS: db partial = sum dy (the conv-bias gradient; it is
// analytically ZERO -- InstanceNorm removes the mean, so sum_v dy = 0 -- and what this sums is the fp16 rounding of dy; the
// training engine does not ask for it and the pass is then a pure stream without block reductions)
template <bool DBIAS>
__global__ __launch_bounds__(NT) void in_lrelu_bwd_apply_kernel(half_t* __restrict__ y, const half_t* __restrict__ dz, int ld_dz,
                                                                long V, int C, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float slope, const double* ws,
                                                                float* pws, int nt) {
    __shared__ float red[NT * 9];
    const RowMap rm = row_map(C);
    const int n = blockIdx.y;
    float part[1][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) part[0][e] = 0.f;
    if (rm.active) {
        float mu[8], rs[8], ga[8], be[8], m1[8], m2[8];
        const float invV = 1.0f / (float)V;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = rm.c8 * 8 + e;
            mu[e] = mean[n * C + c]; rs[e] = rstd[n * C + c]; ga[e] = gamma[c]; be[e] = beta[c];
            m1[e] = (float)(ws[((long)n * C + c) * 3 + 0] * (double)invV);
            m2[e] = (float)(ws[((long)n * C + c) * 3 + 1] * (double)invV);
        }
        half_t* yn = y + (long)n * V * C;
        const half_t* dzn = dz + (long)n * V * ld_dz;
        auto grad = [&](const half8& x, const half8& d) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)x[e] - mu[e]) * rs[e];
                const float pre = ga[e] * xh + be[e];
                const float g = (float)d[e] * (pre > 0.f ? 1.f : slope);
                const half_t r = (half_t)(ga[e] * rs[e] * (g - m1[e] - xh * m2[e]));
                o[e] = r;
                if (DBIAS) part[0][e] += (float)r;
            }
            return o;
        };
        const long end = vrange_end(V, rm), step = rm.VPB;
        long v = vrange_begin(V, rm) + rm.vl;
        half_t* yp = yn + rm.c8 * 8;
        const half_t* dp = dzn + rm.c8 * 8;
        constexpr int U2 = UNR / 2;
        for (; v + (U2 - 1) * step < end; v += U2 * step) {
            half8 x[U2], d[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                x[u] = ld8(yp + (v + u * step) * C, nt);
                d[u] = ld8(dp + (v + u * step) * ld_dz, nt);
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) st8(yp + (v + u * step) * C, grad(x[u], d[u]), nt);
        }
        for (; v < end; v += step)
            *reinterpret_cast<half8*>(yp + v * C) = grad(*reinterpret_cast<const half8*>(yp + v * C), *reinterpret_cast<const half8*>(dp + v * ld_dz));
    }
    if (DBIAS)
        block_channel_reduce<1>(rm, C, part, red, [&](int, int c, float s) {
            pws[((long)blockIdx.x * gridDim.y + n) * C + c] = s;
        });
}

__global__ void in_lrelu_bwd_finalize_kernel(const double* ws, const float* pws, int nblk, int N, int C, float* dgamma,
                                             float* dbeta, float* dbias, float unscale) {
    __shared__ double red[256];
    const int ii = threadIdx.x % FEB, sl = threadIdx.x / FEB;
    const int i = blockIdx.x * FEB + ii;                          // (n, c) entry
    double t[1] = {i < N * C ? slice_sum<FSL>(pws + i, N * C, nblk, sl) : 0.0};
    reduce_slices<FSL, 1>(t, red);
    const double db = t[0];
    if (i >= N * C || sl != 0) return;
    const int c = i % C;
    // atomics: N samples (and, with sample lanes, two HIP streams) add into the same channel
    (void)ws; (void)dgamma; (void)dbeta;        // the affine gradients are added by the sums kernel
    if (dbias) atomicAdd(dbias + c, (float)(db * unscale));
}

// ---- decoder blocks that feed a seg_outputs head: 1x1x1-head backward + InstanceNorm/LeakyReLU backward without dL/dz in HBM ----
// The unfused plan runs lnn_seg1x1_bwd (reads z, writes dz), then the two passes above (each reads dz and y): at the top
// level dz and z are 0.63 GB tensors while dlogits is 0.12 GB.  Here both passes rebuild what they need in registers:
//   dz  = [prior dz, written by the transposed conv of the level above] + fp16( sum_k dlogits[k] w[k][c] )
//   z   = fp16( lrelu(y * gamma*rstd + beta - mean*gamma*rstd) )   -- the forward pass's own expression, for d(seg w)
// so pass 1 reads y (+ prior) + dlogits and pass 2 the same, writing dy over y.  Pass 2 keeps the unfused kernels' rounding
// point (dz through fp16); pass 1 sums with dz and z in fp32.  lrelu' is taken at the forward expression t = y*sc + sh instead
// of gamma*xhat + beta (equal up to fp32 rounding, i.e. except for |t| ~ 1e-7).  K <= 4 (compile-time K).
template <int KT>
__device__ __forceinline__ void seg_dz(const float (&d)[KT], const float (&wr)[KT][8], const half8* prior, half8& dzh) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float o = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) o += d[k] * wr[k][e];
        if (prior) o += (float)(*prior)[e];
        dzh[e] = (half_t)o;
    }
}

// The K dlogits of a voxel.  QUAD (C % 32 == 0: the four lanes of a quad always hold octets of the SAME voxel): lane j of the
// quad loads channel j, the others get it through a DPP quad broadcast -- one 4-byte load per voxel and thread instead of K.
template <int KT, bool QUAD>
struct DlFetch {
    float raw[QUAD ? 1 : KT];
    __device__ __forceinline__ void load(const float* __restrict__ dln, long V, long v) {
        if constexpr (QUAD) {
            const int kq = (int)(threadIdx.x & 3);
            raw[0] = dln[(long)(kq < KT ? kq : KT - 1) * V + v];
        } else {
#pragma unroll
            for (int k = 0; k < KT; ++k) raw[k] = dln[(long)k * V + v];
        }
    }
    __device__ __forceinline__ void get(float (&d)[KT]) const {
        if constexpr (QUAD) {
            const int r = __builtin_bit_cast(int, raw[0]);
            d[0] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(r, 0x00, 0xf, 0xf, true));
            if constexpr (KT > 1) d[1] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(r, 0x55, 0xf, 0xf, true));
            if constexpr (KT > 2) d[2] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(r, 0xaa, 0xf, 0xf, true));
            if constexpr (KT > 3) d[3] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(r, 0xff, 0xf, 0xf, true));
        } else {
#pragma unroll
            for (int k = 0; k < KT; ++k) d[k] = raw[k];
        }
    }
};

// pass 1: s1 = sum g, s2 = sum g * xhat (as in_lrelu_bwd_reduce_kernel) and the head's weight-gradient partials
// pws layout: [2][nblk][N][C] (s1, s2) then [N][nblk][KT][C] (d seg_w)
template <int KT, bool PRIOR, bool QUAD>
__global__ __launch_bounds__(NT, (KT <= 3 && !PRIOR && QUAD) ? 3 : 2) void in_lrelu_seg_bwd_reduce_kernel(const half_t* __restrict__ y, const half_t* __restrict__ dzp,
                                                                     int ld_dz, long V, int C, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float slope,
                                                                     const float* __restrict__ segw, const float* __restrict__ dl,
                                                                     float* pws) {
    constexpr int NA = 2 + KT;
    __shared__ float red[NT * 17];
    const RowMap rm = row_map(C);
    const int n = blockIdx.y;
    float part[NA][8];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) part[a][e] = 0.f;
    if (rm.active) {
        float rs[8], nmr[8], sc[8], sh[8], wr[KT][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = rm.c8 * 8 + e;
            rs[e] = rstd[n * C + c];
            nmr[e] = -mean[n * C + c] * rs[e];
            sc[e] = gamma[c] * rs[e];
            sh[e] = beta[c] - mean[n * C + c] * sc[e];
#pragma unroll
            for (int k = 0; k < KT; ++k) wr[k][e] = segw[k * C + c];
        }
        const half_t* yp = y + (long)n * V * C + rm.c8 * 8;
        const half_t* dp = dzp + (long)n * V * ld_dz + rm.c8 * 8;
        const float* dln = dl + (long)n * KT * V;
        // ~15 fp32 operations per element, written on channel PAIRS so that they become packed (v_pk_*_f32) instructions:
        // this pass is VALU-bound, not HBM-bound (0.75 GB for 315 M elements at the top level).  lrelu(t) = t * sel with
        // sel = lrelu'(t) in {1, slope}; dz and z are used in fp32 here (the unfused kernels round both to fp16 in memory).
        auto accum = [&](const half8& x, const half8& pr, const float (&d)[KT]) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float2v xf = {(float)x[e], (float)x[e + 1]};
                const float2v sc2 = {sc[e], sc[e + 1]}, sh2 = {sh[e], sh[e + 1]}, rs2 = {rs[e], rs[e + 1]}, nm2 = {nmr[e], nmr[e + 1]};
                const float2v t = xf * sc2 + sh2;
                const float2v sel = {t[0] > 0.f ? 1.f : slope, t[1] > 0.f ? 1.f : slope};
                const float2v z = t * sel;
                const float2v xh = xf * rs2 + nm2;
                float2v dz = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const float2v wk = {wr[k][e], wr[k][e + 1]};
                    dz += wk * d[k];
                }
                if (PRIOR) dz += float2v{(float)pr[e], (float)pr[e + 1]};
                const float2v g = dz * sel;
                float2v p0 = {part[0][e], part[0][e + 1]}, p1 = {part[1][e], part[1][e + 1]};
                p0 += g;
                p1 += g * xh;
                part[0][e] = p0[0]; part[0][e + 1] = p0[1];
                part[1][e] = p1[0]; part[1][e + 1] = p1[1];
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    float2v pk = {part[2 + k][e], part[2 + k][e + 1]};
                    pk += z * d[k];
                    part[2 + k][e] = pk[0]; part[2 + k][e + 1] = pk[1];
                }
            }
        };
        const long end = vrange_end(V, rm), step = rm.VPB;
        long v = vrange_begin(V, rm) + rm.vl;
        // voxels in flight per thread: this pass keeps ~100 registers of sums / constants (3 waves per SIMD), so the bytes in
        // flight have to come from the loop: 3 x 16 B of y (+ one 4-byte dlogit each) per thread; a 4th would cost the third wave
        constexpr int U2 = (PRIOR || !QUAD) ? UNR / 2 : 3;
        for (; v + (U2 - 1) * step < end; v += U2 * step) {
            half8 x[U2], pr[U2];
            DlFetch<KT, QUAD> df[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                x[u] = *reinterpret_cast<const half8*>(yp + (v + u * step) * C);
                if (PRIOR) pr[u] = *reinterpret_cast<const half8*>(dp + (v + u * step) * ld_dz);
                df[u].load(dln, V, v + u * step);
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                float d[KT];
                df[u].get(d);
                accum(x[u], pr[u], d);
            }
        }
        for (; v < end; v += step) {
            half8 pr;
            float d[KT];
            DlFetch<KT, QUAD> df;
            if (PRIOR) pr = *reinterpret_cast<const half8*>(dp + v * ld_dz);
            df.load(dln, V, v);
            df.get(d);
            accum(*reinterpret_cast<const half8*>(yp + v * C), pr, d);
        }
    }
    float* const pseg = pws + 2l * gridDim.x * gridDim.y * C;
    // the block reduction two accumulators at a time through the 17-float rows of `red` (LDS stays at 17 KB per block)
#pragma unroll
    for (int a0 = 0; a0 < NA; a0 += 2) {
        float two[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            two[0][e] = part[a0][e];
            two[1][e] = a0 + 1 < NA ? part[a0 + 1 < NA ? a0 + 1 : a0][e] : 0.f;
        }
        block_channel_reduce<2>(rm, C, two, red, [&](int a, int c, float s) {
            const int aa = a0 + a;
            if (aa < 2) pws[(((long)aa * gridDim.x + blockIdx.x) * gridDim.y + n) * C + c] = s;
            else if (aa < NA) pseg[(((long)n * gridDim.x + blockIdx.x) * KT + (aa - 2)) * C + c] = s;
        });
    }
}

// blocks [0, nb_in): the (n, c) totals of pass 1 (= in_lrelu_bwd_sums_kernel); blocks [nb_in, ..): d seg_w[k][c] += unscale * total
__global__ void in_lrelu_seg_bwd_sums_kernel(const float* pws, int nblk, int N, int C, int K, int nb_in, double* ws, float* dgamma,
                                             float* dbeta, float* dsegw, float unscale) {
    __shared__ double red[4 * 256];
    if ((int)blockIdx.x < nb_in) {                     // as in_lrelu_bwd_sums_kernel: per channel, samples in order
        in_bwd_sums_block(pws, nblk, N, C, blockIdx.x, red, ws, dgamma, dbeta, unscale);
        return;
    }
    // d seg_w: FEB consecutive entries (k * C + c) x FSL slices of the N * nblk block partials (partial b of entry i sits at
    // pseg[b * K * C + i])
    const float* pseg = pws + 2l * nblk * N * C;
    const int ii = threadIdx.x % FEB, sl = threadIdx.x / FEB;
    const int i = (blockIdx.x - nb_in) * FEB + ii;
    double t[1] = {i < K * C ? slice_sum<FSL>(pseg + i, (long)K * C, nblk * N, sl) : 0.0};
    reduce_slices<FSL, 1>(t, red);
    if (sl == 0 && i < K * C) atomicAdd(dsegw + i, (float)(t[0] * unscale));      // (one add per entry and launch; two sample lanes may add concurrently)
}

// pass 2: dy = gamma*rstd*(g - s1/V - xhat*s2/V) over y, dz rebuilt as in pass 1
template <int KT, bool PRIOR, bool QUAD>
__global__ __launch_bounds__(NT) void in_lrelu_seg_bwd_apply_kernel(half_t* __restrict__ y, const half_t* __restrict__ dzp, int ld_dz,
                                                                    long V, int C, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float slope,
                                                                    const float* __restrict__ segw, const float* __restrict__ dl,
                                                                    const double* ws) {
    const RowMap rm = row_map(C);
    if (!rm.active) return;
    const int n = blockIdx.y;
    // dy = sc * (g - m1 - xhat * m2) with xhat = y * rs - mean * rs, regrouped per channel as  sc * g + (ca + y * cb):
    // four constants per channel instead of six (this pass sits at the 128-register / 4-waves-per-SIMD boundary)
    float sc[8], sh[8], ca[8], cb[8], wr[KT][8];
    const float invV = 1.0f / (float)V;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = rm.c8 * 8 + e;
        const float rs = rstd[n * C + c], nmr = -mean[n * C + c] * rs;
        const float m1 = (float)(ws[((long)n * C + c) * 3 + 0] * (double)invV);
        const float m2 = (float)(ws[((long)n * C + c) * 3 + 1] * (double)invV);
        sc[e] = gamma[c] * rs;
        sh[e] = beta[c] - mean[n * C + c] * sc[e];
        ca[e] = -sc[e] * (m1 + nmr * m2);
        cb[e] = -sc[e] * rs * m2;
#pragma unroll
        for (int k = 0; k < KT; ++k) wr[k][e] = segw[k * C + c];
    }
    half_t* yp = y + (long)n * V * C + rm.c8 * 8;
    const half_t* dp = dzp + (long)n * V * ld_dz + rm.c8 * 8;
    const float* dln = dl + (long)n * KT * V;
    auto grad = [&](const half8& x, const half8& pr, const float (&d)[KT]) {
        half8 dzh, o;
        seg_dz<KT>(d, wr, PRIOR ? &pr : nullptr, dzh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xf = (float)x[e];
            const float t = xf * sc[e] + sh[e];
            const float g = (float)dzh[e] * (t > 0.f ? 1.f : slope);
            o[e] = (half_t)(sc[e] * g + (xf * cb[e] + ca[e]));
        }
        return o;
    };
    const long end = vrange_end(V, rm), step = rm.VPB;
    long v = vrange_begin(V, rm) + rm.vl;
    constexpr int U2 = UNR / 2;
    for (; v + (U2 - 1) * step < end; v += U2 * step) {
        half8 x[U2], pr[U2];
        DlFetch<KT, QUAD> df[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            x[u] = *reinterpret_cast<const half8*>(yp + (v + u * step) * C);
            if (PRIOR) pr[u] = *reinterpret_cast<const half8*>(dp + (v + u * step) * ld_dz);
            df[u].load(dln, V, v + u * step);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            float d[KT];
            df[u].get(d);
            *reinterpret_cast<half8*>(yp + (v + u * step) * C) = grad(x[u], pr[u], d);
        }
    }
    for (; v < end; v += step) {
        half8 pr;
        float d[KT];
        DlFetch<KT, QUAD> df;
        if (PRIOR) pr = *reinterpret_cast<const half8*>(dp + v * ld_dz);
        df.load(dln, V, v);
        df.get(d);
        *reinterpret_cast<half8*>(yp + v * C) = grad(*reinterpret_cast<const half8*>(yp + v * C), pr, d);
    }
}

// ---- small volumes (the two lowest levels of the U: <= 2048 voxels per sample) ---------------------------------------------------
// There a normalisation is ~1.5 MB and every launch of the multi-block passes costs its fixed ~6 us: statistics, finalize and normalise
// (forward) / reduce, sums and apply (backward) were 3 dependent launches per layer for ~3 us of work.  One launch each: a 512-thread
// block owns 32 CHANNELS of one sample (forward) or of every sample in turn (backward: the affine gradients are an ordered sum over
// the samples): thread = (row lane 0..127, channel octet 0..3), so a wave-wide load covers 16 rows x 64 contiguous bytes; the rows
// are read a second time (from the L2) for the normalisation; sums: butterfly over the row lanes of a wave, then one LDS step in
// fp64.  (A first version gave a block ONE octet -- every lane of a load in its own cache line -- and
// also added the convolution's split-K slices itself: 30-70 us per launch, profiles/r06_small_volume_norm.txt; the slices are 8-25 MB
// and need the whole chip, so lnn_launch_splitk_finalize stays a launch of its own.)
constexpr int SNT = 512, SRL = SNT / 4;        // threads, row lanes
constexpr int SMALL_V = SRL * 16;              // 2048: at most 16 rows per thread and pass

// fp64 totals of 16 per-thread values (2 sums x 8 channels of the thread's octet) over the 128 row lanes: afterwards
// dtot[octet * 16 + j] (shared) holds total j of the octet.  Lanes of one octet within a wave: tid & 3 fixed -> butterfly over
// lane bits 2..5, then the 8 waves meet in LDS.
__device__ __forceinline__ void small_block_sum16(float (&v)[16], double* red, double* dtot) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) v[j] += __shfl_xor(v[j], o, 64);
    }
    __syncthreads();                       // (the previous round's readers of red / dtot are done)
    if (lane < 4) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[((lane * 16) + j) * (SNT / 64) + wid] = (double)v[j];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < SNT / 64; ++w) t += red[threadIdx.x * (SNT / 64) + w];
        dtot[threadIdx.x] = t;
    }
    __syncthreads();
}

// Forward: statistics of y -> mean / rstd -> z = LeakyReLU(gamma * xhat + beta).  grid (C / 32, N); C % 32 == 0 not required: octets
// beyond C are idle lanes.  Rows r = rl + k SRL, SU rows per trip with all their loads in flight (rows beyond V read row V - 1 again:
// unconditional loads; they take no part in the sums / stores); the second pass re-reads y (L2-resident: the tensor is ~1.5 MB).
constexpr int SU = 4;
__global__ __launch_bounds__(SNT) void in_small_fwd_kernel(const half_t* __restrict__ y, half_t* __restrict__ z, int ld_z, int V, int C,
                                                           float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float slope, float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double red[64 * (SNT / 64)];
    __shared__ double dtot[64];
    __shared__ float bc[64];
    const int oc = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int n = blockIdx.y;
    const bool chan_ok = blockIdx.x * 32 + oc * 8 < C;
    const int c0 = chan_ok ? blockIdx.x * 32 + oc * 8 : 0;
    const half_t* yn = y + (long)n * V * C + c0;
    half_t* zn = z + (long)n * V * ld_z + c0;
    float part[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) part[e] = 0.f;
    for (int r0 = rl; r0 < V; r0 += SU * SRL) {
        half8 x[SU];
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            const int r = r0 + k * SRL;
            x[k] = *reinterpret_cast<const half8*>(yn + (r < V ? r : V - 1) * C);
        }
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            const bool live = r0 + k * SRL < V && chan_ok;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = live ? (float)x[k][e] : 0.f;
                part[e] += f;
                part[8 + e] += f * f;
            }
        }
    }
    small_block_sum16(part, red, dtot);
    if (threadIdx.x < 32) {                  // thread = channel of the block's 32
        const int o = threadIdx.x >> 3, e = threadIdx.x & 7, c = blockIdx.x * 32 + threadIdx.x;
        if (c < C) {
            const double m = dtot[o * 16 + e] / (double)V;
            double var = dtot[o * 16 + 8 + e] / (double)V - m * m;
            if (var < 0) var = 0;
            const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
            mean[n * C + c] = mf; rstd[n * C + c] = rf;
            const float sc = gamma[c] * rf;
            bc[threadIdx.x] = sc;
            bc[32 + threadIdx.x] = beta[c] - mf * sc;
        }
    }
    __syncthreads();
    if (!chan_ok) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = bc[oc * 8 + e]; sh[e] = bc[32 + oc * 8 + e]; }
    for (int r0 = rl; r0 < V; r0 += SU * SRL) {
        half8 x[SU];
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            const int r = r0 + k * SRL;
            x[k] = *reinterpret_cast<const half8*>(yn + (r < V ? r : V - 1) * C);
        }
#pragma unroll
        for (int k = 0; k < SU; ++k) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = (float)x[k][e] * sc[e] + sh[e];
                o[e] = (half_t)(t > 0.f ? t : t * slope);
            }
            const int r = r0 + k * SRL;
            if (r < V) *reinterpret_cast<half8*>(zn + r * ld_z) = o;
        }
    }
}

// Backward: g = dz * lrelu'(gamma xhat + beta); s1 = sum g, s2 = sum g xhat per (n, c); dy = gamma rstd (g - s1 / V - xhat s2 / V) in
// place over y; d gamma / d beta = ordered sums over the samples, one add per channel (as in_bwd_sums_block).  grid (C / 32).
__global__ __launch_bounds__(SNT) void in_small_bwd_kernel(half_t* __restrict__ y, const half_t* __restrict__ dz, int ld_dz, int N, int V,
                                                           int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float slope,
                                                           double* __restrict__ ws, float* dgamma, float* dbeta, float unscale) {
    __shared__ double red[64 * (SNT / 64)];
    __shared__ double dtot[64];
    __shared__ float bc[64];
    const int oc = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const bool chan_ok = blockIdx.x * 32 + oc * 8 < C;
    const int c0 = chan_ok ? blockIdx.x * 32 + oc * 8 : 0;
    float ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; }
    const float invV = 1.0f / (float)V;
    double gacc = 0;                      // thread t < 64: sum over n of total (t & 15) of octet t >> 4 (s1 of 8 channels, then s2)
    for (int n = 0; n < N; ++n) {
        float mu[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean[n * C + c0 + e]; rs[e] = rstd[n * C + c0 + e]; }
        half_t* yn = y + (long)n * V * C + c0;
        const half_t* dn = dz + (long)n * V * ld_dz + c0;
        float part[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) part[e] = 0.f;
        for (int r0 = rl; r0 < V; r0 += SU * SRL) {
            half8 xv[SU], gv[SU];
#pragma unroll
            for (int k = 0; k < SU; ++k) {
                const int r = r0 + k * SRL, rc = r < V ? r : V - 1;
                xv[k] = *reinterpret_cast<const half8*>(yn + rc * C);
                gv[k] = *reinterpret_cast<const half8*>(dn + rc * ld_dz);
            }
#pragma unroll
            for (int k = 0; k < SU; ++k) {
                const bool live = r0 + k * SRL < V && chan_ok;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float h = ((float)xv[k][e] - mu[e]) * rs[e];
                    const float pre = ga[e] * h + be[e];
                    const float gg = live ? (float)gv[k][e] * (pre > 0.f ? 1.f : slope) : 0.f;
                    part[e] += gg;
                    part[8 + e] += gg * h;
                }
            }
        }
        small_block_sum16(part, red, dtot);
        if (threadIdx.x < 64) {
            const double t = dtot[threadIdx.x];
            const int o = threadIdx.x >> 4, j = threadIdx.x & 15, c = blockIdx.x * 32 + o * 8 + (j & 7);
            bc[threadIdx.x] = (float)(t * (double)invV);
            gacc += t;
            if (c < C) ws[((long)n * C + c) * 3 + (j >> 3)] = t;
        }
        __syncthreads();
        if (chan_ok) {
            float m1[8], m2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { m1[e] = bc[oc * 16 + e]; m2[e] = bc[oc * 16 + 8 + e]; }
            for (int r0 = rl; r0 < V; r0 += SU * SRL) {
                half8 xv[SU], gv[SU];
#pragma unroll
                for (int k = 0; k < SU; ++k) {
                    const int r = r0 + k * SRL, rc = r < V ? r : V - 1;
                    xv[k] = *reinterpret_cast<const half8*>(yn + rc * C);
                    gv[k] = *reinterpret_cast<const half8*>(dn + rc * ld_dz);
                }
#pragma unroll
                for (int k = 0; k < SU; ++k) {
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float h = ((float)xv[k][e] - mu[e]) * rs[e];
                        const float pre = ga[e] * h + be[e];
                        const float gg = (float)gv[k][e] * (pre > 0.f ? 1.f : slope);
                        o[e] = (half_t)(ga[e] * rs[e] * (gg - m1[e] - h * m2[e]));
                    }
                    const int r = r0 + k * SRL;
                    if (r < V) *reinterpret_cast<half8*>(yn + r * C) = o;
                }
            }
        }
    }
    if (threadIdx.x < 64) {
        const int o = threadIdx.x >> 4, j = threadIdx.x & 15, c = blockIdx.x * 32 + o * 8 + (j & 7);
        if (c < C) {
            if (j < 8) { if (dbeta) atomicAdd(dbeta + c, (float)(gacc * unscale)); }
            else if (dgamma) atomicAdd(dgamma + c, (float)(gacc * unscale));
        }
    }
}

// LNN_IN_SMALL=0: the multi-launch passes on every volume (A/B switch, measurements only)
bool small_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LNN_IN_SMALL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}

int blocks_for(long V, int C) {
    const int vpb = NT / (C / 8);
    long b = (V + (long)vpb * 8 - 1) / ((long)vpb * 8);  // ~8 passes per block
    if (b > MAX_BLOCKS) b = MAX_BLOCKS;
    if (b < 1) b = 1;
    return (int)b;
}

int check_common(const void* y, int N, long V, int C, const char* what) {
    LNN_REQUIRE(y != nullptr && lnn_aligned16(y), "%s: null/misaligned activation", what);
    LNN_REQUIRE(N > 0 && V > 0, "%s: bad dims", what);
    LNN_REQUIRE(C >= 8 && C % 8 == 0 && C <= 2048, "%s: channel count %d must be a multiple of 8 in [8, 2048]", what, C);
    return LNN_OK;
}

}  // namespace

extern "C" int lnn_instnorm_small_volume(void) { return SMALL_V; }

int lnn_launch_in_small_fwd(hipStream_t s, const void* y, void* z, int ld_z, int N, long V, int C, float eps, const float* gamma,
                            const float* beta, float slope, float* mean, float* rstd) {
    if (int e = check_common(y, N, V, C, "lnn_conv3d_fwd_in_lrelu(norm)")) return e;
    LNN_REQUIRE(V <= SMALL_V, "lnn_conv3d_fwd_in_lrelu(norm): %ld voxels per sample exceed the single-launch limit %d", V, SMALL_V);
    LNN_REQUIRE(z != nullptr && lnn_aligned16(z) && ld_z >= C && ld_z % 8 == 0, "lnn_conv3d_fwd_in_lrelu(norm): bad z / ld_z");
    LNN_REQUIRE(mean && rstd && gamma && beta, "lnn_conv3d_fwd_in_lrelu(norm): null parameter");
    const dim3 grid(lnn_cdiv(C, 32), N);
    hipLaunchKernelGGL(in_small_fwd_kernel, grid, dim3(SNT), 0, s, (const half_t*)y, (half_t*)z, ld_z, (int)V, C, eps, gamma, beta, slope,
                       mean, rstd);
    LNN_CHECK_LAUNCH("lnn_conv3d_fwd_in_lrelu(norm)");
    return LNN_OK;
}

int lnn_launch_in_small_bwd(hipStream_t s, void* y, const void* dz, int ld_dz, int N, long V, int C, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, float slope, double* ws, float* dgamma, float* dbeta, float unscale) {
    LNN_REQUIRE(V <= SMALL_V, "lnn_instnorm_lrelu_bwd(small): %ld voxels per sample exceed the single-launch limit %d", V, SMALL_V);
    hipLaunchKernelGGL(in_small_bwd_kernel, dim3(lnn_cdiv(C, 32)), dim3(SNT), 0, s, (half_t*)y, (const half_t*)dz, ld_dz, N, (int)V, C, mean,
                       rstd, gamma, beta, slope, ws, dgamma, dbeta, unscale);
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(small)");
    return LNN_OK;
}

int lnn_launch_in_stats_finalize(hipStream_t s, const float* pws, int nslots, int N, int C, long V, float eps, float* mean, float* rstd) {
    hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(lnn_cdiv(N * C, FEB)), dim3(256), 0, s, pws, nslots, N * C, V, eps, mean, rstd);
    LNN_CHECK_LAUNCH("lnn_instnorm_stats(finalize, fused partials)");
    return LNN_OK;
}

// the sums half of pass 1 over partials that a fused producer epilogue left in ws' partial region (slot rows as the statistics)
int lnn_launch_in_bwd_sums_raw(hipStream_t s, const float* pws, int nslots, int N, int C, const float* mean, const float* rstd,
                               double* ws, float* dgamma, float* dbeta, float unscale) {
    hipLaunchKernelGGL(in_lrelu_bwd_sums_kernel, dim3(lnn_cdiv(C, FEB)), dim3(256), 0, s, pws, nslots, N * C, C, ws, dgamma, dbeta,
                       unscale, mean, rstd);
    LNN_CHECK_LAUNCH("lnn_conv3d_dgrad_in_bwd_sums(sums)");
    return LNN_OK;
}

// [N*C*3 doubles: s1, s2, (unused)] [2 * MAX_BLOCKS * N * C floats: per-block partial sums]
extern "C" size_t lnn_instnorm_ws_doubles(int N, int C) { return (size_t)N * C * 3 + (size_t)MAX_BLOCKS * N * C; }

extern "C" int lnn_instnorm_stats(lnn_stream_t s_, const void* y, int N, long V, int C, float eps, float* mean, float* rstd,
                                  double* ws) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_stats")) return e;
    LNN_REQUIRE(mean && rstd && ws, "lnn_instnorm_stats: null output/workspace");
    float* pws = reinterpret_cast<float*>(ws + (size_t)N * C * 3);
    const int nblk = blocks_for(V, C);
    hipLaunchKernelGGL(in_stats_kernel, dim3(nblk, N), dim3(NT), 0, s, (const half_t*)y, V, C, pws);
    LNN_CHECK_LAUNCH("lnn_instnorm_stats");
    hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(lnn_cdiv(N * C, FEB)), dim3(256), 0, s, pws, nblk, N * C, V, eps, mean, rstd);
    LNN_CHECK_LAUNCH("lnn_instnorm_stats(finalize)");
    return LNN_OK;
}

extern "C" int lnn_instnorm_lrelu_fwd(lnn_stream_t s_, const void* y, void* z, int ld_z, int N, long V, int C,
                                      const float* mean, const float* rstd, const float* gamma, const float* beta,
                                      float slope) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_lrelu_fwd")) return e;
    LNN_REQUIRE(z != nullptr && lnn_aligned16(z) && ld_z >= C && ld_z % 8 == 0, "lnn_instnorm_lrelu_fwd: bad z / ld_z");
    LNN_REQUIRE(mean && rstd && gamma && beta, "lnn_instnorm_lrelu_fwd: null parameter");
    hipLaunchKernelGGL(in_lrelu_fwd_kernel, dim3(blocks_for(V, C), N), dim3(NT), 0, s, (const half_t*)y, (half_t*)z, ld_z,
                       V, C, mean, rstd, gamma, beta, slope, in_nt_flag());
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_fwd");
    return LNN_OK;
}

extern "C" int lnn_instnorm_lrelu_seg_fwd(lnn_stream_t s_, const void* y, void* z, int ld_z, int N, long V, int C, const float* mean,
                                          const float* rstd, const float* gamma, const float* beta, float slope,
                                          const float* seg_w, float* logits, int K) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_lrelu_seg_fwd")) return e;
    LNN_REQUIRE(z == nullptr || (lnn_aligned16(z) && ld_z >= C && ld_z % 8 == 0), "lnn_instnorm_lrelu_seg_fwd: bad z / ld_z");
    LNN_REQUIRE(mean && rstd && gamma && beta && seg_w && logits, "lnn_instnorm_lrelu_seg_fwd: null parameter");
    const int C8 = C >> 3;
    LNN_REQUIRE(K >= 1 && K <= SEG_KMAX && (C8 & (C8 - 1)) == 0 && C8 <= 64,
                "lnn_instnorm_lrelu_seg_fwd: K=%d / C=%d unsupported (K <= %d, C/8 a power of two <= 64): use lnn_instnorm_lrelu_fwd + lnn_seg1x1_fwd",
                K, C, SEG_KMAX);
    // staged logits stores: whole 16-byte pieces need V % 4 == 0 (every class plane starts aligned) and runs of >= 4 voxels
    // Measured on MI355X (C2 top level, 1.38 GB): staged 502 us vs direct 450 us -- the per-trip barrier costs more than the
    // 64-byte stores it removes; kept behind LNN_SEG_FWD_STAGED=1 for A/B, off by default.
    static int want_staged = -1;
    if (want_staged < 0) { const char* e = getenv("LNN_SEG_FWD_STAGED"); want_staged = (e && e[0] == '1') ? 1 : 0; }
    const bool staged = want_staged && C8 >= 4 && (V & 3) == 0 && lnn_aligned16(logits);
#define LNN_SF(ST, KT) hipLaunchKernelGGL((in_lrelu_seg_fwd_kernel<ST, KT>), dim3(blocks_for(V, C), N), dim3(NT), 0, s, (const half_t*)y, \
                                         (half_t*)z, ld_z, V, C, mean, rstd, gamma, beta, slope, seg_w, logits, K)
    if (staged) LNN_SF(true, 0);
    else if (K == 1) LNN_SF(false, 1);
    else if (K == 2) LNN_SF(false, 2);
    else if (K == 3) LNN_SF(false, 3);
    else if (K == 4) LNN_SF(false, 4);
    else LNN_SF(false, 0);
#undef LNN_SF
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_seg_fwd");
    return LNN_OK;
}

extern "C" int lnn_instnorm_lrelu_bwd(lnn_stream_t s_, void* y, const void* dz, int ld_dz, int N, long V, int C,
                                      const float* mean, const float* rstd, const float* gamma, const float* beta,
                                      float slope, float* dgamma, float* dbeta, float* dbias, float grad_unscale,
                                      double* ws) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_lrelu_bwd")) return e;
    LNN_REQUIRE(dz != nullptr && lnn_aligned16(dz) && ld_dz >= C && ld_dz % 8 == 0, "lnn_instnorm_lrelu_bwd: bad dz / ld_dz");
    LNN_REQUIRE(mean && rstd && gamma && beta && ws, "lnn_instnorm_lrelu_bwd: null parameter");
    if (V <= SMALL_V && !dbias && small_enabled())      // the lowest levels: reduce + sums + apply in one launch
        return lnn_launch_in_small_bwd(s, y, dz, ld_dz, N, V, C, mean, rstd, gamma, beta, slope, ws, dgamma, dbeta, grad_unscale);
    float* pws = reinterpret_cast<float*>(ws + (size_t)N * C * 3);
    const int nblk = blocks_for(V, C);
    const dim3 grid(nblk, N);
    hipLaunchKernelGGL(in_lrelu_bwd_reduce_kernel, grid, dim3(NT), 0, s, (const half_t*)y, (const half_t*)dz, ld_dz, V, C,
                       mean, rstd, gamma, beta, slope, pws, in_nt_flag());
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(reduce)");
    hipLaunchKernelGGL(in_lrelu_bwd_sums_kernel, dim3(lnn_cdiv(C, FEB)), dim3(256), 0, s, pws, nblk, N * C, C, ws, dgamma, dbeta,
                       grad_unscale, (const float*)nullptr, (const float*)nullptr);
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(sums)");
    if (dbias) {
        hipLaunchKernelGGL((in_lrelu_bwd_apply_kernel<true>), grid, dim3(NT), 0, s, (half_t*)y, (const half_t*)dz, ld_dz, V, C, mean,
                           rstd, gamma, beta, slope, ws, pws, in_nt_flag());
        LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(apply)");
        hipLaunchKernelGGL(in_lrelu_bwd_finalize_kernel, dim3(lnn_cdiv(N * C, FEB)), dim3(256), 0, s, ws, pws, nblk, N, C, dgamma,
                           dbeta, dbias, grad_unscale);
        LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(finalize)");
    } else {
        hipLaunchKernelGGL((in_lrelu_bwd_apply_kernel<false>), grid, dim3(NT), 0, s, (half_t*)y, (const half_t*)dz, ld_dz, V, C, mean,
                           rstd, gamma, beta, slope, ws, pws, in_nt_flag());
        LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(apply)");
    }
    return LNN_OK;
}

// [N*C*3 doubles: s1, s2, (unused)] [fp32 per-block partials: 2 * MAX_BLOCKS * N * C (s1, s2) + MAX_BLOCKS * N * 4 * C (d seg_w)]
extern "C" size_t lnn_instnorm_lrelu_seg_bwd_ws_doubles(int N, int C) { return (size_t)N * C * 3 + (size_t)MAX_BLOCKS * N * C * 3; }

extern "C" int lnn_instnorm_lrelu_seg_bwd(lnn_stream_t s_, void* y, const void* dz_prior, int ld_dz, const float* seg_w,
                                          const float* dlogits, float* seg_dw, int K, int N, long V, int C, const float* mean,
                                          const float* rstd, const float* gamma, const float* beta, float slope, float* dgamma,
                                          float* dbeta, float grad_unscale, double* ws) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_lrelu_seg_bwd")) return e;
    LNN_REQUIRE(dz_prior == nullptr || (lnn_aligned16(dz_prior) && ld_dz >= C && ld_dz % 8 == 0),
                "lnn_instnorm_lrelu_seg_bwd: bad dz_prior / ld_dz");
    LNN_REQUIRE(mean && rstd && gamma && beta && ws && seg_w && dlogits && seg_dw, "lnn_instnorm_lrelu_seg_bwd: null parameter");
    LNN_REQUIRE(K >= 1 && K <= 4, "lnn_instnorm_lrelu_seg_bwd: K=%d unsupported (1..4): use lnn_seg1x1_bwd + lnn_instnorm_lrelu_bwd", K);
    float* pws = reinterpret_cast<float*>(ws + (size_t)N * C * 3);
    const int nblk = blocks_for(V, C);
    const dim3 grid(nblk, N);
    const half_t* pr = (const half_t*)dz_prior;
    const int nb_in = lnn_cdiv(C, FEB);
#define LNN_SB(KT, PRIOR) do { if (C % 32 == 0) LNN_SBQ(KT, PRIOR, true); else LNN_SBQ(KT, PRIOR, false); } while (0)
#define LNN_SBQ(KT, PRIOR, QUAD)                                                                                                     \
    do {                                                                                                                             \
        hipLaunchKernelGGL((in_lrelu_seg_bwd_reduce_kernel<KT, PRIOR, QUAD>), grid, dim3(NT), 0, s, (const half_t*)y, pr, ld_dz, V, C, \
                           mean, rstd, gamma, beta, slope, seg_w, dlogits, pws);                                                     \
        LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_seg_bwd(reduce)");                                                                      \
        hipLaunchKernelGGL(in_lrelu_seg_bwd_sums_kernel, dim3(nb_in + lnn_cdiv(K * C, FEB)), dim3(256), 0, s, pws, nblk, N, C, K,     \
                           nb_in, ws, dgamma, dbeta, seg_dw, grad_unscale);                                                          \
        LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_seg_bwd(sums)");                                                                        \
        hipLaunchKernelGGL((in_lrelu_seg_bwd_apply_kernel<KT, PRIOR, QUAD>), grid, dim3(NT), 0, s, (half_t*)y, pr, ld_dz, V, C, mean, \
                           rstd, gamma, beta, slope, seg_w, dlogits, ws);                                                            \
        LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_seg_bwd(apply)");                                                                       \
    } while (0)
    if (pr) {
        if (K == 1) LNN_SB(1, true); else if (K == 2) LNN_SB(2, true); else if (K == 3) LNN_SB(3, true); else LNN_SB(4, true);
    } else {
        if (K == 1) LNN_SB(1, false); else if (K == 2) LNN_SB(2, false); else if (K == 3) LNN_SB(3, false); else LNN_SB(4, false);
    }
#undef LNN_SB
#undef LNN_SBQ
    return LNN_OK;
}

// Pass 2 of the backward alone: dy = gamma rstd (g - s1/V - xhat s2/V) in place over y, from the sums a pass 1 left in ws
// (lnn_instnorm_lrelu_bwd_sums or lnn_conv3d_dgrad_in_bwd_sums).
extern "C" int lnn_instnorm_lrelu_bwd_apply(lnn_stream_t s_, void* y, const void* dz, int ld_dz, int N, long V, int C,
                                            const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                                            double* ws) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_lrelu_bwd_apply")) return e;
    LNN_REQUIRE(dz != nullptr && lnn_aligned16(dz) && ld_dz >= C && ld_dz % 8 == 0, "lnn_instnorm_lrelu_bwd_apply: bad dz / ld_dz");
    LNN_REQUIRE(mean && rstd && gamma && beta && ws, "lnn_instnorm_lrelu_bwd_apply: null parameter");
    hipLaunchKernelGGL((in_lrelu_bwd_apply_kernel<false>), dim3(blocks_for(V, C), N), dim3(NT), 0, s, (half_t*)y, (const half_t*)dz, ld_dz,
                       V, C, mean, rstd, gamma, beta, slope, ws, (float*)nullptr, in_nt_flag());
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd_apply");
    return LNN_OK;
}

// Pass 1 of the backward alone (reduce + sums): ws[(n * C + c) * 3 + {0, 1}] = sum g, sum g * xhat and the affine gradients; y is
// not touched.  For the first block, whose pass 2 lives inside its weight gradient (lnn_conv3d_wgrad_c1_in_bwd).
extern "C" int lnn_instnorm_lrelu_bwd_sums(lnn_stream_t s_, const void* y, const void* dz, int ld_dz, int N, long V, int C,
                                           const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                                           float* dgamma, float* dbeta, float grad_unscale, double* ws) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_lrelu_bwd_sums")) return e;
    LNN_REQUIRE(dz != nullptr && lnn_aligned16(dz) && ld_dz >= C && ld_dz % 8 == 0, "lnn_instnorm_lrelu_bwd_sums: bad dz / ld_dz");
    LNN_REQUIRE(mean && rstd && gamma && beta && ws, "lnn_instnorm_lrelu_bwd_sums: null parameter");
    float* pws = reinterpret_cast<float*>(ws + (size_t)N * C * 3);
    const int nblk = blocks_for(V, C);
    hipLaunchKernelGGL(in_lrelu_bwd_reduce_kernel, dim3(nblk, N), dim3(NT), 0, s, (const half_t*)y, (const half_t*)dz, ld_dz, V, C,
                       mean, rstd, gamma, beta, slope, pws, in_nt_flag());
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd_sums(reduce)");
    hipLaunchKernelGGL(in_lrelu_bwd_sums_kernel, dim3(lnn_cdiv(C, FEB)), dim3(256), 0, s, pws, nblk, N * C, C, ws, dgamma, dbeta,
                       grad_unscale, (const float*)nullptr, (const float*)nullptr);
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd_sums(sums)");
    return LNN_OK;
}
