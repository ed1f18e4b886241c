// InstanceNorm3d(affine) + LeakyReLU, forward and backward, channels-last fp16 (HBM-bound kernels).
//
// Thread layout: a 256-thread block covers VPB = 256 / (C/8) voxels per pass; thread = (voxel lane,
// 8-channel vector) so that consecutive threads touch consecutive 16-byte vectors (fully coalesced)
// and every thread keeps a FIXED channel octet -> per-channel partial sums live in registers.
// Per-(n,c) reductions: fp32 in registers over a short strided run, fp64 atomics across blocks.
#include "lnn_common.h"

namespace {

constexpr int NT = 256;

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

__global__ __launch_bounds__(NT) void in_stats_kernel(const half_t* __restrict__ y, long V, int C, double* ws) {
    __shared__ float red[NT * 17];
    const RowMap rm = row_map(C);
    const int n = blockIdx.y;
    const half_t* yn = y + (long)n * V * C;
    float part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) part[0][e] = part[1][e] = 0.f;
    if (rm.active) {
        for (long v = vrange_begin(V, rm) + rm.vl; v < vrange_end(V, rm); v += rm.VPB) {
            const half8 x = *reinterpret_cast<const half8*>(yn + v * C + rm.c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)x[e];
                part[0][e] += f;
                part[1][e] += f * f;
            }
        }
    }
    block_channel_reduce<2>(rm, C, part, red, [&](int a, int c, float s) {
        atomicAdd(ws + ((long)n * C + c) * 2 + a, (double)s);
    });
}

__global__ void in_stats_finalize_kernel(const double* ws, int NC, long V, float eps, float* mean, float* rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    const double m = ws[i * 2] / (double)V;
    double var = ws[i * 2 + 1] / (double)V - m * m;
    if (var < 0) var = 0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(NT) void in_lrelu_fwd_kernel(const half_t* __restrict__ y, half_t* __restrict__ z, int ld_z,
                                                          long V, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float slope) {
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
    for (long v = vrange_begin(V, rm) + rm.vl; v < vrange_end(V, rm); v += rm.VPB) {
        const half8 x = *reinterpret_cast<const half8*>(yn + v * C + rm.c8 * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = (float)x[e] * sc[e] + sh[e];
            o[e] = (half_t)(t > 0.f ? t : t * slope);
        }
        *reinterpret_cast<half8*>(zn + v * ld_z + rm.c8 * 8) = o;
    }
}

// pass 1 of backward: s1 = sum g, s2 = sum g*xhat with g = dz * lrelu'(gamma*xhat+beta)
__global__ __launch_bounds__(NT) void in_lrelu_bwd_reduce_kernel(const half_t* __restrict__ y, const half_t* __restrict__ dz,
                                                                 int ld_dz, long V, int C, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float slope, double* ws) {
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
        for (long v = vrange_begin(V, rm) + rm.vl; v < vrange_end(V, rm); v += rm.VPB) {
            const half8 x = *reinterpret_cast<const half8*>(yn + v * C + rm.c8 * 8);
            const half8 d = *reinterpret_cast<const half8*>(dzn + v * ld_dz + rm.c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)x[e] - mu[e]) * rs[e];
                const float pre = ga[e] * xh + be[e];
                const float g = (float)d[e] * (pre > 0.f ? 1.f : slope);
                part[0][e] += g;
                part[1][e] += g * xh;
            }
        }
    }
    block_channel_reduce<2>(rm, C, part, red, [&](int a, int c, float s) {
        atomicAdd(ws + ((long)n * C + c) * 3 + a, (double)s);
    });
}

// pass 2: dy = gamma*rstd*(g - s1/V - xhat*s2/V), in place over y; db partial = sum dy
__global__ __launch_bounds__(NT) void in_lrelu_bwd_apply_kernel(half_t* __restrict__ y, const half_t* __restrict__ dz, int ld_dz,
                                                                long V, int C, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float slope, double* ws) {
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
        for (long v = vrange_begin(V, rm) + rm.vl; v < vrange_end(V, rm); v += rm.VPB) {
            half8* yp = reinterpret_cast<half8*>(yn + v * C + rm.c8 * 8);
            const half8 x = *yp;
            const half8 d = *reinterpret_cast<const half8*>(dzn + v * ld_dz + rm.c8 * 8);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)x[e] - mu[e]) * rs[e];
                const float pre = ga[e] * xh + be[e];
                const float g = (float)d[e] * (pre > 0.f ? 1.f : slope);
                const half_t r = (half_t)(ga[e] * rs[e] * (g - m1[e] - xh * m2[e]));
                o[e] = r;
                part[0][e] += (float)r;
            }
            *yp = o;
        }
    }
    block_channel_reduce<1>(rm, C, part, red, [&](int, int c, float s) {
        atomicAdd(ws + ((long)n * C + c) * 3 + 2, (double)s);
    });
}

__global__ void in_lrelu_bwd_finalize_kernel(const double* ws, int N, int C, float* dgamma, float* dbeta, float* dbias,
                                             float unscale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0, s2 = 0, db = 0;
    for (int n = 0; n < N; ++n) {
        s1 += ws[((long)n * C + c) * 3 + 0];
        s2 += ws[((long)n * C + c) * 3 + 1];
        db += ws[((long)n * C + c) * 3 + 2];
    }
    // atomics: two sample lanes (HIP streams) may finalise the same layer concurrently
    if (dgamma) atomicAdd(dgamma + c, (float)(s2 * unscale));
    if (dbeta) atomicAdd(dbeta + c, (float)(s1 * unscale));
    if (dbias) atomicAdd(dbias + c, (float)(db * unscale));
}

int blocks_for(long V, int C) {
    const int vpb = NT / (C / 8);
    long b = (V + (long)vpb * 8 - 1) / ((long)vpb * 8);  // ~8 passes per block
    if (b > 1024) b = 1024;
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

extern "C" size_t lnn_instnorm_ws_doubles(int N, int C) { return (size_t)N * C * 3; }

extern "C" int lnn_instnorm_stats(lnn_stream_t s_, const void* y, int N, long V, int C, float eps, float* mean, float* rstd,
                                  double* ws) {
    hipStream_t s = (hipStream_t)s_;
    if (int e = check_common(y, N, V, C, "lnn_instnorm_stats")) return e;
    LNN_REQUIRE(mean && rstd && ws, "lnn_instnorm_stats: null output/workspace");
    hipMemsetAsync(ws, 0, sizeof(double) * 2 * N * C, s);
    hipLaunchKernelGGL(in_stats_kernel, dim3(blocks_for(V, C), N), dim3(NT), 0, s, (const half_t*)y, V, C, ws);
    LNN_CHECK_LAUNCH("lnn_instnorm_stats");
    hipLaunchKernelGGL(in_stats_finalize_kernel, dim3(lnn_cdiv(N * C, 256)), dim3(256), 0, s, ws, N * C, V, eps, mean, rstd);
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
                       V, C, mean, rstd, gamma, beta, slope);
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_fwd");
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
    hipMemsetAsync(ws, 0, sizeof(double) * 3 * N * C, s);
    const dim3 grid(blocks_for(V, C), N);
    hipLaunchKernelGGL(in_lrelu_bwd_reduce_kernel, grid, dim3(NT), 0, s, (const half_t*)y, (const half_t*)dz, ld_dz, V, C,
                       mean, rstd, gamma, beta, slope, ws);
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(reduce)");
    hipLaunchKernelGGL(in_lrelu_bwd_apply_kernel, grid, dim3(NT), 0, s, (half_t*)y, (const half_t*)dz, ld_dz, V, C, mean,
                       rstd, gamma, beta, slope, ws);
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(apply)");
    hipLaunchKernelGGL(in_lrelu_bwd_finalize_kernel, dim3(lnn_cdiv(C, 256)), dim3(256), 0, s, ws, N, C, dgamma, dbeta, dbias,
                       grad_unscale);
    LNN_CHECK_LAUNCH("lnn_instnorm_lrelu_bwd(finalize)");
    return LNN_OK;
}
