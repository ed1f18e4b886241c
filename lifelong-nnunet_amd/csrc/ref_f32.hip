// fp32-STORAGE path of the U-Net (the reference's ``fp16=False`` branch, multihead/nnUNetTrainerMultiHead.py:632-641,
// CLI ``--fp32`` run_training.py:71): every activation, gradient and weight stays fp32, long reductions accumulate in
// fp64 in a fixed order (bit-reproducible).  Direct (non-MFMA) kernels: this path exists for PARITY -- it reproduces the
// reference's CPU/fp32 arithmetic to round-off so that Fisher values, update vectors and multi-iteration losses can be
// asserted at 1e-4 and below -- not for speed (about 30x slower than the fp16-storage MFMA path).
// Layouts: activations NDHWC fp32 with channel stride ``ld`` (views into wider buffers allowed, as in the fp16 path);
// weights are read straight from the parameter arena in PyTorch's own layouts (Conv3d (K,C,3,3,3), ConvTranspose3d
// (Cin,Cout,2,2,2), 1x1x1 (K,C)); logits cross the loss boundary as fp32 NCDHW like everywhere else.
// Ops replaced: nn.Conv3d / nn.ConvTranspose3d / nn.InstanceNorm3d + nn.LeakyReLU of the module tree at
// test/network_architecture/test_MultiHead_Module.py:281-433 (forward at generic_ViT_UNet.py:222-230,261-286).
#include "lnn_common.h"

namespace {

constexpr int NT = 256;

__host__ int blocks_for_elems(long n) {
    long b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 65535L * 16 ? 65535L * 16 : b));
}

// per-axis geometry (round 4): kernel extents k* in {1, 3} (padding k / 2) and strides s* in {1, 2}; transposed convolutions
// have kernel == stride.  The isotropic entry points pass {3,3,3, s,s,s} / {2,2,2, 2,2,2}.
struct F32Geo { int kz, ky, kx, sz, sy, sx; };

// y[n, o, k] = bias[k] + sum_{tap, c} x[n, o*stride + tap - pad, c] * w[k, c, tap]
__global__ __launch_bounds__(NT) void f32_conv_fwd_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int ld_y, int N,
                                                          int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int K, F32Geo g) {
    const long total = (long)N * Do * Ho * Wo * K;
    const int T = g.kz * g.ky * g.kx;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int k = (int)(e % K);
        long v = e / K;
        const int ox = (int)(v % Wo); v /= Wo;
        const int oy = (int)(v % Ho); v /= Ho;
        const int oz = (int)(v % Do);
        const int n = (int)(v / Do);
        double acc = bias ? (double)bias[k] : 0.0;
        for (int dz = 0; dz < g.kz; ++dz) {
            const int iz = oz * g.sz + dz - g.kz / 2;
            if ((unsigned)iz >= (unsigned)Di) continue;
            for (int dy = 0; dy < g.ky; ++dy) {
                const int iy = oy * g.sy + dy - g.ky / 2;
                if ((unsigned)iy >= (unsigned)Hi) continue;
                for (int dx = 0; dx < g.kx; ++dx) {
                    const int ix = ox * g.sx + dx - g.kx / 2;
                    if ((unsigned)ix >= (unsigned)Wi) continue;
                    const float* xp = x + ((((long)n * Di + iz) * Hi + iy) * Wi + ix) * ld_x;
                    const float* wp = w + (long)k * C * T + ((dz * g.ky + dy) * g.kx + dx);
                    for (int c = 0; c < C; ++c) acc += (double)xp[c] * (double)wp[(long)c * T];
                }
            }
        }
        y[((((long)n * Do + oz) * Ho + oy) * Wo + ox) * ld_y + k] = (float)acc;
    }
}

// dx[n, i, c] (+)= sum_{tap, k} dy[n, o, k] * w[k, c, tap]  with  o * stride + tap - pad == i
__global__ __launch_bounds__(NT) void f32_conv_dgrad_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int ld_dx, int N, int Di, int Hi, int Wi, int Do,
                                                            int Ho, int Wo, int C, int K, F32Geo g, int accumulate) {
    const long total = (long)N * Di * Hi * Wi * C;
    const int T = g.kz * g.ky * g.kx;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int c = (int)(e % C);
        long v = e / C;
        const int ix = (int)(v % Wi); v /= Wi;
        const int iy = (int)(v % Hi); v /= Hi;
        const int iz = (int)(v % Di);
        const int n = (int)(v / Di);
        double acc = 0.0;
        for (int dz = 0; dz < g.kz; ++dz) {
            const int tz = iz + g.kz / 2 - dz;
            if (tz < 0 || tz % g.sz != 0 || tz / g.sz >= Do) continue;
            for (int dyy = 0; dyy < g.ky; ++dyy) {
                const int ty = iy + g.ky / 2 - dyy;
                if (ty < 0 || ty % g.sy != 0 || ty / g.sy >= Ho) continue;
                for (int dxx = 0; dxx < g.kx; ++dxx) {
                    const int tx = ix + g.kx / 2 - dxx;
                    if (tx < 0 || tx % g.sx != 0 || tx / g.sx >= Wo) continue;
                    const float* gp = dy + ((((long)n * Do + tz / g.sz) * Ho + ty / g.sy) * Wo + tx / g.sx) * ld_dy;
                    const float* wp = w + (long)c * T + ((dz * g.ky + dyy) * g.kx + dxx);
                    for (int k = 0; k < K; ++k) acc += (double)gp[k] * (double)wp[(long)k * C * T];
                }
            }
        }
        float* o = dx + ((((long)n * Di + iz) * Hi + iy) * Wi + ix) * ld_dx + c;
        *o = accumulate ? *o + (float)acc : (float)acc;
    }
}

__device__ __forceinline__ double block_sum_d1(double v, double* sm) {       // fixed-order sum over the 256 threads
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// dw[k, c, tap] += sum_{n, o} dy[n, o, k] * x[n, o*stride + tap - pad, c]       (one block per (k, c, tap))
__global__ __launch_bounds__(NT) void f32_conv_wgrad_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ dy,
                                                            int ld_dy, float* __restrict__ dw, int N, int Di, int Hi, int Wi, int Do,
                                                            int Ho, int Wo, int C, int K, F32Geo g) {
    __shared__ double sm[4];
    const int T = g.kz * g.ky * g.kx;
    const int tap = blockIdx.x % T, c = (blockIdx.x / T) % C, k = blockIdx.x / (T * C);
    const int dz = tap / (g.ky * g.kx), dyy = (tap / g.kx) % g.ky, dxx = tap % g.kx;
    const long total = (long)N * Do * Ho * Wo;
    double acc = 0.0;
    for (long v0 = threadIdx.x; v0 < total; v0 += NT) {
        long v = v0;
        const int ox = (int)(v % Wo); v /= Wo;
        const int oy = (int)(v % Ho); v /= Ho;
        const int oz = (int)(v % Do);
        const int n = (int)(v / Do);
        const int iz = oz * g.sz + dz - g.kz / 2, iy = oy * g.sy + dyy - g.ky / 2, ix = ox * g.sx + dxx - g.kx / 2;
        if ((unsigned)iz >= (unsigned)Di || (unsigned)iy >= (unsigned)Hi || (unsigned)ix >= (unsigned)Wi) continue;
        acc += (double)dy[v0 * ld_dy + k] * (double)x[((((long)n * Di + iz) * Hi + iy) * Wi + ix) * ld_x + c];
    }
    const double s = block_sum_d1(acc, sm);
    if (threadIdx.x == 0) dw[((long)k * C + c) * T + tap] += (float)s;
}

// ConvTranspose3d kernel == stride: y[n, s*i + a, k] = sum_c x[n, i, c] * w[c, k, a]     (a = the offset inside the s-cell)
__global__ __launch_bounds__(NT) void f32_convT_fwd_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ w,
                                                           float* __restrict__ y, int ld_y, int N, int D, int H, int W, int C, int K,
                                                           F32Geo g) {
    const int T = g.sz * g.sy * g.sx, Do = D * g.sz, Ho = H * g.sy, Wo = W * g.sx;
    const long total = (long)N * Do * Ho * Wo * K;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int k = (int)(e % K);
        long v = e / K;
        const int ox = (int)(v % Wo); v /= Wo;
        const int oy = (int)(v % Ho); v /= Ho;
        const int oz = (int)(v % Do);
        const int n = (int)(v / Do);
        const int a = ((oz % g.sz) * g.sy + (oy % g.sy)) * g.sx + (ox % g.sx);
        const float* xp = x + ((((long)n * D + oz / g.sz) * H + oy / g.sy) * W + ox / g.sx) * ld_x;
        const float* wp = w + (long)k * T + a;
        double acc = 0.0;
        for (int c = 0; c < C; ++c) acc += (double)xp[c] * (double)wp[(long)c * K * T];
        y[((((long)n * Do + oz) * Ho + oy) * Wo + ox) * ld_y + k] = (float)acc;
    }
}

// dx[n, i, c] = sum_{a, k} dy[n, s*i + a, k] * w[c, k, a]
__global__ __launch_bounds__(NT) void f32_convT_dgrad_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ w,
                                                             float* __restrict__ dx, int ld_dx, int N, int D, int H, int W, int C, int K,
                                                             F32Geo g, int accumulate) {
    const int T = g.sz * g.sy * g.sx, Ho = H * g.sy, Wo = W * g.sx, Do = D * g.sz;
    const long total = (long)N * D * H * W * C;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int c = (int)(e % C);
        long v = e / C;
        const int ix = (int)(v % W); v /= W;
        const int iy = (int)(v % H); v /= H;
        const int iz = (int)(v % D);
        const int n = (int)(v / D);
        double acc = 0.0;
        for (int a = 0; a < T; ++a) {
            const int oz = g.sz * iz + a / (g.sy * g.sx), oy = g.sy * iy + (a / g.sx) % g.sy, ox = g.sx * ix + a % g.sx;
            const float* gp = dy + ((((long)n * Do + oz) * Ho + oy) * Wo + ox) * ld_dy;
            const float* wp = w + (long)c * K * T + a;
            for (int k = 0; k < K; ++k) acc += (double)gp[k] * (double)wp[(long)k * T];
        }
        float* o = dx + e / C * ld_dx + c;
        *o = accumulate ? *o + (float)acc : (float)acc;
    }
}

// dw[c, k, a] += sum_{n, i} x[n, i, c] * dy[n, s*i + a, k]       (one block per (c, k, a))
__global__ __launch_bounds__(NT) void f32_convT_wgrad_kernel(const float* __restrict__ x, int ld_x, const float* __restrict__ dy,
                                                             int ld_dy, float* __restrict__ dw, int N, int D, int H, int W, int C, int K,
                                                             F32Geo g) {
    __shared__ double sm[4];
    const int T = g.sz * g.sy * g.sx, Ho = H * g.sy, Wo = W * g.sx, Do = D * g.sz;
    const int a = blockIdx.x % T, k = (blockIdx.x / T) % K, c = blockIdx.x / (T * K);
    const long total = (long)N * D * H * W;
    double acc = 0.0;
    for (long v0 = threadIdx.x; v0 < total; v0 += NT) {
        long v = v0;
        const int ix = (int)(v % W); v /= W;
        const int iy = (int)(v % H); v /= H;
        const int iz = (int)(v % D);
        const int n = (int)(v / D);
        const int oz = g.sz * iz + a / (g.sy * g.sx), oy = g.sy * iy + (a / g.sx) % g.sy, ox = g.sx * ix + a % g.sx;
        acc += (double)x[v0 * ld_x + c] * (double)dy[((((long)n * Do + oz) * Ho + oy) * Wo + ox) * ld_dy + k];
    }
    const double s = block_sum_d1(acc, sm);
    if (threadIdx.x == 0) dw[((long)c * K + k) * T + a] += (float)s;
}

// NCDHW fp32 image -> channels-last (stride ld, channels beyond C left untouched): fp32 copy / fp16 cast of a multi-channel input
template <typename TO>
__global__ __launch_bounds__(NT) void ncdhw_to_cl_kernel(const float* __restrict__ src, TO* __restrict__ dst, int N, int C, long V, int ld) {
    const long total = (long)N * V;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const long n = e / V, v = e % V;
        for (int c = 0; c < C; ++c) dst[e * ld + c] = (TO)src[(n * C + c) * V + v];
    }
}

// InstanceNorm statistics: one block per (n, c): mean, rstd = 1 / sqrt(biased var + eps)
__global__ __launch_bounds__(NT) void f32_in_stats_kernel(const float* __restrict__ y, int ld_y, long V, int C, float eps,
                                                          float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ double sm[4];
    const int c = blockIdx.x % C, n = blockIdx.x / C;
    const float* yp = y + (long)n * V * ld_y + c;
    double s = 0.0, s2 = 0.0;
    for (long v = threadIdx.x; v < V; v += NT) { const double t = yp[v * ld_y]; s += t; s2 += t * t; }
    const double S = block_sum_d1(s, sm);
    const double S2 = block_sum_d1(s2, sm);
    if (threadIdx.x == 0) {
        const double m = S / (double)V;
        double var = S2 / (double)V - m * m;
        if (var < 0) var = 0;
        mean[blockIdx.x] = (float)m;
        rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ __launch_bounds__(NT) void f32_in_lrelu_fwd_kernel(const float* __restrict__ y, int ld_y, float* __restrict__ z, int ld_z,
                                                              int N, long V, int C, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float slope) {
    const long total = (long)N * V * C;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int c = (int)(e % C);
        const long nv = e / C;
        const int n = (int)(nv / V);
        const float t = (y[nv * ld_y + c] - mean[n * C + c]) * rstd[n * C + c] * gamma[c] + beta[c];
        z[nv * ld_z + c] = t > 0.f ? t : t * slope;
    }
}

// backward, pass 1 (one block per (n, c)): g = dz * lrelu'(pre-activation);  sums of g and g * xhat
__global__ __launch_bounds__(NT) void f32_in_bwd_reduce_kernel(const float* __restrict__ y, int ld_y, const float* __restrict__ dz,
                                                               int ld_dz, long V, int C, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float slope, double* __restrict__ sums) {
    __shared__ double sm[4];
    const int c = blockIdx.x % C, n = blockIdx.x / C;
    const float m = mean[blockIdx.x], r = rstd[blockIdx.x], ga = gamma[c], be = beta[c];
    double s = 0.0, sx = 0.0;
    for (long v = threadIdx.x; v < V; v += NT) {
        const long nv = (long)n * V + v;
        const float xh = (y[nv * ld_y + c] - m) * r;
        const float g = dz[nv * ld_dz + c] * ((xh * ga + be) > 0.f ? 1.f : slope);
        s += g; sx += (double)g * xh;
    }
    const double S = block_sum_d1(s, sm);
    const double SX = block_sum_d1(sx, sm);
    if (threadIdx.x == 0) { sums[(long)blockIdx.x * 2] = S; sums[(long)blockIdx.x * 2 + 1] = SX; }
}

// pass 2: dy = rstd * gamma * (g - mean(g) - xhat * mean(g * xhat)), written IN PLACE over y (as the fp16 path does)
__global__ __launch_bounds__(NT) void f32_in_bwd_apply_kernel(float* __restrict__ y, int ld_y, const float* __restrict__ dz, int ld_dz,
                                                              int N, long V, int C, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float slope,
                                                              const double* __restrict__ sums) {
    const long total = (long)N * V * C;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int c = (int)(e % C);
        const long nv = e / C;
        const int n = (int)(nv / V);
        const int i = n * C + c;
        const float xh = (y[nv * ld_y + c] - mean[i]) * rstd[i];
        const float g = dz[nv * ld_dz + c] * ((xh * gamma[c] + beta[c]) > 0.f ? 1.f : slope);
        const double mg = sums[(long)i * 2] / (double)V, mgx = sums[(long)i * 2 + 1] / (double)V;
        y[nv * ld_y + c] = (float)((double)rstd[i] * gamma[c] * ((double)g - mg - (double)xh * mgx));
    }
}

// pass 3 (one thread per channel): dgamma += sum_n S_gx, dbeta += sum_n S_g; the conv bias gradient is the sum of dy,
// which InstanceNorm makes exactly zero analytically: gamma * rstd * (S_g - V * S_g / V - S_gx * sum(xhat) / V) with sum(xhat) = 0
__global__ void f32_in_bwd_params_kernel(const double* __restrict__ sums, int N, int C, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sg = 0.0, sgx = 0.0;
    for (int n = 0; n < N; ++n) { sg += sums[((long)n * C + c) * 2]; sgx += sums[((long)n * C + c) * 2 + 1]; }
    if (dgamma) dgamma[c] += (float)sgx;
    if (dbeta) dbeta[c] += (float)sg;
}

// seg head: logits[n, k, v] = sum_c z[n, v, c] * w[k, c]
__global__ __launch_bounds__(NT) void f32_seg_fwd_kernel(const float* __restrict__ z, int ld_z, const float* __restrict__ w,
                                                         float* __restrict__ logits, int N, long V, int C, int K) {
    const long total = (long)N * K * V;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const long v = e % V;
        const int k = (int)((e / V) % K), n = (int)(e / (V * K));
        const float* zp = z + ((long)n * V + v) * ld_z;
        double acc = 0.0;
        for (int c = 0; c < C; ++c) acc += (double)zp[c] * (double)w[k * C + c];
        logits[e] = (float)acc;
    }
}

__global__ __launch_bounds__(NT) void f32_seg_dgrad_kernel(const float* __restrict__ dl, const float* __restrict__ w,
                                                           float* __restrict__ gz, int ld_gz, int N, long V, int C, int K, int accumulate) {
    const long total = (long)N * V * C;
    for (long e = (long)blockIdx.x * NT + threadIdx.x; e < total; e += (long)gridDim.x * NT) {
        const int c = (int)(e % C);
        const long nv = e / C;
        const long v = nv % V;
        const int n = (int)(nv / V);
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += (double)dl[((long)n * K + k) * V + v] * (double)w[k * C + c];
        float* o = gz + nv * ld_gz + c;
        *o = accumulate ? *o + (float)acc : (float)acc;
    }
}

// dw[k, c] += sum_{n, v} dl[n, k, v] * z[n, v, c]      (one block per (k, c))
__global__ __launch_bounds__(NT) void f32_seg_wgrad_kernel(const float* __restrict__ z, int ld_z, const float* __restrict__ dl,
                                                           float* __restrict__ dw, int N, long V, int C, int K) {
    __shared__ double sm[4];
    const int c = blockIdx.x % C, k = blockIdx.x / C;
    double acc = 0.0;
    for (long nv = threadIdx.x; nv < (long)N * V; nv += NT) {
        const int n = (int)(nv / V);
        const long v = nv % V;
        acc += (double)dl[((long)n * K + k) * V + v] * (double)z[nv * ld_z + c];
    }
    const double s = block_sum_d1(acc, sm);
    if (threadIdx.x == 0) dw[k * C + c] += (float)s;
}

}  // namespace

#define F32_REQ(c, ...) LNN_REQUIRE(c, __VA_ARGS__)

namespace {
int f32_check_geo(const F32Geo& g, bool transposed, const char* what) {
    const int k[3] = {g.kz, g.ky, g.kx}, st[3] = {g.sz, g.sy, g.sx};
    for (int a = 0; a < 3; ++a) {
        F32_REQ(st[a] == 1 || st[a] == 2, "%s: stride %d unsupported", what, st[a]);
        if (transposed) F32_REQ(k[a] == st[a], "%s: kernel must equal stride", what);
        else F32_REQ(k[a] == 1 || k[a] == 3, "%s: kernel extent %d unsupported", what, k[a]);
    }
    return LNN_OK;
}
}  // namespace

extern "C" int lnn_f32_conv3d_fwd_g(lnn_stream_t s_, const float* x, int ld_x, const float* w, const float* bias, float* y, int ld_y, int N,
                                    int Di, int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx) {
    const F32Geo g{kz, ky, kx, sz, sy, sx};
    if (int e = f32_check_geo(g, false, "lnn_f32_conv3d_fwd")) return e;
    F32_REQ(x && w && y && ld_x >= C && ld_y >= K, "lnn_f32_conv3d_fwd: bad arguments");
    const int Do = (Di - 1) / sz + 1, Ho = (Hi - 1) / sy + 1, Wo = (Wi - 1) / sx + 1;
    hipLaunchKernelGGL(f32_conv_fwd_kernel, dim3(blocks_for_elems((long)N * Do * Ho * Wo * K)), dim3(NT), 0, (hipStream_t)s_, x, ld_x,
                       w, bias, y, ld_y, N, Di, Hi, Wi, Do, Ho, Wo, C, K, g);
    LNN_CHECK_LAUNCH("lnn_f32_conv3d_fwd");
    return LNN_OK;
}
extern "C" int lnn_f32_conv3d_fwd(lnn_stream_t s, const float* x, int ld_x, const float* w, const float* bias, float* y, int ld_y, int N,
                                  int Di, int Hi, int Wi, int C, int K, int stride) {
    return lnn_f32_conv3d_fwd_g(s, x, ld_x, w, bias, y, ld_y, N, Di, Hi, Wi, C, K, 3, 3, 3, stride, stride, stride);
}

extern "C" int lnn_f32_conv3d_dgrad_g(lnn_stream_t s_, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N, int Di,
                                      int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx, int accumulate) {
    const F32Geo g{kz, ky, kx, sz, sy, sx};
    if (int e = f32_check_geo(g, false, "lnn_f32_conv3d_dgrad")) return e;
    F32_REQ(dy && w && dx && ld_dy >= K && ld_dx >= C, "lnn_f32_conv3d_dgrad: bad arguments");
    const int Do = (Di - 1) / sz + 1, Ho = (Hi - 1) / sy + 1, Wo = (Wi - 1) / sx + 1;
    hipLaunchKernelGGL(f32_conv_dgrad_kernel, dim3(blocks_for_elems((long)N * Di * Hi * Wi * C)), dim3(NT), 0, (hipStream_t)s_, dy,
                       ld_dy, w, dx, ld_dx, N, Di, Hi, Wi, Do, Ho, Wo, C, K, g, accumulate);
    LNN_CHECK_LAUNCH("lnn_f32_conv3d_dgrad");
    return LNN_OK;
}
extern "C" int lnn_f32_conv3d_dgrad(lnn_stream_t s, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N, int Di,
                                    int Hi, int Wi, int C, int K, int stride, int accumulate) {
    return lnn_f32_conv3d_dgrad_g(s, dy, ld_dy, w, dx, ld_dx, N, Di, Hi, Wi, C, K, 3, 3, 3, stride, stride, stride, accumulate);
}

extern "C" int lnn_f32_conv3d_wgrad_g(lnn_stream_t s_, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N, int Di,
                                      int Hi, int Wi, int C, int K, int kz, int ky, int kx, int sz, int sy, int sx) {
    const F32Geo g{kz, ky, kx, sz, sy, sx};
    if (int e = f32_check_geo(g, false, "lnn_f32_conv3d_wgrad")) return e;
    F32_REQ(x && dy && dw, "lnn_f32_conv3d_wgrad: bad arguments");
    const int Do = (Di - 1) / sz + 1, Ho = (Hi - 1) / sy + 1, Wo = (Wi - 1) / sx + 1;
    hipLaunchKernelGGL(f32_conv_wgrad_kernel, dim3(K * C * kz * ky * kx), dim3(NT), 0, (hipStream_t)s_, x, ld_x, dy, ld_dy, dw, N, Di, Hi,
                       Wi, Do, Ho, Wo, C, K, g);
    LNN_CHECK_LAUNCH("lnn_f32_conv3d_wgrad");
    return LNN_OK;
}
extern "C" int lnn_f32_conv3d_wgrad(lnn_stream_t s, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N, int Di,
                                    int Hi, int Wi, int C, int K, int stride) {
    return lnn_f32_conv3d_wgrad_g(s, x, ld_x, dy, ld_dy, dw, N, Di, Hi, Wi, C, K, 3, 3, 3, stride, stride, stride);
}

extern "C" int lnn_f32_convT3d_fwd_g(lnn_stream_t s_, const float* x, int ld_x, const float* w, float* y, int ld_y, int N, int D,
                                     int H, int W, int C, int K, int sz, int sy, int sx) {
    const F32Geo g{sz, sy, sx, sz, sy, sx};
    if (int e = f32_check_geo(g, true, "lnn_f32_convT3d_fwd")) return e;
    F32_REQ(x && w && y && ld_x >= C && ld_y >= K, "lnn_f32_convT3d_fwd: bad arguments");
    hipLaunchKernelGGL(f32_convT_fwd_kernel, dim3(blocks_for_elems((long)N * sz * sy * sx * D * H * W * K)), dim3(NT), 0, (hipStream_t)s_, x,
                       ld_x, w, y, ld_y, N, D, H, W, C, K, g);
    LNN_CHECK_LAUNCH("lnn_f32_convT3d_fwd");
    return LNN_OK;
}
extern "C" int lnn_f32_convT3d_k2s2_fwd(lnn_stream_t s, const float* x, int ld_x, const float* w, float* y, int ld_y, int N, int D,
                                        int H, int W, int C, int K) {
    return lnn_f32_convT3d_fwd_g(s, x, ld_x, w, y, ld_y, N, D, H, W, C, K, 2, 2, 2);
}

extern "C" int lnn_f32_convT3d_dgrad_g(lnn_stream_t s_, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N,
                                       int D, int H, int W, int C, int K, int sz, int sy, int sx, int accumulate) {
    const F32Geo g{sz, sy, sx, sz, sy, sx};
    if (int e = f32_check_geo(g, true, "lnn_f32_convT3d_dgrad")) return e;
    F32_REQ(dy && w && dx, "lnn_f32_convT3d_dgrad: null pointer");
    hipLaunchKernelGGL(f32_convT_dgrad_kernel, dim3(blocks_for_elems((long)N * D * H * W * C)), dim3(NT), 0, (hipStream_t)s_, dy, ld_dy,
                       w, dx, ld_dx, N, D, H, W, C, K, g, accumulate);
    LNN_CHECK_LAUNCH("lnn_f32_convT3d_dgrad");
    return LNN_OK;
}
extern "C" int lnn_f32_convT3d_k2s2_dgrad(lnn_stream_t s, const float* dy, int ld_dy, const float* w, float* dx, int ld_dx, int N,
                                          int D, int H, int W, int C, int K, int accumulate) {
    return lnn_f32_convT3d_dgrad_g(s, dy, ld_dy, w, dx, ld_dx, N, D, H, W, C, K, 2, 2, 2, accumulate);
}

extern "C" int lnn_f32_convT3d_wgrad_g(lnn_stream_t s_, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N,
                                       int D, int H, int W, int C, int K, int sz, int sy, int sx) {
    const F32Geo g{sz, sy, sx, sz, sy, sx};
    if (int e = f32_check_geo(g, true, "lnn_f32_convT3d_wgrad")) return e;
    F32_REQ(x && dy && dw, "lnn_f32_convT3d_wgrad: null pointer");
    hipLaunchKernelGGL(f32_convT_wgrad_kernel, dim3(C * K * sz * sy * sx), dim3(NT), 0, (hipStream_t)s_, x, ld_x, dy, ld_dy, dw, N, D, H, W,
                       C, K, g);
    LNN_CHECK_LAUNCH("lnn_f32_convT3d_wgrad");
    return LNN_OK;
}
extern "C" int lnn_f32_convT3d_k2s2_wgrad(lnn_stream_t s, const float* x, int ld_x, const float* dy, int ld_dy, float* dw, int N,
                                          int D, int H, int W, int C, int K) {
    return lnn_f32_convT3d_wgrad_g(s, x, ld_x, dy, ld_dy, dw, N, D, H, W, C, K, 2, 2, 2);
}

/* (N, C, V) fp32 -> channels-last with channel stride ld: fp16 (the MFMA path's multi-channel image; ld = 16, the channels beyond C
 * stay zero) or fp32 (the parity path).  Replaces the implicit layout of torch's NCDHW conv input. */
extern "C" int lnn_image_to_cl_h(lnn_stream_t s_, const float* src, void* dst_h, int N, int C, long V, int ld) {
    F32_REQ(src && dst_h && C >= 1 && ld >= C, "lnn_image_to_cl_h: bad arguments");
    hipLaunchKernelGGL((ncdhw_to_cl_kernel<half_t>), dim3(blocks_for_elems((long)N * V)), dim3(NT), 0, (hipStream_t)s_, src, (half_t*)dst_h, N, C, V, ld);
    LNN_CHECK_LAUNCH("lnn_image_to_cl_h");
    return LNN_OK;
}
extern "C" int lnn_f32_image_to_cl(lnn_stream_t s_, const float* src, float* dst, int N, int C, long V, int ld) {
    F32_REQ(src && dst && C >= 1 && ld >= C, "lnn_f32_image_to_cl: bad arguments");
    hipLaunchKernelGGL((ncdhw_to_cl_kernel<float>), dim3(blocks_for_elems((long)N * V)), dim3(NT), 0, (hipStream_t)s_, src, dst, N, C, V, ld);
    LNN_CHECK_LAUNCH("lnn_f32_image_to_cl");
    return LNN_OK;
}

extern "C" int lnn_f32_instnorm_lrelu_fwd(lnn_stream_t s_, const float* y, int ld_y, float* z, int ld_z, int N, long V, int C, float eps,
                                          float* mean, float* rstd, const float* gamma, const float* beta, float slope) {
    F32_REQ(y && z && mean && rstd && gamma && beta, "lnn_f32_instnorm_lrelu_fwd: null pointer");
    hipLaunchKernelGGL(f32_in_stats_kernel, dim3(N * C), dim3(NT), 0, (hipStream_t)s_, y, ld_y, V, C, eps, mean, rstd);
    LNN_CHECK_LAUNCH("lnn_f32_instnorm_lrelu_fwd(stats)");
    hipLaunchKernelGGL(f32_in_lrelu_fwd_kernel, dim3(blocks_for_elems((long)N * V * C)), dim3(NT), 0, (hipStream_t)s_, y, ld_y, z, ld_z,
                       N, V, C, mean, rstd, gamma, beta, slope);
    LNN_CHECK_LAUNCH("lnn_f32_instnorm_lrelu_fwd");
    return LNN_OK;
}

/* y is overwritten with dL/dy; ws >= 2 * N * C doubles */
extern "C" int lnn_f32_instnorm_lrelu_bwd(lnn_stream_t s_, float* y, int ld_y, const float* dz, int ld_dz, int N, long V, int C,
                                          const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                                          float* dgamma, float* dbeta, double* ws) {
    F32_REQ(y && dz && mean && rstd && gamma && beta && ws, "lnn_f32_instnorm_lrelu_bwd: null pointer");
    hipStream_t s = (hipStream_t)s_;
    hipLaunchKernelGGL(f32_in_bwd_reduce_kernel, dim3(N * C), dim3(NT), 0, s, y, ld_y, dz, ld_dz, V, C, mean, rstd, gamma, beta, slope, ws);
    LNN_CHECK_LAUNCH("lnn_f32_instnorm_lrelu_bwd(reduce)");
    hipLaunchKernelGGL(f32_in_bwd_apply_kernel, dim3(blocks_for_elems((long)N * V * C)), dim3(NT), 0, s, y, ld_y, dz, ld_dz, N, V, C,
                       mean, rstd, gamma, beta, slope, ws);
    LNN_CHECK_LAUNCH("lnn_f32_instnorm_lrelu_bwd(apply)");
    hipLaunchKernelGGL(f32_in_bwd_params_kernel, dim3((C + 63) / 64), dim3(64), 0, s, ws, N, C, dgamma, dbeta);
    LNN_CHECK_LAUNCH("lnn_f32_instnorm_lrelu_bwd(params)");
    return LNN_OK;
}

extern "C" int lnn_f32_seg1x1_fwd(lnn_stream_t s_, const float* z, int ld_z, const float* w, float* logits, int N, long V, int C, int K) {
    F32_REQ(z && w && logits, "lnn_f32_seg1x1_fwd: null pointer");
    hipLaunchKernelGGL(f32_seg_fwd_kernel, dim3(blocks_for_elems((long)N * K * V)), dim3(NT), 0, (hipStream_t)s_, z, ld_z, w, logits, N, V, C, K);
    LNN_CHECK_LAUNCH("lnn_f32_seg1x1_fwd");
    return LNN_OK;
}

extern "C" int lnn_f32_seg1x1_bwd(lnn_stream_t s_, const float* z, int ld_z, const float* w, const float* dlogits, float* gz, int ld_gz,
                                  float* dw, int N, long V, int C, int K, int accumulate) {
    F32_REQ(z && w && dlogits && gz && dw, "lnn_f32_seg1x1_bwd: null pointer");
    hipStream_t s = (hipStream_t)s_;
    hipLaunchKernelGGL(f32_seg_dgrad_kernel, dim3(blocks_for_elems((long)N * V * C)), dim3(NT), 0, s, dlogits, w, gz, ld_gz, N, V, C, K, accumulate);
    LNN_CHECK_LAUNCH("lnn_f32_seg1x1_bwd(dgrad)");
    hipLaunchKernelGGL(f32_seg_wgrad_kernel, dim3(K * C), dim3(NT), 0, s, z, ld_z, dlogits, dw, N, V, C, K);
    LNN_CHECK_LAUNCH("lnn_f32_seg1x1_bwd(wgrad)");
    return LNN_OK;
}
