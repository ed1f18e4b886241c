// Segmentation head (1x1x1 conv) and the loss-side reductions: fused softmax + soft-Dice + CE for one
// deep-supervision level (forward statistics, backward dlogits), hard Dice counts, LwF distillation KL.
// All HBM-bound: one pass over the logits per kernel, fp32 math, fp64 cross-block accumulation.
#include "lnn_common.h"
#include <cstdlib>

namespace {

constexpr int NT = 256;
constexpr int KMAX = 8;  // logit channels held in registers
constexpr int LNN_DICE_CE_MAX_BATCH = 4096;   // per-sample CE partials of the finalize kernel live in (dynamic) LDS: 32 KB

// ---------------------------------------------------------------------------------------- seg 1x1x1
__global__ __launch_bounds__(NT) void seg_fwd_kernel(const half_t* __restrict__ z, int ld_z, const float* __restrict__ w,
                                                     float* __restrict__ logits, long V, int C, int K) {
    extern __shared__ float wl[];  // [K][C]
    for (int i = threadIdx.x; i < K * C; i += NT) wl[i] = w[i];
    __syncthreads();
    const int n = blockIdx.y;
    const half_t* zn = z + (long)n * V * ld_z;
    float* ln = logits + (long)n * K * V;
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < V; v += (long)gridDim.x * NT) {
        float acc[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
        for (int c = 0; c < C; c += 8) {
            const half8 x = *reinterpret_cast<const half8*>(zn + v * ld_z + c);
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[k] += (float)x[e] * wl[k * C + c + e];
                }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) ln[(long)k * V + v] = acc[k];
    }
}

// The same head with thread = (voxel, channel octet): the C/8 octets of a voxel are adjacent lanes of ONE wave (a group of G = the next
// power of two >= C/8 lanes, 64 / G voxels per wave and pass), every lane loads 16 bytes (a voxel's channels are one contiguous run),
// holds the K x 8 weights of its octet in registers and a butterfly over the group sums the K logits.  The voxel-per-thread kernel
// above walks a voxel's channels in a dependent loop of strided 16-byte loads: on the low-resolution heads (320 channels x 1200
// voxels) that is 40 round trips on ONE block per sample -- 92 us for 2 MFLOP (profiles/r05_step_timeline_before.txt); this one
// spreads the (sample, voxel) pairs over the chip.  KT: compile-time K (1..4), 0 = runtime K <= KMAX.  C <= 512.
template <int KT>
__global__ __launch_bounds__(NT) void seg_fwd_wave_kernel(const half_t* __restrict__ z, int ld_z, const float* __restrict__ w,
                                                          float* __restrict__ logits, long V, long NV, int C, int K, int G) {
    constexpr int KB = KT > 0 ? KT : KMAX;
    const int lane = threadIdx.x & 63, o = lane & (G - 1), g = lane / G, vpw = 64 / G, C8 = C >> 3;
    const bool act = o < C8;
    float wr[KB][8];
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[k][e] = (act && k < K) ? w[(long)k * C + o * 8 + e] : 0.f;
    const long wave = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nwaves = (long)gridDim.x * (NT / 64);
    for (long base = wave * vpw; base < NV; base += nwaves * vpw) {
        const long nv = base + g;                           // (sample, voxel) pair of this lane's group
        const bool ok = act && nv < NV;
        float acc[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) acc[k] = 0.f;
        if (ok) {
            const half8 x = *reinterpret_cast<const half8*>(z + nv * ld_z + o * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float zf = (float)x[e];
#pragma unroll
                for (int k = 0; k < KB; ++k) acc[k] = __builtin_fmaf(zf, wr[k][e], acc[k]);
            }
        }
        for (int m = 1; m < G; m <<= 1) {
#pragma unroll
            for (int k = 0; k < KB; ++k) acc[k] += __shfl_xor(acc[k], m, 64);
        }
        if (o == 0 && nv < NV) {
            const long n = nv / V, v = nv - n * V;
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k < K) logits[((long)n * K + k) * V + v] = acc[k];
        }
    }
}

// thread = (voxel lane, channel octet): dz = sum_k dl[k] w[k][c]; dw[k][c] partials in registers.
template <int KG>
__global__ __launch_bounds__(NT) void seg_bwd_kernel(const half_t* __restrict__ z, int ld_z, const float* __restrict__ w,
                                                     const float* __restrict__ dl, half_t* __restrict__ dz, int ld_dz,
                                                     float* __restrict__ dw, long V, int C, int K, int k0, int write_dz,
                                                     int accumulate_dz, float unscale, float* __restrict__ pws) {
    __shared__ float red[NT * (KG * 8 + 1)];
    const int C8 = C >> 3, VPB = NT / C8, c8 = threadIdx.x % C8, vl = threadIdx.x / C8;
    const bool active = vl < VPB;
    const int n = blockIdx.y;
    float part[KG][8];
#pragma unroll
    for (int k = 0; k < KG; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) part[k][e] = 0.f;
    if (active) {
        float wr[KMAX][8];
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) wr[k][e] = k < K ? w[k * C + c8 * 8 + e] : 0.f;
        const half_t* zn = z + (long)n * V * ld_z;
        half_t* dzn = dz + (long)n * V * ld_dz;
        const float* dln = dl + (long)n * K * V;
        for (long v = (long)blockIdx.x * VPB + vl; v < V; v += (long)gridDim.x * VPB) {
            float d[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) d[k] = k < K ? dln[(long)k * V + v] : 0.f;
            const half8 x = *reinterpret_cast<const half8*>(zn + v * ld_z + c8 * 8);
#pragma unroll
            for (int k = 0; k < KG; ++k)
                if (k0 + k < K) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) part[k][e] += d[k0 + k] * (float)x[e];
                }
            if (write_dz) {
                half8* dst = reinterpret_cast<half8*>(dzn + v * ld_dz + c8 * 8);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = 0.f;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) o[e] += d[k] * wr[k][e];
                }
                if (accumulate_dz) {
                    const half8 old = *dst;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += (float)old[e];
                }
                half8 ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (half_t)o[e];
                *dst = ov;
            }
        }
    }
    constexpr int W = KG * 8 + 1;
    __syncthreads();
    if (active) {
#pragma unroll
        for (int k = 0; k < KG; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[threadIdx.x * W + k * 8 + e] = part[k][e];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < KG * C; o += NT) {
        const int k = o / C, c = o % C;
        if (k0 + k >= K) continue;
        float s = 0.f;
        for (int l = 0; l < VPB; ++l) s += red[(l * C8 + (c >> 3)) * W + k * 8 + (c & 7)];
        // per-block partial (summed by seg_bwd_finalize_kernel): ~2000 blocks adding to the same K*C addresses
        // serialise in L2; the atomic path remains for callers without a workspace
        if (pws) pws[((long)blockIdx.y * gridDim.x + blockIdx.x) * KMAX * C + (long)(k0 + k) * C + c] = s;
        else atomicAdd(dw + (long)(k0 + k) * C + c, s * unscale);
    }
}

// dw[k][c] += unscale * sum over the nblk * N block partials; one thread block per 16 (k, c) entries
__global__ void seg_bwd_finalize_kernel(const float* __restrict__ pws, int nparts, int K, int C, float* __restrict__ dw,
                                        float unscale) {
    __shared__ double red[256];
    const int ii = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + ii;                // entry k * C + c
    double s = 0;
    if (i < K * C)
        for (int b = sl; b < nparts; b += 16) s += (double)pws[(long)b * KMAX * C + i];
    red[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && i < K * C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k * 16 + ii];
        atomicAdd(dw + i, (float)(s * unscale));       // (one add per entry and launch; two sample lanes may add concurrently)
    }
}

// ---------------------------------------------------------------------------------------- Dice + CE
__device__ __forceinline__ void softmax_k(const float* __restrict__ ln, long V, long v, int K, float inv_t, float (&p)[KMAX],
                                          float& lse, float (&x)[KMAX]) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
            x[k] = ln[(long)k * V + v] * inv_t;
            mx = fmaxf(mx, x[k]);
        }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) {
            p[k] = expf(x[k] - mx);
            s += p[k];
        }
    const float inv = 1.f / s;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) p[k] *= inv;
    lse = mx + logf(s);
}

// ws: [N][K][3] (tp, fp, fn) then [1] ce sum then [1] spare -- totals, written by the finalize kernel -- followed by the
// per-block partials [N][gridDim.x][3*KMAX+1] as fp32 (no atomics: 2048 blocks adding doubles to ONE CE address were
// most of this kernel's time, and the workspace needed a memset per call)
// softmax over register-resident logits; KT > 0: K is a compile-time constant (no per-channel uniform branches)
template <int KT>
__device__ __forceinline__ void softmax_regs(const float (&x)[KMAX], int K, float (&p)[KMAX], float& lse) {
    constexpr int KK = KT > 0 ? KT : KMAX;
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KK; ++k)
        if (KT > 0 || k < K) mx = fmaxf(mx, x[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KK; ++k)
        if (KT > 0 || k < K) {
            p[k] = expf(x[k] - mx);
            s += p[k];
        }
    const float inv = 1.f / s;
#pragma unroll
    for (int k = 0; k < KK; ++k)
        if (KT > 0 || k < K) p[k] *= inv;
    lse = mx + logf(s);
}

// The same softmax with the hardware transcendentals (round 4: the Dice + CE forward issued 111 VALU operations per voxel, 12 per
// accurate expf, 10 per IEEE division, 8 per logf -- the kernel was bound by them at 2.9 TB/s):
//   exp(d) = 2^(d log2 e) by v_exp_f32, with the rounding error of the product d * log2(e) (one fma) and the low word of log2(e) fed
//   back through 2^(t + e) ~= 2^t (1 + e ln 2): ~1 ulp like expf, 5 operations;  1 / s = v_rcp_f32 + one Newton step;
//   log(s) = v_log_f32(s) ln 2 (s in [1, K]: absolute error ~1e-7).
__device__ __forceinline__ float exp_le0(float d) {
    // exp(-126) is 0 in fp32 already; a logit of -inf would make the residual below inf - inf = NaN.  A select, not fmaxf: fmaxf
    // returns the non-NaN operand and a NaN logit (a blown-up step) would become probability 0 and a FINITE loss where torch's
    // softmax / cross_entropy give NaN -- the comparison is false for NaN, so it propagates as in the reference.
    d = d < -126.f ? -126.f : d;
    const float t = d * 1.44269504088896341f;
    float e = __builtin_fmaf(d, 1.44269504088896341f, -t);
    e = __builtin_fmaf(d, 1.92596299112661746e-8f, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, e * 0.69314718055994531f, r);
}
template <int KT>
__device__ __forceinline__ void softmax_regs_hw(const float (&x)[KMAX], int K, float (&p)[KMAX], float& lse) {
    constexpr int KK = KT > 0 ? KT : KMAX;
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KK; ++k)
        if (KT > 0 || k < K) mx = fmaxf(mx, x[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KK; ++k)
        if (KT > 0 || k < K) {
            p[k] = exp_le0(x[k] - mx);
            s += p[k];
        }
    float inv = __builtin_amdgcn_rcpf(s);
    inv = __builtin_fmaf(__builtin_fmaf(-s, inv, 1.f), inv, inv);
#pragma unroll
    for (int k = 0; k < KK; ++k)
        if (KT > 0 || k < K) p[k] *= inv;
    lse = __builtin_fmaf(__builtin_amdgcn_logf(s), 0.69314718055994531f, mx);
}

// KT: compile-time channel count (0 = runtime K); VEC = 4: four consecutive voxels per thread through 16-byte loads (V % 4 == 0)
// -- the scalar one-voxel-per-iteration version ran at 1.4 TB/s
template <int KT, int VEC>
__global__ __launch_bounds__(NT) void dice_ce_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                         int K, long V, double* ws, int N) {
    constexpr int KK = KT > 0 ? KT : KMAX;
    constexpr int NA = 3 * KK + 1;                  // accumulators this instantiation reduces (round 3 reduced 25 for K = 3)
    __shared__ float sm[NA * (NT / 64)];
    const int n = blockIdx.y;
    const float* ln = logits + (long)n * K * V;
    const float* yn = labels + (long)n * V;
    // per class: A = sum p y (tp), B = sum p, C = #{y} (integer): fp = B - A, fn = C - A at the end (3 accumulations per class and
    // voxel instead of 3 products + 3 accumulations)
    float accA[KK], accB[KK], ce = 0.f;
    int cnt[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) { accA[k] = accB[k] = 0.f; cnt[k] = 0; }
    auto load = [&](long v, float (&xv)[KMAX][VEC], float (&yv)[VEC]) {
#pragma unroll
        for (int k = 0; k < KK; ++k)
            if (KT > 0 || k < K) {
                if (VEC == 4) {
                    const floatx4 t = *reinterpret_cast<const floatx4*>(ln + (long)k * V + v);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) xv[k][e] = t[e];
                } else {
                    xv[k][0] = ln[(long)k * V + v];
                }
            }
        if (VEC == 4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(yn + v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) yv[e] = t[e];
        } else {
            yv[0] = yn[v];
        }
    };
    auto consume = [&](const float (&xv)[KMAX][VEC], const float (&yv)[VEC]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float x[KMAX], p[KMAX], lse;
#pragma unroll
            for (int k = 0; k < KK; ++k) x[k] = xv[k][e];
            softmax_regs_hw<KT>(x, K, p, lse);
            const int lab = (int)yv[e];
            float xl = lse;                         // a label outside [0, K) adds nothing to the cross-entropy
#pragma unroll
            for (int k = 0; k < KK; ++k)
                if (KT > 0 || k < K) {
                    const bool hit = k == lab;
                    accA[k] += hit ? p[k] : 0.f;
                    accB[k] += p[k];
                    cnt[k] += hit ? 1 : 0;
                    xl = hit ? x[k] : xl;
                }
            ce += lse - xl;
        }
    };
    // (two voxel groups per iteration -- 8 loads in flight instead of 4 -- measured SLOWER, 51 vs 46 us, while the kernel still issued
    // 111 VALU operations per voxel; profiles/r04_reduction_variants.txt)
    const long step = (long)gridDim.x * NT * VEC;
    for (long v = ((long)blockIdx.x * NT + threadIdx.x) * VEC; v < V; v += step) {
        float xa[KMAX][VEC], ya[VEC];
        load(v, xa, ya);
        consume(xa, ya);
    }
    float acc[NA];
#pragma unroll
    for (int k = 0; k < KK; ++k) {
        acc[k * 3 + 0] = accA[k];
        acc[k * 3 + 1] = accB[k] - accA[k];
        acc[k * 3 + 2] = (float)cnt[k] - accA[k];
    }
    acc[3 * KK] = ce;
    block_sum<NA>(acc, sm);
    if (threadIdx.x == 0) {
        // partials [n][value 0 .. 3 KMAX][block]: the finalize kernel reads one value's partials as a contiguous run
        float* pws = reinterpret_cast<float*>(ws + (long)N * K * 3 + 2) + (long)n * (3 * KMAX + 1) * gridDim.x + blockIdx.x;
#pragma unroll
        for (int i = 0; i < 3 * KK; ++i) pws[(long)i * gridDim.x] = acc[i];
        pws[(long)(3 * KMAX) * gridDim.x] = acc[3 * KK];
    }
}

// loss from the totals in ws ([N][K][3] tp/fp/fn, then the CE sum): -mean_{(n,)k>=1} dice + CE mean
__device__ void dice_ce_loss_from_totals(const double* ws, int N, int K, long V, int batch_dice, float smooth, float* out) {
    double dc_sum = 0;
    int cnt = 0;
    if (batch_dice) {
        for (int k = 1; k < K; ++k) {
            double tp = 0, fp = 0, fn = 0;
            for (int n = 0; n < N; ++n) {
                tp += ws[((long)n * K + k) * 3]; fp += ws[((long)n * K + k) * 3 + 1]; fn += ws[((long)n * K + k) * 3 + 2];
            }
            dc_sum += (2 * tp + smooth) / (2 * tp + fp + fn + smooth + 1e-8);
            ++cnt;
        }
    } else {
        for (int n = 0; n < N; ++n)
            for (int k = 1; k < K; ++k) {
                const double tp = ws[((long)n * K + k) * 3], fp = ws[((long)n * K + k) * 3 + 1], fn = ws[((long)n * K + k) * 3 + 2];
                dc_sum += (2 * tp + smooth) / (2 * tp + fp + fn + smooth + 1e-8);
                ++cnt;
            }
    }
    const double ce = ws[(long)N * K * 3] / ((double)N * (double)V);
    out[0] = (float)(ce - dc_sum / (cnt > 0 ? cnt : 1));
}

// one 256-thread block: totals of the per-block partials (fp64, fixed order), then the loss.  Each of the N * (3K + 1) totals
// is summed by ONE wave (lanes stride the blocks, butterfly at the end); the first version walked the totals one after the other
// with a block-wide tree per total: 20 x 8 barriers = 28 us per deep-supervision level.
constexpr int FIN_NT = 1024;      // 16 waves: the 20 totals of (N = 2, K = 3) in two rounds
__global__ __launch_bounds__(FIN_NT) void dice_ce_finalize_kernel(double* ws, int nblk, int N, int K, long V, int batch_dice,
                                                              float smooth, float* out, float weight, float* total, int accumulate) {
    extern __shared__ double ce_part[];               // N doubles (dynamic: any batch size up to LNN_DICE_CE_MAX_BATCH)
    constexpr int W = 3 * KMAX + 1;
    const float* pws = reinterpret_cast<const float*>(ws + (long)N * K * 3 + 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_n = 3 * K + 1;                      // tp/fp/fn of every class, then the CE sum
    for (int j = wave; j < N * per_n; j += FIN_NT / 64) {
        const int n = j / per_n, jj = j % per_n;
        const int i = jj < 3 * K ? jj : 3 * KMAX;
        double s = 0;
        for (int b = lane; b < nblk; b += 64) s += (double)pws[((long)n * W + i) * nblk + b];
        s = wave_sum_d(s);
        if (lane == 0) {
            if (i < 3 * KMAX) ws[((long)n * K + i / 3) * 3 + i % 3] = s;
            else ce_part[n] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double ce_sum = 0;
    for (int n = 0; n < N; ++n) ce_sum += ce_part[n];
    ws[(long)N * K * 3] = ce_sum;
    dice_ce_loss_from_totals(ws, N, K, V, batch_dice, smooth, out);
    // deep supervision: total (+)= weight * loss of this level (launches of one stream are ordered: plain read-modify-write)
    if (total) total[0] = (accumulate ? total[0] : 0.f) + weight * out[0];
}

// the loss again after the caller changed the totals (data-parallel batch Dice: tp/fp/fn summed over the ranks)
__global__ void dice_ce_from_totals_kernel(const double* ws, int N, int K, long V, int batch_dice, float smooth, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) dice_ce_loss_from_totals(ws, N, K, V, batch_dice, smooth, out);
}

template <int KT, int VEC>
__global__ __launch_bounds__(NT) void dice_ce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                         int N, int K, long V, int batch_dice, float smooth,
                                                         const double* __restrict__ ws, float gscale,
                                                         const float* __restrict__ gscale_dev, float dice_scale,
                                                         float* __restrict__ dlogits) {
    __shared__ float al[KMAX], be[KMAX];
    if (gscale_dev) gscale *= gscale_dev[0];
    const int n = blockIdx.y;
    if (threadIdx.x < KMAX) {
        const int k = threadIdx.x;
        float a = 0.f, b = 0.f;
        if (k >= 1 && k < K) {
            double tp = 0, fp = 0, fn = 0;
            if (batch_dice) {
                for (int m = 0; m < N; ++m) {
                    tp += ws[((long)m * K + k) * 3]; fp += ws[((long)m * K + k) * 3 + 1]; fn += ws[((long)m * K + k) * 3 + 2];
                }
            } else {
                tp = ws[((long)n * K + k) * 3]; fp = ws[((long)n * K + k) * 3 + 1]; fn = ws[((long)n * K + k) * 3 + 2];
            }
            const double cnt = batch_dice ? (double)(K - 1) : (double)N * (K - 1);
            const double num = 2 * tp + smooth, den = 2 * tp + fp + fn + smooth + 1e-8;
            a = (float)(-2.0 / (cnt * den)) * dice_scale;
            b = (float)(num / (cnt * den * den)) * dice_scale;
        }
        al[k] = a; be[k] = b;
    }
    __syncthreads();
    constexpr int KK = KT > 0 ? KT : KMAX;
    const float* ln = logits + (long)n * K * V;
    const float* yn = labels + (long)n * V;
    float* dn = dlogits + (long)n * K * V;
    const float inv_nv = 1.f / ((float)N * (float)V);
    for (long v = ((long)blockIdx.x * NT + threadIdx.x) * VEC; v < V; v += (long)gridDim.x * NT * VEC) {
        float xv[KMAX][VEC], yv[VEC], dv[KMAX][VEC];
#pragma unroll
        for (int k = 0; k < KK; ++k)
            if (KT > 0 || k < K) {
                if (VEC == 4) {
                    const floatx4 t = *reinterpret_cast<const floatx4*>(ln + (long)k * V + v);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) xv[k][e] = t[e];
                } else {
                    xv[k][0] = ln[(long)k * V + v];
                }
            }
        if (VEC == 4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(yn + v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) yv[e] = t[e];
        } else {
            yv[0] = yn[v];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float x[KMAX], p[KMAX], lse;
#pragma unroll
            for (int k = 0; k < KK; ++k) x[k] = xv[k][e];
            softmax_regs_hw<KT>(x, K, p, lse);
            const int lab = (int)yv[e];
            float a[KMAX], dot = 0.f;
#pragma unroll
            for (int k = 0; k < KK; ++k)
                if (KT > 0 || k < K) {
                    a[k] = (k == lab ? al[k] : 0.f) + be[k];
                    dot += a[k] * p[k];
                }
#pragma unroll
            for (int k = 0; k < KK; ++k)
                if (KT > 0 || k < K) {
                    const float y = (k == lab) ? 1.f : 0.f;
                    dv[k][e] = gscale * ((p[k] - y) * inv_nv + p[k] * (a[k] - dot));
                }
        }
#pragma unroll
        for (int k = 0; k < KK; ++k)
            if (KT > 0 || k < K) {
                if (VEC == 4) {
                    const floatx4 t = {dv[k][0], dv[k][1], dv[k][2], dv[k][3]};
                    *reinterpret_cast<floatx4*>(dn + (long)k * V + v) = t;
                } else {
                    dn[(long)k * V + v] = dv[k][0];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------- hard Dice
__global__ __launch_bounds__(NT) void online_dice_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                         int K, long V, float* counts) {
    __shared__ float sm[3 * (KMAX - 1) * (NT / 64)];
    const int n = blockIdx.y;
    const float* ln = logits + (long)n * K * V;
    const float* yn = labels + (long)n * V;
    float acc[3 * (KMAX - 1)];
#pragma unroll
    for (int i = 0; i < 3 * (KMAX - 1); ++i) acc[i] = 0.f;
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < V; v += (long)gridDim.x * NT) {
        int best = 0;
        float bv = ln[v];
#pragma unroll
        for (int k = 1; k < KMAX; ++k)
            if (k < K) {
                const float x = ln[(long)k * V + v];
                if (x > bv) { bv = x; best = k; }
            }
        const int lab = (int)yn[v];
#pragma unroll
        for (int k = 1; k < KMAX; ++k)
            if (k < K) {
                acc[(k - 1) * 3 + 0] += (best == k && lab == k) ? 1.f : 0.f;
                acc[(k - 1) * 3 + 1] += (best == k && lab != k) ? 1.f : 0.f;
                acc[(k - 1) * 3 + 2] += (best != k && lab == k) ? 1.f : 0.f;
            }
    }
    block_sum<3 * (KMAX - 1)>(acc, sm);
    if (threadIdx.x == 0)
        for (int i = 0; i < 3 * (K - 1); ++i) atomicAdd(counts + (long)n * (K - 1) * 3 + i, acc[i]);
}

// ---------------------------------------------------------------------------------------- LwF KL
// KL(softmax(t/T) || softmax(y/T)) summed over classes and voxels.  KT > 0: compile-time class count; VEC = 4: four consecutive
// voxels per thread through 16-byte loads of every class plane (the scalar version with one fp64 atomicAdd per block sat at
// 1.9 TB/s); per-block fp32 partials + a fixed-order fp64 finalize (deterministic, no workspace memset).
template <int KT, int VEC>
__global__ __launch_bounds__(NT) void kl_logits_kernel(const float* __restrict__ pred, const float* __restrict__ teach, int K,
                                                       long V, float inv_t, float* __restrict__ pws) {
    constexpr int KK = KT > 0 ? KT : KMAX;
    __shared__ float sm[NT / 64];
    const int n = blockIdx.y;
    const float* pn = pred + (long)n * K * V;
    const float* tn = teach + (long)n * K * V;
    float acc[1] = {0.f};
    for (long v = ((long)blockIdx.x * NT + threadIdx.x) * VEC; v < V; v += (long)gridDim.x * NT * VEC) {
        float xt[KMAX][VEC], xy[KMAX][VEC];
#pragma unroll
        for (int k = 0; k < KK; ++k)
            if (KT > 0 || k < K) {
                if (VEC == 4) {
                    const floatx4 a = *reinterpret_cast<const floatx4*>(tn + (long)k * V + v);
                    const floatx4 b = *reinterpret_cast<const floatx4*>(pn + (long)k * V + v);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { xt[k][e] = a[e] * inv_t; xy[k][e] = b[e] * inv_t; }
                } else {
                    xt[k][0] = tn[(long)k * V + v] * inv_t;
                    xy[k][0] = pn[(long)k * V + v] * inv_t;
                }
            }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float a[KMAX], b[KMAX], pt[KMAX], py[KMAX], lse_t, lse_y;
#pragma unroll
            for (int k = 0; k < KK; ++k) { a[k] = xt[k][e]; b[k] = xy[k][e]; }
            softmax_regs<KT>(a, K, pt, lse_t);
            softmax_regs<KT>(b, K, py, lse_y);
#pragma unroll
            for (int k = 0; k < KK; ++k)
                if (KT > 0 || k < K) acc[0] += pt[k] * ((a[k] - lse_t) - (b[k] - lse_y));
        }
    }
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) pws[(long)n * gridDim.x + blockIdx.x] = acc[0];
}
__global__ __launch_bounds__(NT) void kl_finalize_kernel(const float* __restrict__ pws, int nparts, int N, float* out) {
    __shared__ double red[NT];
    double s = 0;
    for (int b = threadIdx.x; b < nparts; b += NT) s += (double)pws[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)N);
}

// ---------------------------------------------------------------------------------------- MiB: CE and unbiased KD
// Cross-entropy of softmax(x) against a per-voxel target distribution q:
//   HARD: q = one_hot(label), voxels with label == ignore are skipped, mean over the counted voxels
//         (RobustCrossEntropyLoss(ignore_index=255), the base loss of MultipleOutputLossMiB, DS.py:393);
//   SOFT: q = softmax(alpha * t), mean over all voxels, divided by K
//         (UnbiasedKnowledgeDistillationLoss with equal class sets, knowledge_distillation.py:11-32: new_cl = K,
//          outputs_bkg = x_0 - lse, outputs_no_bkg = x_{1..} - lse, / targets.shape[1]).
// ws[0] = sum_v sum_k q_k * (lse - x_k), ws[1] = number of counted voxels.
template <int SOFT>
__global__ __launch_bounds__(NT) void target_ce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ tgt, int K, long V,
                                                           float alpha, int ignore, double* ws) {
    __shared__ float sm[2 * (NT / 64)];
    const int n = blockIdx.y;
    const float* xn = x + (long)n * K * V;
    float acc[2] = {0.f, 0.f};
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < V; v += (long)gridDim.x * NT) {
        float p[KMAX], xs[KMAX], lse;
        softmax_k(xn, V, v, K, 1.f, p, lse, xs);
        if (SOFT) {
            float q[KMAX], ts[KMAX], lt;
            softmax_k(tgt + (long)n * K * V, V, v, K, alpha, q, lt, ts);
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) acc[0] += q[k] * (lse - xs[k]);
            acc[1] += 1.f;
        } else {
            const int lab = (int)tgt[(long)n * V + v];
            if (lab != ignore) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k == lab) acc[0] += lse - xs[k];
                acc[1] += 1.f;
            }
        }
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) {
        atomicAdd(ws, (double)acc[0]);
        atomicAdd(ws + 1, (double)acc[1]);
    }
}
__global__ void target_ce_finalize_kernel(const double* ws, float scale, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = ws[1] > 0 ? (float)(ws[0] / ws[1]) * scale : 0.f;
}
// dx_k = gscale * scale / count * (softmax(x)_k - q_k)   (sum_k q_k = 1)
template <int SOFT>
__global__ __launch_bounds__(NT) void target_ce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ tgt, int K, long V,
                                                           float alpha, int ignore, const double* __restrict__ ws, float gscale,
                                                           const float* __restrict__ gscale_dev, float* __restrict__ dx) {
    if (gscale_dev) gscale *= gscale_dev[0];
    const float c = ws[1] > 0 ? gscale / (float)ws[1] : 0.f;
    const int n = blockIdx.y;
    const float* xn = x + (long)n * K * V;
    float* dn = dx + (long)n * K * V;
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < V; v += (long)gridDim.x * NT) {
        float p[KMAX], xs[KMAX], lse, q[KMAX];
        softmax_k(xn, V, v, K, 1.f, p, lse, xs);
        float live = 1.f;
        if (SOFT) {
            float ts[KMAX], lt;
            softmax_k(tgt + (long)n * K * V, V, v, K, alpha, q, lt, ts);
        } else {
            const int lab = (int)tgt[(long)n * V + v];
            if (lab == ignore) live = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) q[k] = (k == lab) ? 1.f : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) dn[(long)k * V + v] = live * c * (p[k] - q[k]);
    }
}

// ---------------------------------------------------------------------------------------- sliding-window inference
// One tile of the tiled predictor (upstream SegmentationNetwork._internal_predict_3D_3Dconv_tiled, reached through
// predict.py:208-219 / MH.py:1115): agg[k, o + flipback(v)] += weight * gauss[flipback(v)] * softmax(logits[:, v])[k],
// nb[o + v] += gauss[v] when add_nb.  flip bits: 4 = z, 2 = y, 1 = x (the network saw the tile mirrored on them).
__global__ __launch_bounds__(NT) void softmax_accumulate_kernel(const float* __restrict__ logits, const float* __restrict__ gauss,
                                                                float* __restrict__ agg, float* __restrict__ nb, int K, int pd,
                                                                int ph, int pw, int D, int H, int W, int oz, int oy, int ox,
                                                                int flip, float weight, int add_nb) {
    const long Vp = (long)pd * ph * pw, Vi = (long)D * H * W;
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < Vp; v += (long)gridDim.x * NT) {
        float p[KMAX], x[KMAX], lse;
        softmax_k(logits, Vp, v, K, 1.f, p, lse, x);
        int z = (int)(v / ((long)ph * pw)), y = (int)((v / pw) % ph), xx = (int)(v % pw);
        if (flip & 4) z = pd - 1 - z;
        if (flip & 2) y = ph - 1 - y;
        if (flip & 1) xx = pw - 1 - xx;
        const long vo = ((long)z * ph + y) * pw + xx;
        const float g = gauss ? gauss[vo] : 1.f;
        const long vi = ((long)(oz + z) * H + (oy + y)) * W + (ox + xx);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) agg[(long)k * Vi + vi] += weight * g * p[k];
        if (add_nb) nb[vi] += g;
    }
}

// class probabilities = agg / nb (in place), segmentation = argmax_k (first maximum, like numpy / torch)
__global__ __launch_bounds__(NT) void softmax_finalize_kernel(float* __restrict__ agg, const float* __restrict__ nb, int K, long V,
                                                              int* __restrict__ seg) {
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < V; v += (long)gridDim.x * NT) {
        const float inv = 1.f / nb[v];
        float best = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const float q = agg[(long)k * V + v] * inv;
                agg[(long)k * V + v] = q;
                if (q > best) { best = q; arg = k; }
            }
        seg[v] = arg;
    }
}

int vox_blocks(long V) {
    long b = (V + NT * 8 - 1) / (NT * 8);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int lnn_seg1x1_fwd(lnn_stream_t s_, const void* z, int ld_z, const float* w, float* logits, int N, long V, int C,
                              int K) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(z && lnn_aligned16(z) && w && logits, "lnn_seg1x1_fwd: null/misaligned pointer");
    LNN_REQUIRE(C % 8 == 0 && ld_z >= C && ld_z % 8 == 0, "lnn_seg1x1_fwd: bad channel count / ld");
    LNN_REQUIRE(K >= 1 && K <= KMAX, "lnn_seg1x1_fwd: K=%d unsupported (max %d)", K, KMAX);
    // (128 channels and more: up to 4 voxels per wave and pass.  Below, a voxel's channels are 1-4 loads of the voxel-per-thread
    // kernel, whose logits stores are fully coalesced; this one writes 64-byte pieces)
    if (C >= 128 && C <= 512) {
        int G = 1;
        while (G < C / 8) G <<= 1;
        const long NV = (long)N * V, waves = (NV + 64 / G - 1) / (64 / G);
        long blocks = (waves + NT / 64 - 1) / (NT / 64);
        if (blocks > 2048) blocks = 2048;
#define LNN_SEGW(KT) hipLaunchKernelGGL((seg_fwd_wave_kernel<KT>), dim3((unsigned)blocks), dim3(NT), 0, s, (const half_t*)z, ld_z, w, \
                                        logits, V, NV, C, K, G)
        if (K == 1) LNN_SEGW(1); else if (K == 2) LNN_SEGW(2); else if (K == 3) LNN_SEGW(3); else if (K == 4) LNN_SEGW(4); else LNN_SEGW(0);
#undef LNN_SEGW
        LNN_CHECK_LAUNCH("lnn_seg1x1_fwd(wave)");
        return LNN_OK;
    }
    hipLaunchKernelGGL(seg_fwd_kernel, dim3(vox_blocks(V), N), dim3(NT), (size_t)K * C * sizeof(float), s,
                       (const half_t*)z, ld_z, w, logits, V, C, K);
    LNN_CHECK_LAUNCH("lnn_seg1x1_fwd");
    return LNN_OK;
}

extern "C" size_t lnn_seg1x1_bwd_ws_floats(int N, int C) { return (size_t)N * 1024 * KMAX * C; }

extern "C" int lnn_seg1x1_bwd(lnn_stream_t s_, const void* z, int ld_z, const float* w, const float* dlogits, void* dz,
                              int ld_dz, float* dw, int N, long V, int C, int K, int accumulate_dz, float grad_unscale,
                              float* ws) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(z && lnn_aligned16(z) && w && dlogits && dz && lnn_aligned16(dz) && dw, "lnn_seg1x1_bwd: null/misaligned pointer");
    LNN_REQUIRE(C % 8 == 0 && C <= 2048 && ld_z >= C && ld_z % 8 == 0 && ld_dz >= C && ld_dz % 8 == 0, "lnn_seg1x1_bwd: bad channel count / ld");
    LNN_REQUIRE(K >= 1 && K <= KMAX, "lnn_seg1x1_bwd: K=%d unsupported (max %d)", K, KMAX);
    const int vpb = NT / (C / 8);
    long b = (V + (long)vpb * 8 - 1) / ((long)vpb * 8);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    for (int k0 = 0; k0 < K; k0 += 4) {
        hipLaunchKernelGGL((seg_bwd_kernel<4>), dim3((int)b, N), dim3(NT), 0, s, (const half_t*)z, ld_z, w, dlogits,
                           (half_t*)dz, ld_dz, dw, V, C, K, k0, k0 == 0 ? 1 : 0, accumulate_dz, grad_unscale, ws);
        LNN_CHECK_LAUNCH("lnn_seg1x1_bwd");
    }
    if (ws) {
        hipLaunchKernelGGL(seg_bwd_finalize_kernel, dim3(lnn_cdiv(K * C, 16)), dim3(256), 0, s, ws, (int)b * N, K, C, dw, grad_unscale);
        LNN_CHECK_LAUNCH("lnn_seg1x1_bwd(finalize)");
    }
    return LNN_OK;
}

// totals [N*K*3 + 2] + fp32 per-block partials [N][<=1024 blocks][3*KMAX+1]
extern "C" size_t lnn_dice_ce_ws_doubles(int N, int K) { return (size_t)N * K * 3 + 2 + ((size_t)N * 1024 * (3 * KMAX + 1) + 1) / 2; }

static int dice_ce_fwd_impl(lnn_stream_t s_, const float* logits, const float* labels, int N, int K, long V,
                            int batch_dice, float smooth, float* out_loss, double* ws, float weight, float* total, int accumulate) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(logits && labels && out_loss && ws, "lnn_dice_ce_fwd: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_dice_ce_fwd: K=%d unsupported (2..%d)", K, KMAX);
    LNN_REQUIRE(N >= 1 && N <= LNN_DICE_CE_MAX_BATCH, "lnn_dice_ce_fwd: batch %d unsupported (1..%d)", N, LNN_DICE_CE_MAX_BATCH);
    // 1024 blocks per launch at most: the per-block reduction of 3K + 1 values and the finalize pass scale with the block count
    // (measured at 2 x 3 x 160x192x160, profiles/r04_reduction_variants.txt: with the 111-operation inner loop 256 blocks per sample 57 us,
    // 512 / 1024 46 us, 2048 59 us; with the hardware-transcendental loop 256 per sample 31.9 us, 512 34.0, 1024 39.4)
    int nblk = vox_blocks(V);
    static int cap_env = -1;          // LNN_DCE_BLOCKS: blocks per sample (A/B measurements; tools/microbench_reductions.py)
    if (cap_env < 0) { const char* e = getenv("LNN_DCE_BLOCKS"); cap_env = e ? atoi(e) : 0; }
    const int cap = cap_env > 0 ? (cap_env < 1024 ? cap_env : 1024) : (512 / N > 1 ? 512 / N : 1);
    if (nblk > cap) nblk = cap;
    const bool vec = (V & 3) == 0 && lnn_aligned16(logits) && lnn_aligned16(labels);
#define LNN_DCE_FWD(KT, VEC) hipLaunchKernelGGL((dice_ce_fwd_kernel<KT, VEC>), dim3(nblk, N), dim3(NT), 0, s, logits, labels, K, V, ws, N)
    if (vec) { if (K == 3) LNN_DCE_FWD(3, 4); else if (K == 2) LNN_DCE_FWD(2, 4); else if (K == 4) LNN_DCE_FWD(4, 4); else LNN_DCE_FWD(0, 4); }
    else { if (K == 3) LNN_DCE_FWD(3, 1); else if (K == 2) LNN_DCE_FWD(2, 1); else if (K == 4) LNN_DCE_FWD(4, 1); else LNN_DCE_FWD(0, 1); }
#undef LNN_DCE_FWD
    LNN_CHECK_LAUNCH("lnn_dice_ce_fwd");
    hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(FIN_NT), N * sizeof(double), s, ws, nblk, N, K, V, batch_dice, smooth, out_loss, weight,
                       total, accumulate);
    LNN_CHECK_LAUNCH("lnn_dice_ce_fwd(finalize)");
    return LNN_OK;
}

extern "C" int lnn_dice_ce_fwd(lnn_stream_t s, const float* logits, const float* labels, int N, int K, long V,
                               int batch_dice, float smooth, float* out_loss, double* ws) {
    return dice_ce_fwd_impl(s, logits, labels, N, K, V, batch_dice, smooth, out_loss, ws, 0.f, nullptr, 0);
}

extern "C" int lnn_dice_ce_fwd_ds(lnn_stream_t s, const float* logits, const float* labels, int N, int K, long V,
                                  int batch_dice, float smooth, float* out_loss, double* ws, float weight, float* total,
                                  int accumulate) {
    LNN_REQUIRE(total != nullptr, "lnn_dice_ce_fwd_ds: null total");
    return dice_ce_fwd_impl(s, logits, labels, N, K, V, batch_dice, smooth, out_loss, ws, weight, total, accumulate);
}

extern "C" int lnn_dice_ce_loss_from_totals(lnn_stream_t s_, const double* ws, int N, int K, long V, int batch_dice, float smooth,
                                            float* out_loss) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(ws && out_loss, "lnn_dice_ce_loss_from_totals: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_dice_ce_loss_from_totals: K=%d unsupported (2..%d)", K, KMAX);
    hipLaunchKernelGGL(dice_ce_from_totals_kernel, dim3(1), dim3(64), 0, s, ws, N, K, V, batch_dice, smooth, out_loss);
    LNN_CHECK_LAUNCH("lnn_dice_ce_loss_from_totals");
    return LNN_OK;
}

extern "C" int lnn_dice_ce_bwd(lnn_stream_t s_, const float* logits, const float* labels, int N, int K, long V,
                               int batch_dice, float smooth, const double* ws, float gscale, const float* gscale_dev, float dice_scale,
                               float* dlogits) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(logits && labels && dlogits && ws, "lnn_dice_ce_bwd: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_dice_ce_bwd: K=%d unsupported (2..%d)", K, KMAX);
    const bool vec = (V & 3) == 0 && lnn_aligned16(logits) && lnn_aligned16(labels) && lnn_aligned16(dlogits);
#define LNN_DCE_BWD(KT, VEC)                                                                                                  \
    hipLaunchKernelGGL((dice_ce_bwd_kernel<KT, VEC>), dim3(vox_blocks(V), N), dim3(NT), 0, s, logits, labels, N, K, V, batch_dice, \
                       smooth, ws, gscale, gscale_dev, dice_scale, dlogits)
    if (vec) { if (K == 3) LNN_DCE_BWD(3, 4); else if (K == 2) LNN_DCE_BWD(2, 4); else if (K == 4) LNN_DCE_BWD(4, 4); else LNN_DCE_BWD(0, 4); }
    else { if (K == 3) LNN_DCE_BWD(3, 1); else if (K == 2) LNN_DCE_BWD(2, 1); else if (K == 4) LNN_DCE_BWD(4, 1); else LNN_DCE_BWD(0, 1); }
#undef LNN_DCE_BWD
    LNN_CHECK_LAUNCH("lnn_dice_ce_bwd");
    return LNN_OK;
}

extern "C" int lnn_online_dice_counts(lnn_stream_t s_, const float* logits, const float* labels, int N, int K, long V,
                                      float* counts) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(logits && labels && counts, "lnn_online_dice_counts: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_online_dice_counts: K=%d unsupported (2..%d)", K, KMAX);
    hipMemsetAsync(counts, 0, sizeof(float) * N * (K - 1) * 3, s);
    hipLaunchKernelGGL(online_dice_kernel, dim3(vox_blocks(V), N), dim3(NT), 0, s, logits, labels, K, V, counts);
    LNN_CHECK_LAUNCH("lnn_online_dice_counts");
    return LNN_OK;
}

extern "C" int lnn_target_ce_fwd(lnn_stream_t s_, const float* x, const float* target, int soft, int N, int K, long V, float alpha,
                                 int ignore_index, float scale, float* out, double* ws) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(x && target && out && ws, "lnn_target_ce_fwd: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_target_ce_fwd: K=%d unsupported (2..%d)", K, KMAX);
    hipMemsetAsync(ws, 0, 2 * sizeof(double), s);
    if (soft) hipLaunchKernelGGL((target_ce_fwd_kernel<1>), dim3(vox_blocks(V), N), dim3(NT), 0, s, x, target, K, V, alpha, ignore_index, ws);
    else hipLaunchKernelGGL((target_ce_fwd_kernel<0>), dim3(vox_blocks(V), N), dim3(NT), 0, s, x, target, K, V, alpha, ignore_index, ws);
    LNN_CHECK_LAUNCH("lnn_target_ce_fwd");
    hipLaunchKernelGGL(target_ce_finalize_kernel, dim3(1), dim3(64), 0, s, ws, scale, out);
    LNN_CHECK_LAUNCH("lnn_target_ce_fwd(finalize)");
    return LNN_OK;
}

extern "C" int lnn_target_ce_bwd(lnn_stream_t s_, const float* x, const float* target, int soft, int N, int K, long V, float alpha,
                                 int ignore_index, float scale, const double* ws, float gscale, const float* gscale_dev, float* dx) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(x && target && ws && dx, "lnn_target_ce_bwd: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_target_ce_bwd: K=%d unsupported (2..%d)", K, KMAX);
    if (soft) hipLaunchKernelGGL((target_ce_bwd_kernel<1>), dim3(vox_blocks(V), N), dim3(NT), 0, s, x, target, K, V, alpha, ignore_index, ws,
                                 gscale * scale, gscale_dev, dx);
    else hipLaunchKernelGGL((target_ce_bwd_kernel<0>), dim3(vox_blocks(V), N), dim3(NT), 0, s, x, target, K, V, alpha, ignore_index, ws,
                            gscale * scale, gscale_dev, dx);
    LNN_CHECK_LAUNCH("lnn_target_ce_bwd");
    return LNN_OK;
}

extern "C" int lnn_softmax_accumulate(lnn_stream_t s_, const float* logits, const float* gauss, float* agg, float* nb, int K,
                                      int pd, int ph, int pw, int D, int H, int W, int oz, int oy, int ox, int flip_mask,
                                      float weight, int add_nb) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(logits && agg && (nb || !add_nb), "lnn_softmax_accumulate: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_softmax_accumulate: K=%d unsupported (2..%d)", K, KMAX);
    LNN_REQUIRE(pd > 0 && ph > 0 && pw > 0 && oz >= 0 && oy >= 0 && ox >= 0 && oz + pd <= D && oy + ph <= H && ox + pw <= W,
                "lnn_softmax_accumulate: tile (%d,%d,%d)+(%d,%d,%d) outside the volume (%d,%d,%d)", oz, oy, ox, pd, ph, pw, D, H, W);
    hipLaunchKernelGGL(softmax_accumulate_kernel, dim3(vox_blocks((long)pd * ph * pw)), dim3(NT), 0, s, logits, gauss, agg, nb, K, pd,
                       ph, pw, D, H, W, oz, oy, ox, flip_mask, weight, add_nb);
    LNN_CHECK_LAUNCH("lnn_softmax_accumulate");
    return LNN_OK;
}

extern "C" int lnn_softmax_finalize(lnn_stream_t s_, float* agg, const float* nb, int K, long V, int* seg) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(agg && nb && seg, "lnn_softmax_finalize: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_softmax_finalize: K=%d unsupported (2..%d)", K, KMAX);
    hipLaunchKernelGGL(softmax_finalize_kernel, dim3(vox_blocks(V)), dim3(NT), 0, s, agg, nb, K, V, seg);
    LNN_CHECK_LAUNCH("lnn_softmax_finalize");
    return LNN_OK;
}

extern "C" size_t lnn_kl_logits_ws_doubles(int N) { return ((size_t)N * 1024 + 1) / 2 + 1; }      // fp32 partial per (sample, block)

extern "C" int lnn_kl_logits(lnn_stream_t s_, const float* pred, const float* teach, int N, int K, long V, float T,
                             float* out, double* ws) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(pred && teach && out && ws, "lnn_kl_logits: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX && T > 0.f, "lnn_kl_logits: K=%d / T unsupported", K);
    const int nblk = vox_blocks(V);
    float* pws = reinterpret_cast<float*>(ws);
    const bool vec = (V & 3) == 0 && lnn_aligned16(pred) && lnn_aligned16(teach);
#define LNN_KL(KT, VEC) hipLaunchKernelGGL((kl_logits_kernel<KT, VEC>), dim3(nblk, N), dim3(NT), 0, s, pred, teach, K, V, 1.f / T, pws)
    if (vec) { if (K == 3) LNN_KL(3, 4); else if (K == 2) LNN_KL(2, 4); else if (K == 4) LNN_KL(4, 4); else LNN_KL(0, 4); }
    else { if (K == 3) LNN_KL(3, 1); else if (K == 2) LNN_KL(2, 1); else if (K == 4) LNN_KL(4, 1); else LNN_KL(0, 1); }
#undef LNN_KL
    LNN_CHECK_LAUNCH("lnn_kl_logits");
    hipLaunchKernelGGL(kl_finalize_kernel, dim3(1), dim3(NT), 0, s, pws, nblk * N, N, out);
    LNN_CHECK_LAUNCH("lnn_kl_logits(finalize)");
    return LNN_OK;
}
