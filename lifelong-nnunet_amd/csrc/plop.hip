// PLOP / POD (nnunet_ext/training/loss_functions/deep_supervision.py:217-381, embeddings.py:3-41): the two device
// passes the distillation trainers add to a training step.  Both are HBM-bound streaming reductions, value-only
// (the reference stores DETACHED conv outputs in its forward hooks, plop/nnUNetTrainerPLOP.py:352-357, and the
// pseudo labels come from the old model), so neither has a backward.
//   * pseudo labels (DS.py:292-318): per voxel softmax of the OLD logits, its argmax and normalised entropy
//     (crossentropy.py:6-16), compared with the per-class threshold; emits the two label volumes the two CE terms
//     read (255 = ignored) and integer num/den counts per (sample, x-column) for the adaptive factor.
//   * local POD (embeddings.py:9-41): per (sample, channel, z) the L2 norm of the width- and height-pooled
//     multi-scale window means of (h - h_old), averaged; works on any 5-D strided view (channels-last fp16
//     activations, NCDHW fp32 logits) without a copy.  No atomics: every partial has one writer, the final sum runs
//     in a fixed order.
#include "lnn_common.h"

namespace {

constexpr int NT = 256;
constexpr int KMAX = 8;
constexpr int MAXS = 6;  // pooling scales held in registers

__global__ __launch_bounds__(NT) void plop_pseudo_label_kernel(const float* __restrict__ xo, const float* __restrict__ y,
                                                               const float* __restrict__ thr, float max_entropy, int K,
                                                               long V, int W, float* __restrict__ lab,
                                                               float* __restrict__ pseudo, int* __restrict__ num,
                                                               int* __restrict__ den) {
    const int n = blockIdx.y;
    const float* xn = xo + (long)n * K * V;
    const float factor = 1.f / logf((float)K + 1e-8f);
    for (long v = (long)blockIdx.x * NT + threadIdx.x; v < V; v += (long)gridDim.x * NT) {
        float x[KMAX], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                x[k] = xn[(long)k * V + v];
                mx = fmaxf(mx, x[k]);
            }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                x[k] = expf(x[k] - mx);
                s += x[k];
            }
        const float inv = 1.f / s;
        float pl = 0.f, e = 0.f;
        int am = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const float p = x[k] * inv;
                if (p > pl) {
                    pl = p;
                    am = k;  // first maximum, as torch.max
                }
                e += p * logf(p + 1e-8f);
            }
        const float ent = -factor * (e / (float)K);
        const bool valid = (ent / max_entropy) < thr[am];
        const float yv = y[(long)n * V + v];
        const bool bg = yv == 0.f;
        const bool m = valid && bg;
        lab[(long)n * V + v] = m ? 255.f : yv;
        pseudo[(long)n * V + v] = m ? (float)am : 255.f;
        const int col = n * W + (int)(v % W);
        if (bg) atomicAdd(den + col, 1);
        if (m) atomicAdd(num + col, 1);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void local_pod_kernel(const T* __restrict__ a, const T* __restrict__ b, long sn, long sc,
                                                       long sd, long sy, long sx, int N, int C, int D, int S, int scales,
                                                       int CB, float* __restrict__ normsq) {
    __shared__ float red[NT];
    const int d = blockIdx.x, n = blockIdx.y, cl = threadIdx.x % CB, slot = threadIdx.x / CB, LPB = NT / CB;
    const int c = blockIdx.z * CB + cl;
    int win[MAXS], P[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        win[s] = s < scales ? (S >> s) : 1;
        P[s] = (s < scales && s > 0 && S > win[s]) ? (S - win[s] + win[s] - 1) / win[s] : 0;  // len(range(0, S - win, win))
    }
    for (int dir = 0; dir < 2; ++dir) {
        // dir 0: a line is a column (fixed last index), walked along dim -2  -> height-pooled means (h_p)
        // dir 1: a line is a row, walked along the last dim                  -> width-pooled means  (w_p)
        const long sl = dir == 0 ? sx : sy, sw = dir == 0 ? sy : sx;
        float acc = 0.f;
        if (c < C) {
            for (int line = slot; line < S; line += LPB) {
                const long base = (long)n * sn + (long)c * sc + (long)d * sd + (long)line * sl;
                float sum[MAXS];
                int cnt[MAXS], k[MAXS];
#pragma unroll
                for (int s = 0; s < MAXS; ++s) sum[s] = 0.f, cnt[s] = 0, k[s] = 0;
                for (int p = 0; p < S; ++p) {
                    const float v = (float)a[base + (long)p * sw] - (float)b[base + (long)p * sw];
#pragma unroll
                    for (int s = 1; s < MAXS; ++s)
                        if (k[s] < P[s]) {
                            sum[s] += v;
                            if (++cnt[s] == win[s]) {
                                if (line < P[s] * win[s]) {
                                    const float m = sum[s] / (float)win[s];
                                    acc += m * m;
                                }
                                sum[s] = 0.f, cnt[s] = 0, ++k[s];
                            }
                        }
                }
            }
        }
        __syncthreads();
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < CB && c < C) {
            float t = 0.f;
            for (int j = 0; j < LPB; ++j) t += red[j * CB + threadIdx.x];
            normsq[(((long)dir * N + n) * C + c) * D + d] = t;
        }
    }
}

// pod = mean(sqrt(normsq)); dist = (dist + lambda * pod) / num_layers   (the division sits INSIDE the layer loop, DS.py:276)
__global__ __launch_bounds__(NT) void local_pod_finalize_kernel(const float* __restrict__ normsq, long count, float lambda,
                                                                int num_layers, float* __restrict__ dist, float* __restrict__ pod) {
    __shared__ double sm[NT];
    double acc = 0;
    for (long i = threadIdx.x; i < count; i += NT) acc += (double)sqrtf(normsq[i]);
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(sm[0] / (double)count);
        if (pod) pod[0] = mean;
        if (dist) dist[0] = (dist[0] + lambda * mean) / (float)num_layers;
    }
}

}  // namespace

extern "C" int lnn_plop_pseudo_labels(lnn_stream_t s_, const float* x_old, const float* y, const float* thresholds,
                                      float max_entropy, int N, int K, int D, int H, int W, float* labels_not_pseudo,
                                      float* labels_pseudo, int* num, int* den) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(x_old && y && thresholds && labels_not_pseudo && labels_pseudo && num && den, "lnn_plop_pseudo_labels: null pointer");
    LNN_REQUIRE(K >= 2 && K <= KMAX, "lnn_plop_pseudo_labels: K=%d unsupported (2..%d)", K, KMAX);
    LNN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "lnn_plop_pseudo_labels: empty volume");
    const long V = (long)D * H * W;
    hipMemsetAsync(num, 0, sizeof(int) * N * W, s);
    hipMemsetAsync(den, 0, sizeof(int) * N * W, s);
    const int blocks = (int)((V + NT - 1) / NT < 2048 ? (V + NT - 1) / NT : 2048);
    hipLaunchKernelGGL(plop_pseudo_label_kernel, dim3(blocks, N), dim3(NT), 0, s, x_old, y, thresholds, max_entropy, K, V, W,
                       labels_not_pseudo, labels_pseudo, num, den);
    LNN_CHECK_LAUNCH("lnn_plop_pseudo_labels");
    return LNN_OK;
}

extern "C" int lnn_local_pod(lnn_stream_t s_, const void* h, const void* h_old, int is_fp16, int N, int C, int D, int S,
                             long sn, long sc, long sd, long sy, long sx, int scales, float pod_lambda, int num_layers,
                             float* ws, float* dist_inout, float* pod_out) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(h && h_old && ws, "lnn_local_pod: null pointer");
    LNN_REQUIRE(N > 0 && C > 0 && D > 0 && S > 0, "lnn_local_pod: empty tensor");
    LNN_REQUIRE(scales >= 2 && scales <= MAXS, "lnn_local_pod: scales=%d unsupported (2..%d; scale 0 contributes no window)", scales, MAXS);
    LNN_REQUIRE((S >> (scales - 1)) > 0,
                "lnn_local_pod: the number of scales (%d) is too big: the window of the last scale is 0 for a side of %d", scales, S);
    LNN_REQUIRE(num_layers > 0 || !dist_inout, "lnn_local_pod: num_layers must be positive");
    const int CB = (sc == 1 && C >= 8) ? 8 : 1;  // channels-last: 8 adjacent channels per voxel share a line
    const dim3 grid(D, N, (C + CB - 1) / CB);
    if (is_fp16)
        hipLaunchKernelGGL((local_pod_kernel<half_t>), grid, dim3(NT), 0, s, (const half_t*)h, (const half_t*)h_old, sn, sc, sd, sy, sx,
                           N, C, D, S, scales, CB, ws);
    else
        hipLaunchKernelGGL((local_pod_kernel<float>), grid, dim3(NT), 0, s, (const float*)h, (const float*)h_old, sn, sc, sd, sy, sx, N,
                           C, D, S, scales, CB, ws);
    LNN_CHECK_LAUNCH("lnn_local_pod");
    hipLaunchKernelGGL(local_pod_finalize_kernel, dim3(1), dim3(NT), 0, s, ws, 2L * N * C * D, pod_lambda, num_layers, dist_inout,
                       pod_out);
    LNN_CHECK_LAUNCH("lnn_local_pod(finalize)");
    return LNN_OK;
}
