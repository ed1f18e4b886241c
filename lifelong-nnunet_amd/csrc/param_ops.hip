// Flat-arena parameter kernels (fp32, HBM-bound, one pass each): weight panel packing / gradient panel
// unpacking, EWC penalty forward + backward, Fisher extraction, gradient norm, fused SGD-Nesterov.
// Plus the library-wide error string and device query.
#include "lnn_common.h"
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void lnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* lnn_last_error(void) { return g_err; }
extern "C" int lnn_version(void) { return 100; }
extern "C" int lnn_device_info(int* cu_count, int* clock_khz, char* name, int name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { lnn_set_error("lnn_device_info: no device"); return LNN_ERR_LAUNCH; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { lnn_set_error("lnn_device_info: query failed"); return LNN_ERR_LAUNCH; }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (name && name_len > 0) { strncpy(name, prop.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    return LNN_OK;
}

namespace {
constexpr int NT = 256;
constexpr int RED_BLOCKS = 512;          // partial sums per reduction = what lnn_flat_reduce_ws_doubles() reports

int red_blocks(long n) {                 // blocks of a two-stage reduction: <= RED_BLOCKS partial sums
    long b = (n + (long)NT * 16 - 1) / ((long)NT * 16);
    return (int)(b < 1 ? 1 : (b > RED_BLOCKS ? RED_BLOCKS : b));
}

int flat_blocks(long n, int per_thread) {
    long b = (n + (long)NT * per_thread - 1) / ((long)NT * per_thread);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

__global__ void pack_weights_kernel(const float* __restrict__ src, half_t* __restrict__ dst, int ntaps, int M, int KC, int Mpad,
                                    int KCpad, long sm, long skc, long st) {
    const long total = (long)ntaps * Mpad * KCpad;
    const int nck = KCpad / 16;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // blocked panel: i = (((mb * nck + ck) * ntaps + t) * 32 + row) * 16 + c16
        const int c16 = (int)(i & 15), row = (int)((i >> 4) & 31);
        long r = i >> 9;
        const int t = (int)(r % ntaps); r /= ntaps;
        const int ck = (int)(r % nck), mb = (int)(r / nck);
        const int m = mb * 32 + row, kc = ck * 16 + c16;
        float v = 0.f;
        if (m < M && kc < KC) v = src[m * sm + kc * skc + t * st];
        dst[i] = (half_t)v;
    }
}

// ---- batched variants: ONE launch for every layer of the network (a step needs 53 packs + 27 unpacks, each a
// 5-20 us launch on its own).  desc[j] = {src_off, dst_off, stride_m, stride_kc, stride_t, ntaps, M, KC, first} in
// int64 (element offsets relative to the base pointers; `first` = index of the layer's first work item).
constexpr int DESC_W = 9, DESC_MAX = 128;

__device__ __forceinline__ int find_desc(const long* firsts, int n, long i) {
    int lo = 0, hi = n - 1;                       // last j with firsts[j] <= i
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (firsts[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(NT) void pack_weights_batched_kernel(const float* __restrict__ src, half_t* __restrict__ dst,
                                                                  const long* __restrict__ desc, int n, long total) {
    __shared__ long firsts[DESC_MAX];
    for (int j = threadIdx.x; j < n; j += NT) firsts[j] = desc[j * DESC_W + 8];
    __syncthreads();
    // one thread = 8 consecutive panel elements (same tap and row, channels kc .. kc+7): one 16-byte store, the
    // descriptor search and the index decomposition once per 8 elements (panel sizes are multiples of 512)
    const long total8 = total >> 3;
    for (long g8 = (long)blockIdx.x * NT + threadIdx.x; g8 < total8; g8 += (long)gridDim.x * NT) {
        const long g = g8 << 3;
        const long* d = desc + find_desc(firsts, n, g) * DESC_W;
        const long i = g - d[8];
        const int ntaps = (int)d[5], M = (int)d[6], KC = (int)d[7];
        const int nck = (KC + 15) >> 4;
        const int c16 = (int)(i & 15), row = (int)((i >> 4) & 31);
        long r = i >> 9;
        const int t = (int)(r % ntaps); r /= ntaps;
        const int ck = (int)(r % nck), mb = (int)(r / nck);
        const int m = mb * 32 + row, kc = ck * 16 + c16;
        half8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m < M) {
            const float* sp = src + d[0] + m * d[2] + (long)kc * d[3] + t * d[4];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kc + j < KC) o[j] = (half_t)sp[j * d[3]];
        }
        *reinterpret_cast<half8*>(dst + d[1] + i) = o;
    }
}

// Tiled variant (round 2): the blocked panel was designed so that one (32-row block, 16-channel chunk) is a contiguous
// ntaps x 1 KB run, but the element-wise kernel above gathers it with eight 4-byte loads 108 B apart per thread (64
// cache lines per wave-wide load: 0.38 ms per step at 0.66 TB/s).  Here a block owns one such unit: it reads the
// 32 x 16 x ntaps source elements in SOURCE order (runs of 16 x 27 or 32 x 27 consecutive floats), transposes through
// LDS and writes the run with 16-byte stores.
// NTAPS: compile-time tap count (27 / 8 / 1): e % ntaps and e / ntaps are 54 runtime integer divisions per thread and unit
// otherwise -- the kernel sat at 1.85 TB/s, bound by that arithmetic, not by memory.
template <int NTAPS>
__device__ __forceinline__ void pack_unit_load(const float* __restrict__ sp, half_t* tile, int TS, int mb, int ck, int M, int KC, long sm,
                                               long skc, long st, bool kc_inner) {
    constexpr int E = 512 * NTAPS;
    // full units of tap-contiguous tensors: the unit is 32 (16) source runs of 16 (32) x NTAPS consecutive floats -> 16-byte loads
    // (a quarter of the load instructions; the scalar loop below remains for ragged units and the first layer)
    const bool full = mb * 32 + 32 <= M && ck * 16 + 16 <= KC && st == 1 && (kc_inner ? skc : sm) == NTAPS &&
                      ((kc_inner ? sm : skc) & 3) == 0 && (reinterpret_cast<unsigned long long>(sp) & 15) == 0;
    // (only the OUTER stride has to keep the run starts 16-byte aligned: the inner one is NTAPS itself -- 27 for every 3x3x3
    // layer -- and a unit starts at a multiple of 16 * NTAPS or 32 * NTAPS floats, a multiple of 4 either way)
    if (NTAPS > 1 && full) {
        const int RL4 = (kc_inner ? 4 : 8) * NTAPS;          // float4 per run
        const float* base = sp + (long)mb * 32 * sm + (long)ck * 16 * skc;
        for (int q = threadIdx.x; q < 128 * NTAPS; q += NT) {
            const int o = q / RL4, j = q - o * RL4;
            const floatx4 v = *reinterpret_cast<const floatx4*>(base + o * (kc_inner ? sm : skc) + 4 * j);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * j + k, ir = x / NTAPS, t = x - ir * NTAPS;
                const int kcl = kc_inner ? ir : o, ml = kc_inner ? o : ir;
                tile[t * TS + ml * 16 + kcl] = (half_t)v[k];
            }
        }
        return;
    }
    for (int e = threadIdx.x; e < E; e += NT) {
        const int t = e % NTAPS, r = e / NTAPS;
        const int kcl = kc_inner ? r % 16 : r / 32, ml = kc_inner ? r / 16 : r % 32;
        const int m = mb * 32 + ml, kc = ck * 16 + kcl;
        const float v = (m < M && kc < KC) ? sp[m * sm + kc * skc + t * st] : 0.f;
        tile[t * TS + ml * 16 + kcl] = (half_t)v;
    }
}

__global__ __launch_bounds__(NT) void pack_weights_tiled_kernel(const float* __restrict__ src, half_t* __restrict__ dst,
                                                                const long* __restrict__ desc, int n) {
    constexpr int TS = 520;                        // tap pitch in LDS (halves): 512 + 8 keeps consecutive taps on different banks
    __shared__ long ufirst[DESC_MAX + 1];
    __shared__ __attribute__((aligned(16))) half_t tile[27 * TS];
    if (threadIdx.x == 0) {
        long acc = 0;
        for (int j = 0; j < n; ++j) {
            ufirst[j] = acc;
            const long M = desc[j * DESC_W + 6], KC = desc[j * DESC_W + 7];
            acc += ((M + 31) >> 5) * ((KC + 15) >> 4);
        }
        ufirst[n] = acc;
    }
    __syncthreads();
    const long units = ufirst[n];
    for (long u = blockIdx.x; u < units; u += gridDim.x) {
        const long* d = desc + find_desc(ufirst, n, u) * DESC_W;
        const int ntaps = (int)d[5], M = (int)d[6], KC = (int)d[7];
        const int nck = (KC + 15) >> 4;
        const long lu = u - ufirst[find_desc(ufirst, n, u)];
        const int mb = (int)(lu / nck), ck = (int)(lu % nck);
        const long sm = d[2], skc = d[3], st = d[4];
        const bool kc_inner = skc < sm;            // forward orientation: (kc, t) runs inside a row; dgrad: (m, t) runs
        const int E = 512 * ntaps;
        const float* sp = src + d[0];
        if (ntaps == 27) pack_unit_load<27>(sp, tile, TS, mb, ck, M, KC, sm, skc, st, kc_inner);
        else if (ntaps == 8) pack_unit_load<8>(sp, tile, TS, mb, ck, M, KC, sm, skc, st, kc_inner);
        else if (ntaps == 1) pack_unit_load<1>(sp, tile, TS, mb, ck, M, KC, sm, skc, st, kc_inner);
        else {
            for (int e = threadIdx.x; e < E; e += NT) {
                const int t = e % ntaps, r = e / ntaps;
                const int kcl = kc_inner ? r % 16 : r / 32, ml = kc_inner ? r / 16 : r % 32;
                const int m = mb * 32 + ml, kc = ck * 16 + kcl;
                const float v = (m < M && kc < KC) ? sp[m * sm + kc * skc + t * st] : 0.f;
                tile[t * TS + ml * 16 + kcl] = (half_t)v;
            }
        }
        __syncthreads();
        half_t* out = dst + d[1] + ((long)mb * nck + ck) * ntaps * 512;
        for (int i = threadIdx.x * 8; i < E; i += NT * 8)
            *reinterpret_cast<half8*>(out + i) = *reinterpret_cast<const half8*>(tile + (i >> 9) * TS + (i & 511));
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void unpack_wgrad_batched_kernel(const float* __restrict__ dwp, float* __restrict__ dst,
                                                                  const long* __restrict__ desc, int n, long total, float scale,
                                                                  int accumulate) {
    __shared__ long firsts[DESC_MAX];
    for (int j = threadIdx.x; j < n; j += NT) firsts[j] = desc[j * DESC_W + 8];
    __syncthreads();
    for (long g = (long)blockIdx.x * NT + threadIdx.x; g < total; g += (long)gridDim.x * NT) {
        const long* d = desc + find_desc(firsts, n, g) * DESC_W;
        const long i = g - d[8];
        const int ntaps = (int)d[5], M = (int)d[6], KC = (int)d[7];
        const int Mpad = (M + 31) & ~31, KCpad = (KC + 31) & ~31;
        const int t = (int)(i % ntaps), kc = (int)((i / ntaps) % KC), m = (int)(i / ((long)ntaps * KC));
        const float v = scale * dwp[d[0] + ((long)t * Mpad + m) * KCpad + kc];
        float* o = dst + d[1] + m * d[2] + kc * d[3] + t * d[4];
        *o = accumulate ? *o + v : v;
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dst, int ntaps, int M, int KC, int Mpad,
                                    int KCpad, long sm, long skc, long st, float scale, int accumulate) {
    const long total = (long)ntaps * M * KC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // iterate in DESTINATION-friendly order: t fastest (conv weights have taps contiguous)
        const int t = (int)(i % ntaps), kc = (int)((i / ntaps) % KC), m = (int)(i / ((long)ntaps * KC));
        const float v = scale * dwp[((long)t * Mpad + m) * KCpad + kc];
        float* d = dst + m * sm + kc * skc + t * st;
        *d = accumulate ? *d + v : v;
    }
}

// Flat-arena reductions are deterministic: every block writes its partial sum to the caller's scratch and ONE block adds
// the partials in a fixed order (fp64 atomics from ~2000 blocks onto one address serialise in L2, 50-100 ns each, and
// make the clip coefficient differ between data-parallel ranks in the last bit).  All flat kernels move 16 bytes per lane.

__device__ __forceinline__ void block_sum_d(double (&v)[2], double* sm) {      // fixed-order block reduction (256 threads)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = wave_sum_d(v[i]);
    __syncthreads();
    if (lane == 0) { sm[wid * 2] = v[0]; sm[wid * 2 + 1] = v[1]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        v[0] = v[1] = 0.0;
        for (int w = 0; w < NT / 64; ++w) { v[0] += sm[w * 2]; v[1] += sm[w * 2 + 1]; }
    }
}

// out[k] (+)= sum_b partial[2 b + k], k = 0, 1
__global__ __launch_bounds__(NT) void reduce_partials_kernel(const double* __restrict__ partial, int nblocks, double* out, int accumulate) {
    __shared__ double sm[2 * (NT / 64)];
    double v[2] = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += NT) { v[0] += partial[2 * b]; v[1] += partial[2 * b + 1]; }
    block_sum_d(v, sm);
    if (threadIdx.x == 0) {
        out[0] = (accumulate ? out[0] : 0.0) + v[0];
        out[1] = (accumulate ? out[1] : 0.0) + v[1];
    }
}

__global__ __launch_bounds__(NT) void ewc_fwd_kernel(const float* __restrict__ th, const float* __restrict__ ts,
                                                     const float* __restrict__ f, long n, double* partial) {
    __shared__ double sm[2 * (NT / 64)];
    float acc = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        const floatx4 a = reinterpret_cast<const floatx4*>(th)[i], b = reinterpret_cast<const floatx4*>(ts)[i],
                      c = reinterpret_cast<const floatx4*>(f)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[e] - b[e]; acc += c[e] * d * d; }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float d = th[i] - ts[i];
        acc += f[i] * d * d;
    }
    double v[2] = {(double)acc, 0.0};
    block_sum_d(v, sm);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = v[0]; partial[2 * blockIdx.x + 1] = 0.0; }
}
__global__ void ewc_finalize_kernel(const double* ws, float lambda, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(0.5 * (double)lambda * ws[0]);
}
__global__ __launch_bounds__(NT) void ewc_bwd_kernel(const float* __restrict__ th, const float* __restrict__ ts,
                                                     const float* __restrict__ f, long n, float coef,
                                                     const float* __restrict__ coef_dev, float* __restrict__ g, int vec) {
    if (coef_dev) coef *= coef_dev[0];
    const long n4 = vec ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        const floatx4 a = reinterpret_cast<const floatx4*>(th)[i], b = reinterpret_cast<const floatx4*>(ts)[i],
                      c = reinterpret_cast<const floatx4*>(f)[i];
        floatx4 o = reinterpret_cast<floatx4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += coef * c[e] * (a[e] - b[e]);
        reinterpret_cast<floatx4*>(g)[i] = o;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) g[i] += coef * f[i] * (th[i] - ts[i]);
}

template <int MODE>  // 0 square, 1 accumulate, 2 ema
__global__ __launch_bounds__(NT) void fisher_kernel(const float* __restrict__ g, float* __restrict__ f, long n, float unscale,
                                                    float a, int vec) {
    const long n4 = vec ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        const floatx4 gv = reinterpret_cast<const floatx4*>(g)[i];
        floatx4 fv = MODE == 0 ? floatx4{0, 0, 0, 0} : reinterpret_cast<floatx4*>(f)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = gv[e] * unscale, sq = x * x;
            if (MODE == 0) fv[e] = sq;
            else if (MODE == 1) fv[e] += a * sq;
            else fv[e] = a * sq + (1.f - a) * fv[e];
        }
        reinterpret_cast<floatx4*>(f)[i] = fv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float x = g[i] * unscale, sq = x * x;
        if (MODE == 0) f[i] = sq;
        else if (MODE == 1) f[i] += a * sq;
        else f[i] = a * sq + (1.f - a) * f[i];
    }
}

// Four independent 16-byte loads per lane and iteration (round 4: 29.2 -> 23.0 us for the 125 MB arena, 4.28 -> 5.43 TB/s).  The
// partial pairs are added by reduce_partials_kernel in index order (bit-reproducible).  Folding that second launch into this
// kernel (ticket counter, the block that draws the last ticket adds the partials) was measured and rejected: the device-scope
// release every block needs across the 8 XCD L2s cost 10 us, twice what the dependent launch costs
// (profiles/r04_reduction_variants.txt).
__global__ __launch_bounds__(NT) void gradnorm_kernel(const float* __restrict__ g, long n, float unscale, double* partial, int vec) {
    __shared__ double sm[2 * (NT / 64)];
    float acc[2] = {0.f, 0.f};
    const long n4 = vec ? n >> 2 : 0;
    const long stride = (long)gridDim.x * NT;
    auto take = [&](const floatx4 gv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = gv[e] * unscale;
            if (!isfinite(x)) acc[1] += 1.f; else acc[0] += x * x;
        }
    };
    long i = (long)blockIdx.x * NT + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const floatx4 a = reinterpret_cast<const floatx4*>(g)[i], b = reinterpret_cast<const floatx4*>(g)[i + stride],
                      c = reinterpret_cast<const floatx4*>(g)[i + 2 * stride], d = reinterpret_cast<const floatx4*>(g)[i + 3 * stride];
        take(a); take(b); take(c); take(d);
    }
    for (; i < n4; i += stride) take(reinterpret_cast<const floatx4*>(g)[i]);
    for (long j = (n4 << 2) + (long)blockIdx.x * NT + threadIdx.x; j < n; j += stride) {
        const float x = g[j] * unscale;
        if (!isfinite(x)) acc[1] += 1.f; else acc[0] += x * x;
    }
    double v[2] = {(double)acc[0], (double)acc[1]};
    block_sum_d(v, sm);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = v[0]; partial[2 * blockIdx.x + 1] = v[1]; }
}

// torch.optim.SGD(nesterov=True, dampening=0): g += wd*theta; buf = g (first) | mu*buf + g; theta -= lr*(g + mu*buf)
__global__ __launch_bounds__(NT) void sgd_kernel(float* __restrict__ th, float* __restrict__ buf, const float* __restrict__ g,
                                                 long n, float lr, float mu, float wd, float gs, int first) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float t = th[i];
        const float d = g[i] * gs + wd * t;
        const float b = first ? d : mu * buf[i] + d;
        buf[i] = b;
        th[i] = t - lr * (d + mu * b);
    }
}

// Same update, but unscale / clip factor / skip decision come from device memory (no host sync):
//   ctrl[0] = sum (g*inv_scale)^2, ctrl[1] = #non-finite  (lnn_gradnorm_sumsq)
//   coef = min(1, max_norm / (sqrt(ctrl[0]) + 1e-6))   (torch.nn.utils.clip_grad_norm_, MH.py:629)
//   any non-finite gradient -> the whole step is skipped (GradScaler.step semantics, MH.py:630)
__global__ __launch_bounds__(NT) void sgd_clipped_kernel(float* __restrict__ th, float* __restrict__ buf,
                                                         const float* __restrict__ g, long n, float lr, float mu, float wd,
                                                         float inv_scale, float max_norm, const double* __restrict__ ctrl, int vec) {
    if (ctrl[1] > 0.0) return;
    float gs = inv_scale;
    if (max_norm > 0.f) {
        const float coef = max_norm / ((float)sqrt(ctrl[0]) + 1e-6f);
        if (coef < 1.f) gs *= coef;
    }
    const long n4 = vec ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        floatx4 t = reinterpret_cast<floatx4*>(th)[i], b = reinterpret_cast<floatx4*>(buf)[i];
        const floatx4 gv = reinterpret_cast<const floatx4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = gv[e] * gs + wd * t[e];
            b[e] = mu * b[e] + d;
            t[e] = t[e] - lr * (d + mu * b[e]);
        }
        reinterpret_cast<floatx4*>(buf)[i] = b;
        reinterpret_cast<floatx4*>(th)[i] = t;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float t = th[i];
        const float d = g[i] * gs + wd * t;
        const float b = mu * buf[i] + d;
        buf[i] = b;
        th[i] = t - lr * (d + mu * b);
    }
}

// Riemannian-Walk running statistics (rw/nnUNetTrainerRW.py:231-265) on the flat arena, fused:
//   g      = grad * inv_scale * clip_coef                      (what param.grad holds after unscale_ + clip_grad_norm_)
//   score += max(0, g*(prev-theta) / (0.5*F*(theta-prev)^2 + eps))          (only if a previous snapshot exists)
//   prev   = theta ;  F = alpha*g^2 + (1-alpha)*F
__global__ __launch_bounds__(NT) void rw_update_kernel(const float* __restrict__ th, float* __restrict__ prev,
                                                       const float* __restrict__ g, float* __restrict__ fisher,
                                                       float* __restrict__ score, long n, float inv_scale, float max_norm,
                                                       const double* __restrict__ ctrl, float alpha, float eps, int have_prev) {
    float gs = inv_scale;
    if (ctrl) {
        if (ctrl[1] > 0.0) return;          // non-finite gradient: the optimiser step was skipped, keep the statistics
        if (max_norm > 0.f) {
            const float coef = max_norm / ((float)sqrt(ctrl[0]) + 1e-6f);
            if (coef < 1.f) gs *= coef;
        }
    }
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float t = th[i], gi = g[i] * gs, f = fisher[i];
        if (have_prev) {
            const float d = prev[i] - t;
            const float sc = (gi * d) / (0.5f * f * d * d + eps);
            if (sc > 0.f) score[i] += sc;
        }
        prev[i] = t;
        fisher[i] = alpha * gi * gi + (1.f - alpha) * f;
    }
}

__global__ __launch_bounds__(NT) void cast_kernel(const float* __restrict__ s, half_t* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) d[i] = (half_t)s[i];
}
}  // namespace

extern "C" size_t lnn_packed_weight_elems(int ntaps, int M, int KC) {
    return (size_t)ntaps * lnn_round_up(M, 32) * lnn_round_up(KC, 16);
}

extern "C" int lnn_pack_weights(lnn_stream_t s_, const float* src, void* dst, int ntaps, int M, int KC, long sm, long skc,
                                long st) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(src && dst && lnn_aligned16(dst), "lnn_pack_weights: null/misaligned pointer");
    LNN_REQUIRE(ntaps > 0 && M > 0 && KC > 0, "lnn_pack_weights: bad dims");
    const int Mpad = lnn_round_up(M, 32), KCpad = lnn_round_up(KC, 16);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(flat_blocks((long)ntaps * Mpad * KCpad, 4)), dim3(NT), 0, s, src,
                       (half_t*)dst, ntaps, M, KC, Mpad, KCpad, sm, skc, st);
    LNN_CHECK_LAUNCH("lnn_pack_weights");
    return LNN_OK;
}

extern "C" int lnn_unpack_wgrad(lnn_stream_t s_, const float* dwp, float* dst, int ntaps, int M, int KC, long sm, long skc,
                                long st, float scale, int accumulate) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(dwp && dst, "lnn_unpack_wgrad: null pointer");
    const int Mpad = lnn_round_up(M, 32), KCpad = lnn_round_up(KC, 32);
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(flat_blocks((long)ntaps * M * KC, 4)), dim3(NT), 0, s, dwp, dst, ntaps, M,
                       KC, Mpad, KCpad, sm, skc, st, scale, accumulate);
    LNN_CHECK_LAUNCH("lnn_unpack_wgrad");
    return LNN_OK;
}

extern "C" int lnn_pack_weights_batched(lnn_stream_t s_, const float* src_base, void* dst_base, const long* desc_dev, int n,
                                        long total) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(src_base && dst_base && desc_dev && lnn_aligned16(dst_base), "lnn_pack_weights_batched: null/misaligned pointer");
    LNN_REQUIRE(n > 0 && n <= DESC_MAX && total > 0, "lnn_pack_weights_batched: 1..%d descriptors, total > 0", DESC_MAX);
    LNN_REQUIRE(total % 8 == 0, "lnn_pack_weights_batched: total %ld is not a sum of padded panel sizes", total);
    static int tiled = -1;          // LNN_PACK_ELEMENTWISE=1: the round-1 gather kernel (A/B measurements)
    if (tiled < 0) { const char* e = getenv("LNN_PACK_ELEMENTWISE"); tiled = (e && e[0] == '1') ? 0 : 1; }
    if (tiled) hipLaunchKernelGGL(pack_weights_tiled_kernel, dim3(2048), dim3(NT), 0, s, src_base, (half_t*)dst_base, desc_dev, n);
    else hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(flat_blocks(total / 8, 2)), dim3(NT), 0, s, src_base, (half_t*)dst_base,
                            desc_dev, n, total);
    LNN_CHECK_LAUNCH("lnn_pack_weights_batched");
    return LNN_OK;
}


namespace {
// Tiled unpack: the element-wise kernel reads the fp32 panel [tap][Mpad][KCpad] with a stride of Mpad*KCpad between the taps
// of a destination run.  A block owns 8 rows x 32 channels x all taps: panel reads in 128-byte segments, LDS transpose,
// destination writes in DESTINATION order (runs of 32 x ntaps or 8 x ntaps consecutive floats).
template <int NTAPS>
__device__ __forceinline__ void unpack_unit_store(float* __restrict__ op, const float* tile, int m0, int kc0, int M, int KC, long sm,
                                                  long skc, long st, bool kc_inner, float scale, int accumulate) {
    constexpr int E = 8 * 32 * NTAPS;
    // full units of [m][kc][tap] tensors: a row's 32 x NTAPS destination floats are one contiguous run with the tile's own
    // layout -> 16-byte LDS reads and 16-byte read-modify-writes
    const bool full = kc_inner && m0 + 8 <= M && kc0 + 32 <= KC && st == 1 && skc == NTAPS && (sm & 3) == 0 &&
                      (reinterpret_cast<unsigned long long>(op) & 15) == 0;
    if (NTAPS > 1 && full) {
        constexpr int RL4 = 8 * NTAPS;                       // float4 per row run
        for (int q = threadIdx.x; q < 8 * RL4; q += NT) {
            const int ml = q / RL4, j = q - ml * RL4;
            floatx4* o = reinterpret_cast<floatx4*>(op + (long)(m0 + ml) * sm + (long)kc0 * NTAPS + 4 * j);
            const floatx4 v = *reinterpret_cast<const floatx4*>(tile + ml * 32 * NTAPS + 4 * j) * scale;
            *o = accumulate ? *o + v : v;
        }
        return;
    }
    for (int e = threadIdx.x; e < E; e += NT) {
        const int t = e % NTAPS, r = e / NTAPS;
        const int kcl = kc_inner ? r % 32 : r / 8, ml = kc_inner ? r / 32 : r % 8;
        const int m = m0 + ml, kc = kc0 + kcl;
        if (m < M && kc < KC) {
            float* o = op + m * sm + kc * skc + t * st;
            const float v = scale * tile[(ml * 32 + kcl) * NTAPS + t];
            *o = accumulate ? *o + v : v;
        }
    }
}

__global__ __launch_bounds__(NT) void unpack_wgrad_tiled_kernel(const float* __restrict__ dwp, float* __restrict__ dst,
                                                                const long* __restrict__ desc, int n, float scale, int accumulate) {
    __shared__ long ufirst[DESC_MAX + 1];
    __shared__ __attribute__((aligned(16))) float tile[8 * 32 * 27];
    if (threadIdx.x == 0) {
        long acc = 0;
        for (int j = 0; j < n; ++j) {
            ufirst[j] = acc;
            const long M = desc[j * DESC_W + 6], KC = desc[j * DESC_W + 7];
            acc += ((M + 7) >> 3) * ((KC + 31) >> 5);
        }
        ufirst[n] = acc;
    }
    __syncthreads();
    const long units = ufirst[n];
    for (long u = blockIdx.x; u < units; u += gridDim.x) {
        const int j = find_desc(ufirst, n, u);
        const long* d = desc + j * DESC_W;
        const int ntaps = (int)d[5], M = (int)d[6], KC = (int)d[7];
        const int Mpad = (M + 31) & ~31, KCpad = (KC + 31) & ~31, nck = (KC + 31) >> 5;
        const long lu = u - ufirst[j];
        const int m0 = (int)(lu / nck) * 8, kc0 = (int)(lu % nck) * 32;
        const int E = 8 * 32 * ntaps;
        const float* pp = dwp + d[0];
        if (m0 + 8 <= M && (reinterpret_cast<unsigned long long>(pp) & 15) == 0) {       // panel rows are padded to 32 channels
            for (int e4 = threadIdx.x; e4 < E / 4; e4 += NT) {
                const int kcl = (e4 & 7) * 4, ml = (e4 >> 3) & 7, t = e4 >> 6;
                const floatx4 v = *reinterpret_cast<const floatx4*>(pp + ((long)t * Mpad + m0 + ml) * KCpad + kc0 + kcl);
#pragma unroll
                for (int k = 0; k < 4; ++k) tile[(ml * 32 + kcl + k) * ntaps + t] = v[k];
            }
        } else {
            for (int e = threadIdx.x; e < E; e += NT) {
                const int kcl = e & 31, ml = (e >> 5) & 7, t = e >> 8;
                const int m = m0 + ml, kc = kc0 + kcl;
                tile[(ml * 32 + kcl) * ntaps + t] = (m < M && kc < KC) ? pp[((long)t * Mpad + m) * KCpad + kc] : 0.f;
            }
        }
        __syncthreads();
        const long sm = d[2], skc = d[3], st = d[4];
        const bool kc_inner = skc < sm;
        float* op = dst + d[1];
        if (ntaps == 27) unpack_unit_store<27>(op, tile, m0, kc0, M, KC, sm, skc, st, kc_inner, scale, accumulate);
        else if (ntaps == 8) unpack_unit_store<8>(op, tile, m0, kc0, M, KC, sm, skc, st, kc_inner, scale, accumulate);
        else if (ntaps == 1) unpack_unit_store<1>(op, tile, m0, kc0, M, KC, sm, skc, st, kc_inner, scale, accumulate);
        else {
            for (int e = threadIdx.x; e < E; e += NT) {
                const int t = e % ntaps, r = e / ntaps;
                const int kcl = kc_inner ? r % 32 : r / 8, ml = kc_inner ? r / 32 : r % 8;
                const int m = m0 + ml, kc = kc0 + kcl;
                if (m < M && kc < KC) {
                    float* o = op + m * sm + kc * skc + t * st;
                    const float v = scale * tile[(ml * 32 + kcl) * ntaps + t];
                    *o = accumulate ? *o + v : v;
                }
            }
        }
        __syncthreads();
    }
}
}  // namespace

extern "C" int lnn_unpack_wgrad_batched(lnn_stream_t s_, const float* panel_base, float* dst_base, const long* desc_dev, int n,
                                        long total, float scale, int accumulate) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(panel_base && dst_base && desc_dev, "lnn_unpack_wgrad_batched: null pointer");
    LNN_REQUIRE(n > 0 && n <= DESC_MAX && total > 0, "lnn_unpack_wgrad_batched: 1..%d descriptors, total > 0", DESC_MAX);
    static int tiled = -1;          // LNN_PACK_ELEMENTWISE=1: the round-1 gather kernel (A/B measurements)
    if (tiled < 0) { const char* e = getenv("LNN_PACK_ELEMENTWISE"); tiled = (e && e[0] == '1') ? 0 : 1; }
    if (tiled) hipLaunchKernelGGL(unpack_wgrad_tiled_kernel, dim3(2048), dim3(NT), 0, s, panel_base, dst_base, desc_dev, n, scale, accumulate);
    else hipLaunchKernelGGL(unpack_wgrad_batched_kernel, dim3(flat_blocks(total, 4)), dim3(NT), 0, s, panel_base, dst_base, desc_dev,
                            n, total, scale, accumulate);
    LNN_CHECK_LAUNCH("lnn_unpack_wgrad_batched");
    return LNN_OK;
}

extern "C" int lnn_ewc_penalty_fwd(lnn_stream_t s_, const float* theta, const float* theta_star, const float* fisher, long n,
                                   float lambda, float* out, double* ws) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(theta && theta_star && fisher && out && ws, "lnn_ewc_penalty_fwd: null pointer");
    LNN_REQUIRE(lnn_aligned16(theta) && lnn_aligned16(theta_star) && lnn_aligned16(fisher), "lnn_ewc_penalty_fwd: arenas must be 16-byte aligned");
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(ewc_fwd_kernel, dim3(nb), dim3(NT), 0, s, theta, theta_star, fisher, n, ws + 2);
    LNN_CHECK_LAUNCH("lnn_ewc_penalty_fwd");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(NT), 0, s, ws + 2, nb, ws, 0);
    hipLaunchKernelGGL(ewc_finalize_kernel, dim3(1), dim3(64), 0, s, ws, lambda, out);
    LNN_CHECK_LAUNCH("lnn_ewc_penalty_fwd(finalize)");
    return LNN_OK;
}

extern "C" int lnn_ewc_penalty_bwd(lnn_stream_t s_, const float* theta, const float* theta_star, const float* fisher, long n,
                                   float lambda, float gscale, const float* gscale_dev, float* grad) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(theta && theta_star && fisher && grad, "lnn_ewc_penalty_bwd: null pointer");
    const int vec = lnn_aligned16(theta) && lnn_aligned16(theta_star) && lnn_aligned16(fisher) && lnn_aligned16(grad);
    hipLaunchKernelGGL(ewc_bwd_kernel, dim3(flat_blocks(n, 16)), dim3(NT), 0, s, theta, theta_star, fisher, n, lambda * gscale,
                       gscale_dev, grad, vec);
    LNN_CHECK_LAUNCH("lnn_ewc_penalty_bwd");
    return LNN_OK;
}

extern "C" int lnn_fisher_square(lnn_stream_t s_, const float* grad, float* fisher, long n, float unscale) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(grad && fisher, "lnn_fisher_square: null pointer");
    hipLaunchKernelGGL((fisher_kernel<0>), dim3(flat_blocks(n, 16)), dim3(NT), 0, s, grad, fisher, n, unscale, 0.f,
                       (int)(lnn_aligned16(grad) && lnn_aligned16(fisher)));
    LNN_CHECK_LAUNCH("lnn_fisher_square");
    return LNN_OK;
}
extern "C" int lnn_fisher_accumulate(lnn_stream_t s_, const float* grad, float* fisher, long n, float unscale, float weight) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(grad && fisher, "lnn_fisher_accumulate: null pointer");
    hipLaunchKernelGGL((fisher_kernel<1>), dim3(flat_blocks(n, 16)), dim3(NT), 0, s, grad, fisher, n, unscale, weight,
                       (int)(lnn_aligned16(grad) && lnn_aligned16(fisher)));
    LNN_CHECK_LAUNCH("lnn_fisher_accumulate");
    return LNN_OK;
}
extern "C" int lnn_fisher_ema(lnn_stream_t s_, const float* grad, float* fisher, long n, float unscale, float alpha) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(grad && fisher, "lnn_fisher_ema: null pointer");
    hipLaunchKernelGGL((fisher_kernel<2>), dim3(flat_blocks(n, 16)), dim3(NT), 0, s, grad, fisher, n, unscale, alpha,
                       (int)(lnn_aligned16(grad) && lnn_aligned16(fisher)));
    LNN_CHECK_LAUNCH("lnn_fisher_ema");
    return LNN_OK;
}

extern "C" long lnn_flat_reduce_ws_doubles(void) { return 2 + 2 * RED_BLOCKS; }       // the pair, then the block partials

extern "C" int lnn_gradnorm_sumsq(lnn_stream_t s_, const float* grad, long n, float unscale, double* out2, int zero_first) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(grad && out2, "lnn_gradnorm_sumsq: null pointer");
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(gradnorm_kernel, dim3(nb), dim3(NT), 0, s, grad, n, unscale, out2 + 2, (int)lnn_aligned16(grad));
    LNN_CHECK_LAUNCH("lnn_gradnorm_sumsq");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(NT), 0, s, out2 + 2, nb, out2, zero_first ? 0 : 1);
    LNN_CHECK_LAUNCH("lnn_gradnorm_sumsq(reduce)");
    return LNN_OK;
}

extern "C" int lnn_sgd_nesterov_step(lnn_stream_t s_, float* theta, float* buf, const float* grad, long n, float lr,
                                     float momentum, float weight_decay, float grad_scale, int first_step) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(theta && buf && grad, "lnn_sgd_nesterov_step: null pointer");
    hipLaunchKernelGGL(sgd_kernel, dim3(flat_blocks(n, 8)), dim3(NT), 0, s, theta, buf, grad, n, lr, momentum, weight_decay,
                       grad_scale, first_step);
    LNN_CHECK_LAUNCH("lnn_sgd_nesterov_step");
    return LNN_OK;
}

extern "C" int lnn_sgd_nesterov_step_clipped(lnn_stream_t s_, float* theta, float* buf, const float* grad, long n, float lr,
                                             float momentum, float weight_decay, float inv_scale, float max_norm,
                                             const double* ctrl) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(theta && buf && grad && ctrl, "lnn_sgd_nesterov_step_clipped: null pointer");
    hipLaunchKernelGGL(sgd_clipped_kernel, dim3(flat_blocks(n, 16)), dim3(NT), 0, s, theta, buf, grad, n, lr, momentum,
                       weight_decay, inv_scale, max_norm, ctrl, (int)(lnn_aligned16(theta) && lnn_aligned16(buf) && lnn_aligned16(grad)));
    LNN_CHECK_LAUNCH("lnn_sgd_nesterov_step_clipped");
    return LNN_OK;
}

extern "C" int lnn_rw_update(lnn_stream_t s_, const float* theta, float* prev, const float* grad, float* fisher, float* score,
                             long n, float inv_scale, float max_norm, const double* ctrl, float alpha, float eps,
                             int have_prev) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(theta && prev && grad && fisher && score, "lnn_rw_update: null pointer");
    LNN_REQUIRE(alpha > 0.f && alpha <= 1.f, "lnn_rw_update: alpha %g outside (0, 1]", (double)alpha);
    if (n <= 0) return LNN_OK;
    hipLaunchKernelGGL(rw_update_kernel, dim3(flat_blocks(n, 8)), dim3(NT), 0, s, theta, prev, grad, fisher, score, n, inv_scale,
                       max_norm, ctrl, alpha, eps, have_prev);
    LNN_CHECK_LAUNCH("lnn_rw_update");
    return LNN_OK;
}

extern "C" int lnn_cast_f32_to_h(lnn_stream_t s_, const float* src, void* dst, long n) {
    hipStream_t s = (hipStream_t)s_;
    LNN_REQUIRE(src && dst, "lnn_cast_f32_to_h: null pointer");
    hipLaunchKernelGGL(cast_kernel, dim3(flat_blocks(n, 8)), dim3(NT), 0, s, src, (half_t*)dst, n);
    LNN_CHECK_LAUNCH("lnn_cast_f32_to_h");
    return LNN_OK;
}
