// Infinity-Cache (MALL) reuse probe: does a streaming kernel that walks a buffer in the REVERSE order of the kernel that
// touched it last get cache hits on the tail?  Producer = a streaming write or read of X (forward), consumer = a streaming
// read of X forward or reversed; consumer time and effective GB/s per size.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mall_probe tools/probes/mall_probe.hip && tools/probes/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int BLOCK = 256, CHUNK = BLOCK * 4;   // uint4 per thread, 4 per chunk iteration

__global__ void __launch_bounds__(BLOCK) fill_kernel(uint4* x, long nchunks) {
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[c * CHUNK + j * BLOCK + threadIdx.x] = make_uint4(c, j, 1, 2);
    }
}
template <bool REV>
__global__ void __launch_bounds__(BLOCK) read_kernel(const uint4* __restrict__ x, long nchunks, unsigned* out) {
    unsigned acc = 0;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long cc = REV ? nchunks - 1 - c : c;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = x[cc * CHUNK + j * BLOCK + threadIdx.x];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool REV>
__global__ void __launch_bounds__(BLOCK) copy_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, long nchunks) {
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long cc = REV ? nchunks - 1 - c : c;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = x[cc * CHUNK + j * BLOCK + threadIdx.x];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[cc * CHUNK + j * BLOCK + threadIdx.x] = v[j];
    }
}

int main() {
    const long maxb = 1280L << 20;
    uint4 *x, *y; unsigned* out;
    CK(hipMalloc(&x, maxb)); CK(hipMalloc(&y, maxb)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    printf("size_MB  producer  consumer      ms    GB/s(consumer bytes)\n");
    for (long mb : {64L, 128L, 192L, 256L, 315L, 448L, 630L, 945L, 1260L}) {
        const long nchunks = (mb << 20) / (CHUNK * 16);
        for (int prod = 0; prod < 2; ++prod)
            for (int cons = 0; cons < 4; ++cons) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    if (prod == 0) fill_kernel<<<grid, BLOCK>>>(x, nchunks);
                    else read_kernel<false><<<grid, BLOCK>>>(x, nchunks, out);
                    CK(hipEventRecord(e0));
                    switch (cons) {
                        case 0: read_kernel<false><<<grid, BLOCK>>>(x, nchunks, out); break;
                        case 1: read_kernel<true><<<grid, BLOCK>>>(x, nchunks, out); break;
                        case 2: copy_kernel<false><<<grid, BLOCK>>>(x, y, nchunks); break;
                        case 3: copy_kernel<true><<<grid, BLOCK>>>(x, y, nchunks); break;
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                const double bytes = double(nchunks) * CHUNK * 16 * (cons >= 2 ? 2 : 1);
                printf("%6ld   %-8s  %-11s %7.3f  %7.0f\n", mb, prod ? "read" : "write",
                       cons == 0 ? "read fwd" : cons == 1 ? "read rev" : cons == 2 ? "copy fwd" : "copy rev", best, bytes / best * 1e-6);
            }
    }
    return 0;
}
