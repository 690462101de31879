// How fast 0.8 GB is COPIED (read + written: 1.6 GB moved, the bytes of the sparse GiB's compress launch) by launch shape:
//  (a) grid-stride loop, 1024 workgroups of 256 (the bench's copy_ceiling kernel)   (b) the same with 4096 / 16384 workgroups
//  (c) one 16-byte load and store per thread, 4 KiB per workgroup, no loop          (d) four per thread, 16 KiB per workgroup, no loop
// hipcc --offload-arch=gfx950 -O3 -o copy_shapes_time copy_shapes_time.hip && ./copy_shapes_time
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_loop(const u32x4 *in, u32x4 *out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy_once(const u32x4 *in, u32x4 *out, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) out[i] = in[i];
}
__global__ __launch_bounds__(256) void copy_once4(const u32x4 *in, u32x4 *out, size_t n16) {
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = (size_t)blockIdx.x * 1024u + 256u * k + threadIdx.x;
        v[k] = i < n16 ? in[i] : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = (size_t)blockIdx.x * 1024u + 256u * k + threadIdx.x;
        if (i < n16) out[i] = v[k];
    }
}

int main() {
    const size_t n16 = 794113824 / 16; // 0.794 GB each way
    u32x4 *in, *out;
    hipMalloc(&in, n16 * 16), hipMalloc(&out, n16 * 16);
    hipMemset(in, 1, n16 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    struct { const char *name; void (*k)(const u32x4 *, u32x4 *, size_t); unsigned grid; } ks[] = {
        {"(a) grid-stride loop, 1024 workgroups", copy_loop, 1024}, {"(b) grid-stride loop, 4096 workgroups", copy_loop, 4096},
        {"(b) grid-stride loop, 16384 workgroups", copy_loop, 16384}, {"(c) one load + store per thread, no loop", copy_once, (unsigned)((n16 + 255) / 256)},
        {"(d) four per thread, no loop", copy_once4, (unsigned)((n16 + 1023) / 1024)}};
    for (int rep = 0; rep < 2; ++rep)
        for (auto &k : ks) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k.k, dim3(k.grid), dim3(256), 0, 0, in, out, n16);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k.k, dim3(k.grid), dim3(256), 0, 0, in, out, n16);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 20;
            printf("%-45s %.4f ms  %.0f GB/s (read + written)\n", k.name, ms, 2.0 * n16 * 16 / ms / 1e6);
        }
    return 0;
}
