// What does a plain copy reach on this chip?  Variants of the 16 B/lane copy used as the on-box ceiling by bench.py.
// hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip && ./copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(const u32x4 *in, u32x4 *out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(in + i + k * stride) : in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], out + i + k * stride); else out[i + k * stride] = v[k];
        }
    }
    for (; i < n16; i += stride) out[i] = in[i];
}
template <int U>
__global__ __launch_bounds__(256) void read_k(const u32x4 *in, u32x4 *out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) acc ^= v[k];
    }
    if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void write_k(u32x4 *out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = u32x4{1, 2, 3, 4};
}
template <typename F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < reps; ++r) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t bytes = 1ull << 30, n16 = bytes / 16;
    u32x4 *in, *out; hipMalloc(&in, bytes); hipMalloc(&out, bytes); hipMemset(in, 1, bytes); hipMemset(out, 0, bytes);
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        float t1 = timeit([&] { hipLaunchKernelGGL((copy_k<1, false>), dim3(grid), dim3(256), 0, 0, in, out, n16); });
        float t4 = timeit([&] { hipLaunchKernelGGL((copy_k<4, false>), dim3(grid), dim3(256), 0, 0, in, out, n16); });
        float t4n = timeit([&] { hipLaunchKernelGGL((copy_k<4, true>), dim3(grid), dim3(256), 0, 0, in, out, n16); });
        float tr = timeit([&] { hipLaunchKernelGGL((read_k<4>), dim3(grid), dim3(256), 0, 0, in, out, n16); });
        float tw = timeit([&] { hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, out, n16); });
        printf("grid %5d: copy u1 %.0f GB/s  u4 %.0f  u4 nontemporal %.0f | read only %.0f GB/s | write only %.0f GB/s\n", grid,
               2.0 * bytes / t1 / 1e6, 2.0 * bytes / t4 / 1e6, 2.0 * bytes / t4n / 1e6, 1.0 * bytes / tr / 1e6, 1.0 * bytes / tw / 1e6);
        fflush(stdout);
    }
    float tm = timeit([&] { hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); });
    printf("hipMemcpy D2D: %.0f GB/s\n", 2.0 * bytes / tm / 1e6);
    return 0;
}
