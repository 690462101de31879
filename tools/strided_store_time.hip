// How fast 1 GiB is WRITTEN with the store shapes a decoder could use (gfx950), nothing read:
//  (a) coalesced 16-byte stores (the fill kernel's shape)
//  (b) 248-byte rows of 4-byte stores, 16 per 3968-byte segment  (decode_expand_kernel / decode_tile_kernel today)
//  (c) lane-sequential, one segment per wave: lane pair m writes 124 bytes at 124 m -- even lane 64 bytes (4 x 16),
//      odd lane 60 bytes (3 x 16 + 8 + 4), addresses 4-byte aligned only
//  (d) lane-sequential, two segments per wave: every lane 124 bytes (7 x 16 + 8 + 4) at 124 lane
// hipcc --offload-arch=gfx950 -O3 -o strided_store_time strided_store_time.hip && ./strided_store_time
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x27000);
}
constexpr u32 kSegWords = 992;

__global__ __launch_bounds__(256) void fill_a(u32 *out, u32 n_seg) {
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    const u32x4 v = {lane, lane + 1, lane + 2, lane + 3};
    for (u32 s = wave; s < n_seg; s += waves) {
        const __amdgpu_buffer_rsrc_t r = rsrc(out + (uint64_t)s * kSegWords, kSegWords * 4u);
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, r, lane * 16u + 1024u * k, 0, 0); // the 4th is cut at 3968
    }
}
__global__ __launch_bounds__(256) void rows_b(u32 *out, u32 n_seg) {
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    u32 soff = (lane & 31u) != 31u ? (lane - (lane >> 5)) * 4u : 0xFFFFF000u;
    for (u32 s = wave; s < n_seg; s += waves) {
        const __amdgpu_buffer_rsrc_t r = rsrc(out + (uint64_t)s * kSegWords, kSegWords * 4u);
#pragma unroll
        for (int k = 0; k < 16; ++k) __builtin_amdgcn_raw_buffer_store_b32(lane + k, r, soff + 248u * k, 0, 0);
    }
}
__global__ __launch_bounds__(256) void pairs_c(u32 *out, u32 n_seg) {
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    const u32x4 v = {lane, lane + 1, lane + 2, lane + 3};
    const u32 off = (lane >> 1) * 124u + (lane & 1u) * 64u;
    const bool odd = lane & 1u;
    for (u32 s = wave; s < n_seg; s += waves) {
        const __amdgpu_buffer_rsrc_t r = rsrc(out + (uint64_t)s * kSegWords, kSegWords * 4u);
#pragma unroll
        for (int k = 0; k < 3; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, r, off + 16u * k, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{lane, lane}, r, off + 48u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(lane, r, off + 56u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(lane, r, odd ? 0xFFFFF000u : off + 60u, 0, 0);
    }
}
__global__ __launch_bounds__(256) void lanes_d(u32 *out, u32 n_seg) {
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    const u32x4 v = {lane, lane + 1, lane + 2, lane + 3};
    const u32 off = lane * 124u;
    for (u32 s = 2u * wave; s < n_seg; s += 2u * waves) {
        const __amdgpu_buffer_rsrc_t r = rsrc(out + (uint64_t)s * kSegWords, (s + 1 < n_seg ? 2u : 1u) * kSegWords * 4u);
#pragma unroll
        for (int k = 0; k < 7; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, r, off + 16u * k, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{lane, lane}, r, off + 112u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(lane, r, off + 120u, 0, 0);
    }
}

int main() {
    const u32 n_seg = 270600; // 1 GiB
    u32 *out;
    hipMalloc(&out, (size_t)n_seg * kSegWords * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    struct { const char *name; void (*k)(u32 *, u32); } ks[] = {
        {"(a) coalesced 16-byte stores", fill_a}, {"(b) 248-byte rows of 4-byte stores", rows_b},
        {"(c) lane pairs: 64 + 60 bytes at 124 m", pairs_c}, {"(d) lanes: 124 bytes at 124 lane, two segments", lanes_d}};
    for (u32 wgs : {2048u, 4096u, 8192u}) {
        for (auto &k : ks) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k.k, dim3(wgs), dim3(256), 0, 0, out, n_seg);
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k.k, dim3(wgs), dim3(256), 0, 0, out, n_seg);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 20;
            printf("%5u workgroups  %-50s %.4f ms  %.0f GB/s\n", wgs, k.name, ms, (double)n_seg * kSegWords * 4 / ms / 1e6);
        }
    }
    return 0;
}
