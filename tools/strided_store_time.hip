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
// (a) with nontemporal stores
__global__ __launch_bounds__(256) void fill_a_nt(u32 *out, u32 n_seg) {
    const u32 lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    const u32x4 v = {lane, lane + 1, lane + 2, lane + 3};
    for (u32 s = wave; s < n_seg; s += waves) {
        const __amdgpu_buffer_rsrc_t r = rsrc(out + (uint64_t)s * kSegWords, kSegWords * 4u);
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, r, lane * 16u + 1024u * k, 0, 2);
    }
}
// (a) as a plain grid-stride fill: thread t writes 16 bytes at 16 t, 16 (t + threads), ...
__global__ __launch_bounds__(256) void fill_plain(u32 *out, u32 n_seg) {
    const size_t n16 = (size_t)n_seg * kSegWords / 4, step = (size_t)gridDim.x * blockDim.x;
    const u32x4 v = {1, 2, 3, 4};
    u32x4 *o = reinterpret_cast<u32x4 *>(out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += step) o[i] = v;
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

// one-shot forms: one wavefront per segment, no loop (grid = segments / 4)
__global__ __launch_bounds__(256) void fill_a_once(u32 *out, u32 n_seg) {
    const u32 lane = threadIdx.x & 63u, s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= n_seg) return;
    const u32x4 v = {lane, lane + 1, lane + 2, lane + 3};
    const __amdgpu_buffer_rsrc_t r = rsrc(out + (uint64_t)s * kSegWords, kSegWords * 4u);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b128(v, r, lane * 16u + 1024u * k, 0, 0);
}
__global__ __launch_bounds__(256) void rows_b_once(u32 *out, u32 n_seg) {
    const u32 lane = threadIdx.x & 63u, s = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (s >= n_seg) return;
    u32 soff = (lane & 31u) != 31u ? (lane - (lane >> 5)) * 4u : 0xFFFFF000u;
    const __amdgpu_buffer_rsrc_t r = rsrc(out + (uint64_t)s * kSegWords, kSegWords * 4u);
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_amdgcn_raw_buffer_store_b32(lane + k, r, soff + 248u * k, 0, 0);
}
// a workgroup of 256 threads writes 16 KiB: thread t 16 bytes at 16 t + 4096 k (the shape of torch's fill)
__global__ __launch_bounds__(256) void fill_chunks_once(u32 *out, u32 n_seg) {
    const size_t n16 = (size_t)n_seg * kSegWords / 4;
    const u32x4 v = {1, 2, 3, 4};
    u32x4 *o = reinterpret_cast<u32x4 *>(out);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = (size_t)blockIdx.x * 1024u + 256u * k + threadIdx.x;
        if (i < n16) o[i] = v;
    }
}

// ONE 16-byte store per thread, 4 KiB per workgroup: the launch shape of torch's fill (rocprofv3: 256 x 262144, 4 registers)
__global__ __launch_bounds__(256) void fill_one_store(u32 *out, u32 n_seg) {
    const size_t n16 = (size_t)n_seg * kSegWords / 4, i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) reinterpret_cast<u32x4 *>(out)[i] = u32x4{(u32)i, 2, 3, 4};
}
// ... the same with one constant in every word (what a fill writes)
__global__ __launch_bounds__(256) void fill_chunks_const(u32 *out, u32 n_seg) {
    const size_t n16 = (size_t)n_seg * kSegWords / 4;
    const u32x4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    u32x4 *o = reinterpret_cast<u32x4 *>(out);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = (size_t)blockIdx.x * 1024u + 256u * k + threadIdx.x;
        if (i < n16) o[i] = v;
    }
}
// ... and with words that differ from lane to lane and from store to store (what a decoder writes)
__global__ __launch_bounds__(256) void fill_chunks_mixed(u32 *out, u32 n_seg) {
    const size_t n16 = (size_t)n_seg * kSegWords / 4;
    u32x4 *o = reinterpret_cast<u32x4 *>(out);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t i = (size_t)blockIdx.x * 1024u + 256u * k + threadIdx.x;
        const u32 h = (u32)i * 2654435761u;
        if (i < n16) o[i] = u32x4{h, h ^ (h >> 7), h * 3u, ~h};
    }
}

int main() {
    const u32 n_seg = 270600; // 1 GiB
    u32 *out;
    hipMalloc(&out, (size_t)n_seg * kSegWords * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    struct { const char *name; void (*k)(u32 *, u32); } ks[] = {
        {"(a) coalesced 16-byte stores", fill_a}, {"(a) nontemporal", fill_a_nt}, {"(a) plain grid-stride fill", fill_plain}, {"(b) 248-byte rows of 4-byte stores", rows_b},
        {"(c) lane pairs: 64 + 60 bytes at 124 m", pairs_c}, {"(d) lanes: 124 bytes at 124 lane, two segments", lanes_d}};
    for (u32 wgs : {2048u}) {
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
    struct { const char *name; void (*k)(u32 *, u32); u32 grid; } once[] = {
        {"(a) one wavefront per segment, no loop", fill_a_once, (n_seg + 3) / 4}, {"(b) one wavefront per segment, no loop", rows_b_once, (n_seg + 3) / 4},
        {"16 KiB per workgroup, four stores per thread, no loop", fill_chunks_once, (u32)(((size_t)n_seg * kSegWords / 4 + 1023) / 1024)},
        {"4 KiB per workgroup, one store per thread", fill_one_store, (u32)(((size_t)n_seg * kSegWords / 4 + 255) / 256)},
        {"... one constant in every word", fill_chunks_const, (u32)(((size_t)n_seg * kSegWords / 4 + 1023) / 1024)},
        {"... words that all differ", fill_chunks_mixed, (u32)(((size_t)n_seg * kSegWords / 4 + 1023) / 1024)}};
    for (auto &k : once) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k.k, dim3(k.grid), dim3(256), 0, 0, out, n_seg);
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k.k, dim3(k.grid), dim3(256), 0, 0, out, n_seg);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        printf("%6u workgroups  %-50s %.4f ms  %.0f GB/s\n", k.grid, k.name, ms, (double)n_seg * kSegWords * 4 / ms / 1e6);
    }
    return 0;
}
