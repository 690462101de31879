// Hardware facts the pair-layout compress kernel relies on (gfx950), checked once on the GPU box:
//  (1) a 16-byte raw buffer store that straddles the end of the descriptor writes exactly the dwords that lie inside it;
//  (2) 16-byte buffer stores at addresses that are only 4-byte aligned are written correctly;
//  (3) a 16-byte LDS-DMA load (buffer_load_dwordx4 ... lds) past the end of the descriptor deposits zeros.
// hipcc --offload-arch=gfx950 -O3 -o store_bounds store_bounds.hip && ./store_bounds
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x27000);
}

// every lane stores {lane*4+1 ..} at 16*lane through a descriptor of `bytes` bytes based at out + shift words
__global__ void store_k(uint32_t *out, uint32_t shift, uint32_t bytes) {
    const uint32_t l = threadIdx.x;
    const u32x4 v = {4 * l + 1, 4 * l + 2, 4 * l + 3, 4 * l + 4};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(out + shift, bytes), 16 * l, 0, 0);
}

__global__ void load_k(const uint32_t *in, uint32_t shift, uint32_t bytes, uint32_t *out) {
    const uint32_t l = threadIdx.x;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc(in + shift, bytes), 16 * l, 0, 0);
    out[4 * l] = v.x, out[4 * l + 1] = v.y, out[4 * l + 2] = v.z, out[4 * l + 3] = v.w;
}

__global__ void ldsdma_k(const uint32_t *in, uint32_t bytes, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint32_t s[512];
    const uint32_t l = threadIdx.x;
    for (int i = l; i < 512; i += 64) s[i] = 0xDEADBEEFu;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc(in, bytes), (__attribute__((address_space(3))) void *)s, 16, 16 * l, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc(in, bytes), (__attribute__((address_space(3))) void *)(s + 256), 16, 16 * l, 0, 1024, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 512; i += 64) out[i] = s[i];
}

int main() {
    uint32_t *d, *d2;
    hipMalloc(&d, 4096 * 4);
    hipMalloc(&d2, 4096 * 4);
    std::vector<uint32_t> h(4096);
    int bad = 0;
    for (uint32_t shift : {0u, 1u, 2u, 3u})
        for (uint32_t words : {1u, 2u, 3u, 4u, 5u, 6u, 7u, 9u, 63u, 130u, 255u, 256u}) {
            hipMemset(d, 0xEE, 4096 * 4);
            hipLaunchKernelGGL(store_k, dim3(1), dim3(64), 0, 0, d, shift, words * 4);
            hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
            for (uint32_t i = 0; i < 600; ++i) {
                const uint32_t want = (i >= shift && i < shift + words && i - shift < 256) ? i - shift + 1 : 0xEEEEEEEEu;
                if (h[i] != want) {
                    if (bad < 20) printf("STORE shift %u words %u: out[%u] = %08x want %08x\n", shift, words, i, h[i], want);
                    ++bad;
                }
            }
        }
    printf("partial / misaligned 16-byte buffer stores: %s\n", bad ? "MISMATCH" : "dwords inside the descriptor written, nothing else");
    std::vector<uint32_t> src(4096);
    for (int i = 0; i < 4096; ++i) src[i] = i + 1;
    hipMemcpy(d, src.data(), 4096 * 4, hipMemcpyHostToDevice);
    int badl = 0;
    for (uint32_t shift : {0u, 1u, 3u})
        for (uint32_t words : {1u, 2u, 3u, 5u, 63u, 130u}) {
            hipLaunchKernelGGL(load_k, dim3(1), dim3(64), 0, 0, d, shift, words * 4, d2);
            hipMemcpy(h.data(), d2, 256 * 4, hipMemcpyDeviceToHost);
            for (uint32_t i = 0; i < 256; ++i) {
                const uint32_t want = i < words ? shift + i + 1 : 0u;
                if (h[i] != want) {
                    if (badl < 20) printf("LOAD shift %u words %u: v[%u] = %08x want %08x\n", shift, words, i, h[i], want);
                    ++badl;
                }
            }
        }
    printf("partial / misaligned 16-byte buffer loads: %s\n", badl ? "MISMATCH" : "dwords inside the descriptor read, zero behind it");
    int badd = 0;
    for (uint32_t words : {512u, 300u, 130u, 5u, 0u}) {
        hipLaunchKernelGGL(ldsdma_k, dim3(1), dim3(64), 0, 0, d, words * 4, d2);
        hipMemcpy(h.data(), d2, 512 * 4, hipMemcpyDeviceToHost);
        for (uint32_t i = 0; i < 512; ++i) {
            const uint32_t want = i < words ? i + 1 : 0u;
            if (h[i] != want) {
                if (badd < 20) printf("LDSDMA words %u: lds[%u] = %08x want %08x\n", words, i, h[i], want);
                ++badd;
            }
        }
    }
    printf("16-byte LDS-DMA past the descriptor: %s\n", badd ? "MISMATCH (see above)" : "zeros deposited");
    return 0;
}
