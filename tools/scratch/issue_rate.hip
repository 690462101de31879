// Microbenchmark: instruction issue rate per SIMD on gfx950 for VALU / SALU / mixed streams at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void k(uint32_t* out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // 32 independent VALU
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7\n"
                             "v_xor_b32 %1, %1, %0\n v_xor_b32 %3, %3, %2\n v_xor_b32 %5, %5, %4\n v_xor_b32 %7, %7, %6\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 1) { // 32 dependent VALU (single chain)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                             "v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                             : "+v"(a0), "+v"(a1));
            }
        } else if (MODE == 2) { // 32 independent SALU
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3\n s_xor_b32 %1, %1, %0\n s_xor_b32 %3, %3, %2\n"
                             "s_add_u32 %0, %0, %1\n s_add_u32 %2, %2, %3\n s_xor_b32 %1, %1, %0\n s_xor_b32 %3, %3, %2\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
            }
        } else { // 16 VALU + 16 SALU interleaved
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_add_u32 %2, %2, %3\n s_xor_b32 %5, %5, %4\n"
                             "v_xor_b32 %1, %1, %0\n s_add_u32 %4, %4, %5\n v_xor_b32 %3, %3, %2\n s_xor_b32 %5, %5, %4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : : "scc");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
}
template <int MODE>
void run(const char* name, uint32_t* d) {
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 256 * wps > 1024 ? 1024 : 256 * wps; // waves per SIMD = threads/256 (one block per CU) 
        const int blocks = 256 * (256 * wps / threads);
        const int iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<MODE><<<blocks, threads>>>(d, 10);
        hipEventRecord(e0);
        k<MODE><<<blocks, threads>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = 32.0 * iters;
        const double waves_per_simd = wps;
        // cycles at ~2.1 GHz
        const double cyc = ms * 1e-3 * 2.1e9;
        printf("%-28s %d waves/SIMD: %.3f ms -> %.2f cycles per instr per wave, %.2f cycles per instr per SIMD (at 2.1 GHz)\n",
               name, wps, ms, cyc / instr_per_wave, cyc / (instr_per_wave * waves_per_simd));
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 8 * 4);
    run<0>("VALU independent", d);
    run<1>("VALU dependent chain", d);
    run<2>("SALU independent", d);
    run<3>("VALU+SALU interleaved", d);
    return 0;
}
