#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ uint32_t scan32(uint32_t v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
    return v;
}
__global__ void k(const uint32_t* in, uint32_t* out) { out[threadIdx.x] = scan32(in[threadIdx.x]); }
int main() {
    uint32_t h[64], r[64], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = (i * 7 + 3) % 11;
    hipMalloc(&di, 256); hipMalloc(&dout, 256);
    hipMemcpy(di, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(r, dout, 256, hipMemcpyDeviceToHost);
    uint32_t acc = 0; int bad = 0;
    for (int i = 0; i < 64; ++i) { acc += h[i]; if (r[i] != acc) { if (bad < 8) printf("lane %d got %u want %u\n", i, r[i], acc); bad++; } }
    printf("bad lanes: %d\n", bad);
    return bad != 0;
}
