// How many arrival tickets per microsecond can a launch draw?  (draw_tile: one returning agent-scope atomic add per
// workgroup on ONE address.)  The question behind kernels of very many short-lived ticketed workgroups.
//   mode 0: no atomic at all (the dispatch rate of N empty workgroups of 256 threads)
//   mode 1: thread 0 draws a ticket (returning atomic), barrier, every wave reads it      (draw_tile)
//   mode 2: mode 1 + a second, NON-returning atomic add on another line at the workgroup's end (a "done" counter)
//   mode 3: the ticket drawn by every 4th workgroup only, for itself and the three behind it (blockIdx & 3: what a
//           workgroup of 1024 threads with four roles would cost)
// each with `work` dependent 16-byte loads per thread in front (0 = none) so that the workgroups live for a while.
// hipcc --offload-arch=gfx950 -O3 -o ticket_rate ticket_rate.hip && ./ticket_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *ctr, uint32_t *out, const u32x4 *data, int work) {
    __shared__ uint32_t s_t;
    uint32_t t = blockIdx.x;
    if (MODE == 1 || MODE == 2 || (MODE == 3 && (blockIdx.x & 3u) == 0u)) {
        if (threadIdx.x == 0) s_t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        t = s_t;
    }
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < work; ++i) acc += data[((size_t)(t & 0xFFFFu) * 256u + threadIdx.x + (acc.x & 1u)) + (size_t)i * 65536u * 256u];
    if (threadIdx.x == 0 || acc.x == 0x12345u) out[blockIdx.x] = t + acc.y;
    if (MODE == 2 && threadIdx.x == 0) __hip_atomic_fetch_add(ctr + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
static void run(const char *name, uint32_t *ctr, uint32_t *out, const u32x4 *data, unsigned n, int work) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, ctr, out, data, work);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, ctr, out, data, work);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-46s n %7u work %d  %.4f ms  %.1f workgroups/us  %.1f ns each\n", name, n, work, ms, n / ms / 1e3, ms * 1e6 / n);
    fflush(stdout);
}

int main() {
    uint32_t *ctr, *out;
    u32x4 *data;
    (void)hipMalloc(&ctr, 4096), (void)hipMalloc(&out, 4u << 20), (void)hipMalloc(&data, (size_t)65536 * 256 * 16 * 4 + 4096);
    (void)hipMemset(ctr, 0, 4096), (void)hipMemset(data, 0, (size_t)65536 * 256 * 16 * 4 + 4096);
    for (int work : {0, 1, 3})
        for (unsigned n : {20000u, 100000u}) {
            run<0>("no ticket", ctr, out, data, n, work);
            run<1>("ticket (returning atomic, one address)", ctr, out, data, n, work);
            run<2>("ticket + non-returning done counter", ctr, out, data, n, work);
            run<3>("ticket by every 4th workgroup", ctr, out, data, n, work);
        }
    return 0;
}
