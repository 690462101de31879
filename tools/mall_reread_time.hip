// Is a SECOND read of a stream cheap when it follows the first within the memory-side cache's reach?
// The question behind a decoder that counts a batch of the stream K batches AHEAD of the one it expands (the count's
// loads allocate in the 256 MiB Infinity Cache, the expansion's loads of the same lines K batches later hit there)
// instead of holding the batch on the chip between count and expansion.
//   mode 0: every workgroup reads its 64 KiB chunk i ONCE (nontemporal) and writes `ratio` x 64 KiB   (the one-pass decoder's bytes)
//   mode 1: reads chunk i + K (first touch, default policy) AND chunk i (second touch, K chunks later) (look-ahead count)
//   mode 2: reads chunk i of TWO different buffers, both nontemporal                                  (two HBM reads: the two launches' bytes)
//   mode 3: as mode 1 with the first touch nontemporal                                                 (control: the second touch should miss)
// hipcc --offload-arch=gfx950 -O3 -o mall_reread_time mall_reread_time.hip && ./mall_reread_time
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kThreads = 512, kChunk16 = 4096; // 64 KiB per workgroup

template <int MODE, int RATIO>
__global__ __launch_bounds__(kThreads) void probe(const u32x4 *in, const u32x4 *in2, u32x4 *out, unsigned chunks, unsigned K) {
    const unsigned i = blockIdx.x, t = threadIdx.x;
    u32x4 a[8], b[8];
    const u32x4 *p = in + (size_t)i * kChunk16;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = MODE == 1 || MODE == 3 ? p[t + kThreads * k] : __builtin_nontemporal_load(p + t + kThreads * k);
    if (MODE != 0) {
        const unsigned j = MODE == 2 ? i : (i + K < chunks ? i + K : i);
        const u32x4 *q = (MODE == 2 ? in2 : in) + (size_t)j * kChunk16;
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = MODE == 1 ? q[t + kThreads * k] : __builtin_nontemporal_load(q + t + kThreads * k);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] ^= b[k];
    }
    u32x4 *o = out + (size_t)i * kChunk16 * RATIO;
#pragma unroll
    for (int r = 0; r < RATIO; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) o[(size_t)r * kChunk16 + t + kThreads * k] = a[k] + (uint32_t)r;
}

template <int MODE, int RATIO>
static void run(const char *name, const u32x4 *in, const u32x4 *in2, u32x4 *out, unsigned chunks, unsigned K) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, RATIO>), dim3(chunks), dim3(kThreads), 0, 0, in, in2, out, chunks, K);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<MODE, RATIO>), dim3(chunks), dim3(kThreads), 0, 0, in, in2, out, chunks, K);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double once = (double)chunks * 65536.0 * (1 + RATIO);
    printf("ratio %d  %-58s K %5u  %.4f ms  %.0f GB/s of (stream once + output)\n", RATIO, name, K, ms, once / ms / 1e6);
    fflush(stdout);
}

int main() {
    const unsigned chunks = 8192; // 512 MiB of stream
    u32x4 *in, *in2, *out;
    hipMalloc(&in, (size_t)chunks * 65536), hipMalloc(&in2, (size_t)chunks * 65536), hipMalloc(&out, (size_t)chunks * 65536 * 2);
    hipMemset(in, 1, (size_t)chunks * 65536), hipMemset(in2, 2, (size_t)chunks * 65536);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 2>("read once (nt)", in, in2, out, chunks, 0);
        run<2, 2>("two buffers, both from HBM (nt)", in, in2, out, chunks, 0);
        for (unsigned K : {64u, 192u, 384u, 768u, 1536u, 4096u}) run<1, 2>("chunk i + K first touch, chunk i second touch", in, in2, out, chunks, K);
        run<3, 2>("control: first touch nt, second default", in, in2, out, chunks, 384);
        run<0, 1>("read once (nt)", in, in2, out, chunks, 0);
        run<2, 1>("two buffers, both from HBM (nt)", in, in2, out, chunks, 0);
        for (unsigned K : {64u, 192u, 384u, 768u, 1536u}) run<1, 1>("chunk i + K first touch, chunk i second touch", in, in2, out, chunks, K);
    }
    return 0;
}
