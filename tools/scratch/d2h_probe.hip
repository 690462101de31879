// How fast can a fresh, free()-able host buffer be filled from the device?  (host boundary of wah_decompress)
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/d2h_probe tools/scratch/d2h_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void *fresh(size_t bytes, bool huge) {
    const size_t h = size_t(2) << 20;
    void *p = std::aligned_alloc(h, (bytes + h - 1) & ~(h - 1));
    if (huge) madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}
#include <atomic>
#include <memory>
static void piped(void *d, size_t bytes, int n_threads, size_t chunk, bool huge, bool populate) {
    void *h = fresh(bytes, huge);
    double t0 = now();
    const size_t n_chunks = (bytes + chunk - 1) / chunk;
    std::unique_ptr<std::atomic<uint8_t>[]> ready(new std::atomic<uint8_t>[n_chunks]);
    for (size_t i = 0; i < n_chunks; ++i) ready[i] = 0;
    double t_fault_done = 0;
    std::atomic<int> left{n_threads};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&, t] {
            for (size_t i = t; i < n_chunks; i += n_threads) {
                char *p = (char *)h + i * chunk;
                size_t len = std::min(chunk, bytes - i * chunk);
                if (!populate || madvise(p, len, 23 /* MADV_POPULATE_WRITE */) != 0)
                    for (size_t o = 0; o < len; o += 4096) ((volatile char *)p)[o] = 0;
                ready[i].store(1, std::memory_order_release);
            }
            if (--left == 0) t_fault_done = now();
        });
    for (size_t i = 0; i < n_chunks; ++i) {
        while (!ready[i].load(std::memory_order_acquire)) std::this_thread::yield();
        hipMemcpy((char *)h + i * chunk, (char *)d + i * chunk, std::min(chunk, bytes - i * chunk), hipMemcpyDeviceToHost);
    }
    for (auto &x : th) x.join();
    double t1 = now();
    printf("piped threads=%d chunk=%zu MiB huge=%d populate=%d: faults done at %.1f ms, all %.1f ms (%.1f GB/s)\n", n_threads, chunk >> 20,
           huge, populate, 1e3 * (t_fault_done - t0), 1e3 * (t1 - t0), bytes / (t1 - t0) / 1e9);
    free(h);
}
int main() {
    const size_t bytes = size_t(1) << 30;
    void *d;
    hipMalloc(&d, bytes);
    hipMemset(d, 0x5a, bytes);
    hipDeviceSynchronize();
    for (int huge = 0; huge < 2; ++huge)
        for (int populate = 0; populate < 2; ++populate)
            for (int nt : {1, 4, 8}) piped(d, bytes, nt, size_t(16) << 20, huge, populate);
    piped(d, bytes, 4, size_t(64) << 20, true, true);
    piped(d, bytes, 4, size_t(4) << 20, true, true);
    piped(d, bytes, 4, size_t(64) << 20, false, false);
    piped(d, bytes, 8, size_t(4) << 20, false, false);
    for (int rep = 0; rep < 1; ++rep) {
        for (int huge = 0; huge < 2; ++huge) {
            void *h = fresh(bytes, huge);
            double t0 = now();
            hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
            double t1 = now();
            printf("plain hipMemcpy, huge=%d: %.1f ms (%.1f GB/s)\n", huge, 1e3 * (t1 - t0), bytes / (t1 - t0) / 1e9);
            t0 = now();
            hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
            t1 = now();
            printf("  again into the touched buffer: %.1f ms (%.1f GB/s)\n", 1e3 * (t1 - t0), bytes / (t1 - t0) / 1e9);
            free(h);
        }
        {
            void *h = fresh(bytes, true);
            double t0 = now();
            hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault);
            double t1 = now();
            hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
            double t2 = now();
            hipHostUnregister(h);
            double t3 = now();
            printf("register(%d) %.1f + copy %.1f + unregister %.1f = %.1f ms (%.1f GB/s)\n", (int)e, 1e3 * (t1 - t0), 1e3 * (t2 - t1),
                   1e3 * (t3 - t2), 1e3 * (t3 - t0), bytes / (t3 - t0) / 1e9);
            free(h);
        }
        for (int nthreads : {2, 4, 8}) {
            // pinned staging ring + copy threads
            const size_t chunk = size_t(8) << 20;
            const int slots = 4;
            void *stage[slots];
            for (auto &s : stage) hipHostMalloc(&s, chunk, hipHostMallocDefault);
            void *h = fresh(bytes, true);
            hipStream_t st;
            hipStreamCreate(&st);
            hipEvent_t ev[slots];
            for (auto &e : ev) hipEventCreate(&e);
            double t0 = now();
            const size_t nchunks = bytes / chunk;
            for (size_t c = 0; c < nchunks + slots; ++c) {
                if (c >= slots) {
                    const size_t k = c - slots;
                    hipEventSynchronize(ev[k % slots]);
                    std::vector<std::thread> th;
                    const size_t part = chunk / nthreads;
                    for (int t = 0; t < nthreads; ++t)
                        th.emplace_back([&, t] { memcpy((char *)h + k * chunk + t * part, (char *)stage[k % slots] + t * part, part); });
                    for (auto &x : th) x.join();
                }
                if (c < nchunks) {
                    hipMemcpyAsync(stage[c % slots], (char *)d + c * chunk, chunk, hipMemcpyDeviceToHost, st);
                    hipEventRecord(ev[c % slots], st);
                }
            }
            double t1 = now();
            printf("staged, %d copy threads: %.1f ms (%.1f GB/s)\n", nthreads, 1e3 * (t1 - t0), bytes / (t1 - t0) / 1e9);
            free(h);
            for (auto &s : stage) hipHostFree(s);
        }
    }
    return 0;
}
