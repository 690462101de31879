// wah_api.hip -- the C ABI of include/wah.h and the C++-linkage drop-ins of
// include/compress.h / include/decompress.h, on top of the kernels.
//
// Host orchestration replacing compress.cu:41-209 and decompress.cu:18-141:
// same phases and the same three timings, but the device phase is
// memset(control block) + kernel(s) with no allocation, no blocking 8-byte
// read-back in the middle and no library scan.
#include <cstdio>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <cstdlib>
#include <cstring>

// (-fvisibility=hidden: the reference's two entry points are exported like everything include/wah.h declares; their
//  headers stay the reference's text)
#pragma GCC visibility push(default)
#include "../../include/compress.h"
#include "../../include/decompress.h"
#pragma GCC visibility pop
#include "../../include/wah.h"
#include "wah_internal.hpp"

namespace {

thread_local char g_err[256] = "";

void set_err(const char *what, hipError_t e = hipSuccess) {
    if (e != hipSuccess)
        std::snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    else
        std::snprintf(g_err, sizeof g_err, "%s", what);
}

inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
inline size_t round256(size_t x) { return (x + 255) & ~(size_t)255; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Workspace layouts.  Everything a launch stamps with its epoch lives in areas whose PLACE depends on the workspace's size
// only, never on the size of the input: one workspace serves inputs of different sizes in turn (include/wah.h), and an
// area that moved with the input would let one call's data be read as another call's published entries.  Both layouts
// are [control block][first half][second half], the halves being (workspace_bytes - control block) / 2 each.
struct CompressLayout {
    uint64_t n_groups, n_segments, n_tiles;
    uint32_t wave_segs;
    size_t ctrl_off, desc_off, unseg_off, half, total; // total: what an input of this size needs
};

// first half: scan area of compress_pair_kernel / compress_tile_kernel (one block of kScanBlockWords per 64 x 256 tiles, +
// the block a full last superrow publishes into); second half: scan area of the unsegmented mode (blocks of
// kUnsegBlockWords), which the no-wait route borrows for its table of tile counts (offsets below 2^48: never a valid
// epoch stamp)
CompressLayout compress_layout(uint64_t n_words, size_t workspace_bytes = 0) {
    CompressLayout l;
    l.n_groups = wah_max_compressed_words(n_words);
    l.n_segments = ceil_div(l.n_groups, wah::kSegGroups);
    l.wave_segs = wah::compress_wave_segs(l.n_segments);
    l.n_tiles = ceil_div(l.n_segments, (uint64_t)wah::kCompressTileWaves * l.wave_segs);
    // sized for the shortest tiles: a workspace serves any smaller bitmap too
    const uint64_t blocks = ceil_div(l.n_segments, (uint64_t)wah::kCompressTileWaves) / wah::kScanBlockTiles + 1;
    static_assert(wah::kUnsegBlockWords >= wah::kScanBlockWords, "the halves are sized by the larger block");
    const size_t need_half = round256(blocks * wah::kUnsegBlockWords * sizeof(uint32_t));
    l.ctrl_off = 0;
    l.desc_off = wah::kCtlWords * sizeof(uint32_t);
    l.total = l.desc_off + 2 * need_half;
    const size_t w = workspace_bytes ? workspace_bytes : l.total;
    l.half = w > l.desc_off ? ((w - l.desc_off) / 2) & ~(size_t)255 : 0;
    l.unseg_off = l.desc_off + l.half;
    if (l.half < need_half) l.total = w + 1; // (does not fit: the callers compare workspace_bytes with total)
    return l;
}

struct DecodeLayout {
    uint64_t n_tiles;
    size_t ctrl_off, desc_off, base_off, flags_off, defer_off, bucket_off, half, total, scan_bytes;
};

// first half: scan area of the sums kernel / of the one-pass decoder (one block per 64 x 256 workgroup tiles, + the one a full
// last superrow publishes into; the one-pass decoder's workgroup tiles are the shorter ones: 8192 words); second half: tile
// bases, one flag byte per tile, the one-pass decoder's list of deferred tiles (rewritten by every call, no epochs)
DecodeLayout decode_layout(uint64_t c_words, size_t workspace_bytes = 0) {
    DecodeLayout l;
    l.n_tiles = ceil_div(c_words, (uint64_t)wah::kScanTileWords);
    const uint64_t wg_tiles = ceil_div(c_words, (uint64_t)wah::kDecodeTileWords);
    const uint64_t blocks = wg_tiles / wah::kSumScanBlockTiles + 1;
    const size_t scan_need = blocks * wah::kSumScanBlockWords * sizeof(uint32_t);
    const size_t base_bytes = round256((l.n_tiles + 4) * sizeof(uint64_t)); // (+ two words for wah_validate_device)
    const size_t flag_bytes = round256(l.n_tiles + 16);                       // one byte per tile: contains empty fills
    const size_t defer_bytes = round256((size_t)wah::decode_defer_capacity(l.n_tiles, c_words) * 2 * sizeof(uint64_t)); // the list of deferred / shared-out tiles
    const size_t bucket_bytes = round256((l.n_tiles + 1) * 64 * sizeof(uint32_t)); // ... and their sums of group counts per 64 words
    const size_t rest_need = base_bytes + flag_bytes + defer_bytes + bucket_bytes;
    const size_t need_half = round256(scan_need > rest_need ? scan_need : rest_need);
    l.ctrl_off = 0;
    l.desc_off = wah::kCtlWords * sizeof(uint32_t);
    l.total = l.desc_off + 2 * need_half;
    const size_t w = workspace_bytes ? workspace_bytes : l.total;
    l.half = w > l.desc_off ? ((w - l.desc_off) / 2) & ~(size_t)255 : 0;
    l.scan_bytes = l.half; // (the wrap-around clear covers the whole first half, whatever was launched into it)
    l.base_off = l.desc_off + l.half;
    l.flags_off = l.base_off + base_bytes;
    l.defer_off = l.flags_off + flag_bytes; // the list (its counters: kCtlDefer in the control block)
    l.bucket_off = l.defer_off + defer_bytes;
    if (l.half < need_half) l.total = w + 1;
    return l;
}

int status_from_bits(uint32_t err);

// status word of a launch, and optionally `n_vals` 64-bit results of it, in ONE round trip to the device
int read_status(void *d_workspace, void *stream, const uint64_t *d_vals = nullptr, uint64_t *vals = nullptr, int n_vals = 0) {
    uint32_t err = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(&err, static_cast<uint32_t *>(d_workspace) + wah::kCtlError, sizeof err,
                                  hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && n_vals) e = hipMemcpyAsync(vals, d_vals, n_vals * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        set_err("status read-back", e);
        return WAH_ERR_HIP;
    }
    return status_from_bits(err);
}

int status_from_bits(uint32_t err) {
    if (err & wah::kErrWorkspace) {
        set_err("compress workspace was not initialised (wah_workspace_init_device)");
        return WAH_ERR_WORKSPACE;
    }
    if (err & wah::kErrTimeout) {
        set_err("in-kernel bounded wait expired");
        return WAH_ERR_TIMEOUT;
    }
    if (err & wah::kErrStream) {
        set_err("malformed compressed stream");
        return WAH_ERR_STREAM;
    }
    if (err & wah::kErrCapacity) {
        set_err("output capacity too small");
        return WAH_ERR_CAPACITY;
    }
    return WAH_OK;
}

// Host-pointer entry points: the results of a launch (kCtlResult) lie beside its error word, so one 32-byte copy into
// page-locked memory + one stream synchronisation is the whole read-back (two pageable copies cost 2 x 15 us more).
int read_status_packed(void *d_workspace, uint32_t *pinned, uint64_t *vals, int n_vals) {
    hipError_t e = hipMemcpyAsync(pinned, static_cast<uint32_t *>(d_workspace) + wah::kCtlError, 8 * sizeof(uint32_t),
                                  hipMemcpyDeviceToHost, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        set_err("status read-back", e);
        return WAH_ERR_HIP;
    }
    for (int i = 0; i < n_vals; ++i) std::memcpy(&vals[i], pinned + (wah::kCtlResult - wah::kCtlError) + 2 * i, sizeof(uint64_t));
    return status_from_bits(pinned[0]);
}

// ... or without any copy: the last tile of the launch wrote {1 | error bits << 32, results...} into page-locked host
// memory; wait for the stream and read them.  A launch that ended early (unusable workspace) leaves the mark unset.
int wait_host_result(volatile uint64_t *result, void *d_workspace, uint32_t *pinned, uint64_t *vals, int n_vals) {
    const hipError_t e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        set_err("waiting for the device", e);
        return WAH_ERR_HIP;
    }
    if (!(result[0] & 1ull)) return read_status_packed(d_workspace, pinned, vals, n_vals);
    for (int i = 0; i < n_vals; ++i) vals[i] = result[1 + i];
    return status_from_bits((uint32_t)(result[0] >> 32));
}

// Device buffers of the host-pointer entry points.  The reference allocates and frees them inside every call
// (compress.cu:57-114,177-202; decompress.cu:34-54,124-131), which puts hipMalloc/hipFree of up to two bitmap-sized
// buffers -- milliseconds, and cold address translations for the kernels that follow -- into the timings it
// reports.  Here the buffers are kept between calls: grow-only, one set per device, the call holds the set's
// mutex.  wah_host_cache_release() frees them; WAH_HOST_CACHE=0 in the environment restores allocate-and-free.
struct HostCache {
    static constexpr int kSlots = 6;
    std::mutex m;
    void *buf[kSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[kSlots] = {0, 0, 0, 0, 0, 0};
    int device = -1; // the device the kept buffers live on
    uint32_t *pinned = nullptr; // 64 bytes of page-locked host memory: where status + sizes of a launch land (copied, or
                                // written by the kernel itself: CompressArgs::host_result)
    void release_locked() {
        for (int i = 0; i < kSlots; ++i) {
            if (buf[i]) (void)hipFree(buf[i]);
            buf[i] = nullptr;
            cap[i] = 0;
        }
    }
};
// one set per device: host threads that drive different GPUs (one host thread per GPU) neither
// share a mutex nor evict each other's buffers
constexpr int kMaxDevices = 64;
HostCache &host_cache(int device) {
    static HostCache *c = new HostCache[kMaxDevices]; // never destroyed: the HIP runtime may be gone before static destructors run
    return c[device >= 0 && device < kMaxDevices ? device : 0];
}
int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    return dev;
}
bool host_cache_enabled() {
    static const bool on = [] {
        const char *e = std::getenv("WAH_HOST_CACHE");
        return !(e && e[0] == '0');
    }();
    return on;
}

// RAII bundle for the host-pointer paths: owns the timing events and the call's claim on the buffer set
// (or, without the cache, frees whatever was allocated, like the error exits of compress.cu:89-114).
struct HostCall {
    HostCache &cache = host_cache(current_device());
    std::unique_lock<std::mutex> lock{cache.m};
    const bool keep = host_cache_enabled();
    hipEvent_t ev[2] = {nullptr, nullptr};
    ~HostCall() {
        release();
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
    }
    void release() {
        if (!keep) cache.release_locked();
    }
    // slot: which buffer of the set; a kept buffer that is large enough is handed out again as it is
    // (quiet: a failure is the caller's to handle -- no message, no error text)
    bool alloc(int slot, void **p, size_t bytes, const char *what, bool *fresh = nullptr, bool quiet = false) {
        if (bytes == 0) bytes = 16;
        if (fresh) *fresh = false;
        if (cache.buf[slot] && cache.cap[slot] >= bytes) {
            *p = cache.buf[slot];
            return true;
        }
        if (fresh) *fresh = true;
        if (cache.buf[slot]) (void)hipFree(cache.buf[slot]);
        cache.buf[slot] = nullptr;
        cache.cap[slot] = 0;
        hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) {
            if (quiet) {
                (void)hipGetLastError();
                return false;
            }
            set_err(what, e);
            std::fprintf(stderr, "wah: could not allocate %s (%zu bytes): %s\n", what, bytes, hipGetErrorString(e));
            return false;
        }
        cache.buf[slot] = *p;
        cache.cap[slot] = bytes;
        return true;
    }
    bool init() {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        cache.device = dev;
        if (!cache.pinned && hipHostMalloc(reinterpret_cast<void **>(&cache.pinned), 128, hipHostMallocDefault) != hipSuccess) {
            cache.pinned = nullptr;
            return false;
        }
        return hipEventCreate(&ev[0]) == hipSuccess && hipEventCreate(&ev[1]) == hipSuccess;
    }
    void start() { (void)hipEventRecord(ev[0], nullptr); }
    // end of a phase that ends with device work: the event goes into the stream right behind the launch, so that it
    // carries the time at which the device finished, not the time at which the host noticed
    void mark() { (void)hipEventRecord(ev[1], nullptr); }
    float since_mark() {
        float ms = 0.f;
        (void)hipEventSynchronize(ev[1]);
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        return ms;
    }
    float stop() {
        float ms = 0.f;
        (void)hipEventRecord(ev[1], nullptr);
        (void)hipEventSynchronize(ev[1]);
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        return ms;
    }
};

// Result buffers of the host-pointer paths: free()-compatible like the reference's malloc (compress.cu:177,
// decompress.cu:127), but large ones are 2 MiB aligned and marked for transparent huge pages -- the copy back from the
// device is the first touch of this memory, and with 4 KiB pages its page faults, not PCIe, set the pace.
// *huge: the buffer is backed by huge pages on first touch (what copy_to_fresh_host() needs to know).
bool thp_available() {
    static const bool on = [] {
        char line[128] = "";
        if (FILE *f = std::fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r")) {
            if (!std::fgets(line, sizeof line, f)) line[0] = 0;
            std::fclose(f);
        }
        return line[0] != 0 && std::strstr(line, "[never]") == nullptr;
    }();
    return on;
}
void *host_result_alloc(size_t bytes, bool *huge) {
    constexpr size_t kHuge = size_t(2) << 20;
    *huge = false;
    if (bytes >= 2 * kHuge && thp_available()) {
        const size_t rounded = (bytes + kHuge - 1) & ~(kHuge - 1);
        void *p = std::aligned_alloc(kHuge, rounded);
        if (p) {
            *huge = madvise(p, rounded, MADV_HUGEPAGE) == 0;
            return p;
        }
    }
    return std::malloc(bytes ? bytes : sizeof(uint32_t));
}

// WAH_FAULT_INJECT=timeout (include/wah.h; read per call): the host-pointer entry points treat their first launch as if one
// of its bounded waits had expired, which sends them down the no-wait route -- the only way to see that route taken by
// itself on a healthy GPU.
bool test_timeout_hook() {
    const char *e = std::getenv("WAH_FAULT_INJECT");
    return e && std::strcmp(e, "timeout") == 0;
}

thread_local int g_last_route = wah::kRouteNone; // which decoder the calling thread's last decode call launched
thread_local int g_last_bitop_route = 0;         // ... and which route its last indexed bit operation took (WAH_BITOP_ROUTE_*)

bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    set_err(what, e);
    std::fprintf(stderr, "wah: %s failed: %s\n", what, hipGetErrorString(e));
    return false;
}

// Device -> fresh host buffer.  Measured on the box (tools/scratch/d2h_probe.hip, 1 GiB): hipMemcpy into memory that
// has been touched runs at 56 GB/s, into untouched memory at 17-21 GB/s, because the copy takes the page faults one at
// a time.  With huge pages four threads fault 1 GiB in within 11 ms, so they run ahead of the copy, chunk by chunk, and
// the whole transfer reaches 47 GB/s.  (With 4 KiB pages concurrent faults contend and the plain copy is the faster
// one; MADV_POPULATE_WRITE does not scale over threads either.)
bool copy_to_fresh_host(void *host, const void *dev, size_t bytes, bool huge, const char *what) {
    constexpr size_t kChunk = size_t(16) << 20;
    const size_t n_chunks = (bytes + kChunk - 1) / kChunk;
    const unsigned hw = std::thread::hardware_concurrency();
    const unsigned n_threads = hw >= 8 ? 4 : hw >= 4 ? 2 : 1;
    if (!huge || n_chunks < 2) return hip_ok(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost), what);

    std::unique_ptr<std::atomic<uint8_t>[]> ready(new std::atomic<uint8_t>[n_chunks]);
    for (size_t i = 0; i < n_chunks; ++i) ready[i].store(0, std::memory_order_relaxed);
    std::atomic<bool> stop{false};
    auto fault_in = [&](unsigned t) {
        for (size_t i = t; i < n_chunks && !stop.load(std::memory_order_relaxed); i += n_threads) {
            volatile char *p = static_cast<char *>(host) + i * kChunk;
            const size_t len = std::min(kChunk, bytes - i * kChunk);
            for (size_t o = 0; o < len; o += 4096) p[o] = 0;
            ready[i].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> helpers;
    helpers.reserve(n_threads);
    try {
        for (unsigned t = 0; t < n_threads; ++t) helpers.emplace_back(fault_in, t);
    } catch (...) { // no threads to be had: whoever did start keeps going, the rest of the chunks are touched here
    }
    const unsigned started = (unsigned)helpers.size();
    bool ok = true;
    for (size_t i = 0; i < n_chunks && ok; ++i) {
        if (i % n_threads >= started) { // the helper that owns this chunk does not exist
            volatile char *p = static_cast<char *>(host) + i * kChunk;
            const size_t len = std::min(kChunk, bytes - i * kChunk);
            for (size_t o = 0; o < len; o += 4096) p[o] = 0;
            ready[i].store(1, std::memory_order_release);
        }
        while (!ready[i].load(std::memory_order_acquire)) std::this_thread::yield();
        const size_t len = std::min(kChunk, bytes - i * kChunk);
        ok = hip_ok(hipMemcpy(static_cast<char *>(host) + i * kChunk, static_cast<const char *>(dev) + i * kChunk, len,
                              hipMemcpyDeviceToHost), what);
    }
    stop.store(true, std::memory_order_relaxed);
    for (std::thread &h : helpers) h.join();
    return ok;
}

} // namespace

extern "C" {

const char *wah_last_error(void) { return g_err; }
const char *wah_version(void) { return "wah-mi355x 0.1 gfx950"; }
void wah_free(void *p) { std::free(p); }
void wah_host_cache_release(void) {
    for (int d = 0; d < kMaxDevices; ++d) {
        HostCache &c = host_cache(d);
        std::lock_guard<std::mutex> g(c.m);
        if (c.device < 0) continue; // never used
        int prev = 0;
        const bool switched = hipGetDevice(&prev) == hipSuccess && prev != c.device && hipSetDevice(c.device) == hipSuccess;
        c.release_locked();
        if (switched) (void)hipSetDevice(prev);
    }
}

uint64_t wah_max_compressed_words(uint64_t n_words) { return (32u * n_words + 30u) / 31u; }
uint64_t wah_decoded_words(uint64_t n_groups) { return (31u * n_groups + 31u) / 32u; }

size_t wah_compress_workspace_bytes(uint64_t n_words) { return compress_layout(n_words).total; }
size_t wah_decompress_workspace_bytes(uint64_t c_words, uint64_t out_capacity_words) {
    (void)out_capacity_words; // the workspace depends on the stream length only
    return decode_layout(c_words).total;
}

// clear_first: the caller's workspace is scratch of unknown content (the bitop paths): zero all of it in front of the
// launch, which makes it a fresh workspace every time -- and keep whatever error an upstream pass leaves in it.
static int compress_device_impl(const uint32_t *d_in, const uint32_t *d_in2, int op, const wah::PairCheck *check, uint64_t n_words,
                                uint32_t *d_out, uint64_t out_capacity_words, uint64_t *d_out_words, uint64_t *d_segment_offsets,
                                void *d_workspace, size_t workspace_bytes, void *stream, bool clear_first,
                                uint64_t *host_result = nullptr, const wah::BitopOperands *indexed = nullptr, bool unsegmented = false,
                                bool no_wait = false) {
    g_err[0] = 0;
    // WAH_FORCE_FALLBACK=1: every plain compress launch takes the no-wait route (tests; a GPU shared in ways that starve
    // the scan route's waits).  Read per call: it is a switch for a running process too.
    if (!no_wait) {
        const char *f = std::getenv("WAH_FORCE_FALLBACK");
        no_wait = f && f[0] == '1';
    }
    if (!d_out_words || !d_workspace || (n_words && ((!d_in && !indexed) || !d_out))) {
        set_err("null pointer");
        return WAH_ERR_ARG;
    }
    if (n_words >= (1ull << 40) || (reinterpret_cast<uintptr_t>(d_in) & 3u) || (reinterpret_cast<uintptr_t>(d_workspace) & 255u)) {
        set_err("size out of range or misaligned pointer");
        return WAH_ERR_ARG;
    }
    const CompressLayout l = compress_layout(n_words, workspace_bytes);
    if (workspace_bytes < l.total) {
        set_err("workspace too small");
        return WAH_ERR_WORKSPACE;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(d_workspace);
    hipError_t e = hipSuccess;
    if (clear_first) {
        e = wah::launch_clear(ws, l.desc_off + 2 * l.half, s);
        if (e != hipSuccess) {
            set_err("clearing the workspace", e);
            return WAH_ERR_HIP;
        }
    }
    if (n_words == 0) {
        e = wah::launch_clear(d_out_words, sizeof(uint64_t), s);
        if (e == hipSuccess && d_segment_offsets) e = wah::launch_clear(d_segment_offsets, sizeof(uint64_t), s);
        if (e == hipSuccess && !clear_first) e = wah::launch_clear(ws + wah::kCtlError * sizeof(uint32_t), sizeof(uint32_t), s);
        if (e != hipSuccess) {
            set_err("clearing the outputs", e);
            return WAH_ERR_HIP;
        }
        return WAH_OK;
    }
    wah::CompressArgs a{};
    a.in = d_in;
    a.n_words = n_words;
    a.n_segments = (uint32_t)l.n_segments;
    a.n_tiles = (uint32_t)l.n_tiles;
    a.wave_segs = l.wave_segs;
    a.unseg_desc = nullptr;
    a.pair_layout = 0;
    if (!d_in2 && !indexed && !unsegmented && !no_wait) { // the plain compress: pair-layout kernel, its own tile shape
        const wah::TileShape shape = wah::compress_tile_shape(l.n_segments);
        if (shape.body_pairs) {
            a.pair_layout = 1;
            a.wave_segs = 2 * shape.body_pairs;
            a.tail_pairs = shape.tail_pairs;
            a.big_tiles = shape.big_tiles;
            a.n_tiles = shape.n_tiles;
        }
    }
    if (unsegmented) { // fills cross the segment cut (compress_unseg_pair_kernel): its own scan area, the tile shapes of the plain compress
        a.unseg_desc = reinterpret_cast<uint32_t *>(ws + l.unseg_off);
        const wah::TileShape shape = wah::compress_tile_shape(l.n_segments);
        const uint32_t body = shape.body_pairs ? shape.body_pairs : 3u; // (WAH_WAVE_PAIRS=0 switches only the plain compress's kernel off)
        a.pair_layout = 1;
        a.wave_segs = 2 * body;
        a.tail_pairs = shape.body_pairs ? shape.tail_pairs : 3u;
        a.big_tiles = shape.body_pairs ? shape.big_tiles : (uint32_t)ceil_div((l.n_segments + 1) / 2, (uint64_t)wah::kCompressTileWaves * 3u);
        a.n_tiles = shape.body_pairs ? shape.n_tiles : a.big_tiles;
    }
    if (indexed) { // groups come from two indexed streams (bitop_tile_kernel): its own tile shape
        a.wave_segs = wah::kIndexedSegsPerWave;
        a.n_tiles = (uint32_t)ceil_div(l.n_segments, (uint64_t)wah::kCompressTileWaves * wah::kIndexedSegsPerWave);
    }
    if (no_wait) { // count / scan / place: nobody waits for anybody.  Its own tile shape (two segments per wave; the plain
                   // compress: two pairs); the table of tile counts lies in the workspace's second half
        a.wave_segs = (d_in2 || indexed) ? 2u : wah::compress_nowait_wave_segs(); // (plain and unsegmented: two PAIRS per wave)
        a.n_tiles = (uint32_t)ceil_div(l.n_segments, (uint64_t)wah::kCompressTileWaves * a.wave_segs);
        a.tile_counts = reinterpret_cast<uint64_t *>(ws + l.unseg_off);
        a.pair_layout = 0;
    }
    a.fast_segments = aligned16(d_in) ? 1u : 0u;
    a.full_segments = (uint32_t)(n_words / wah::kSegWords);
    a.tail_bytes = (uint32_t)(n_words % wah::kSegWords) * 4u;
    a.last_segment_groups = (uint32_t)(l.n_groups - (l.n_segments - 1) * wah::kSegGroups);
    a.out = d_out;
    a.out_capacity = out_capacity_words;
    a.out_words = d_out_words;
    a.seg_offsets = d_segment_offsets;
    a.ctrl = reinterpret_cast<uint32_t *>(ws + l.ctrl_off);
    a.gen_desc = reinterpret_cast<uint32_t *>(ws + l.desc_off);
    a.scan_words = 2 * l.half / sizeof(uint32_t); // both halves (the wrap-around clear covers them, whatever was launched into them)
    a.keep_error = clear_first ? 1 : 0;
    a.host_result = host_result;
#ifdef WAH_DIAG
    {
        static const uint32_t tune = [] { // diagnostic build only: 77 / 78 = per-tile time line (tools/tile_timeline.py)
            const char *e = std::getenv("WAH_TUNE"); // (this block exists in the diagnostic build only)
            return e ? (uint32_t)std::strtoul(e, nullptr, 0) : 0u;
        }();
        a.tune = tune;
    }
#endif
    a.in2 = d_in2;
    a.op = (uint32_t)op;
    if (d_in2) {
        if (!a.fast_segments || !aligned16(d_in2)) {
            set_err("pair mode needs 16-byte aligned bitmaps");
            return WAH_ERR_ARG;
        }
        if (check) e = wah::launch_bitop_check(check->info_a, check->info_b, check->ctrl_a, check->ctrl_b, check->groups, a.ctrl, s);
    }
    if (e == hipSuccess) e = indexed ? wah::launch_bitop_tiles(a, *indexed, s) : no_wait ? wah::launch_compress_nowait(a, s) : wah::launch_compress(a, s);
    if (e != hipSuccess) {
        set_err("compress kernel launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

int wah_workspace_init_device(void *d_workspace, size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!d_workspace || (reinterpret_cast<uintptr_t>(d_workspace) & 255u)) {
        set_err("null or misaligned workspace");
        return WAH_ERR_ARG;
    }
    const hipError_t e = wah::launch_clear(d_workspace, workspace_bytes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_err("clearing the workspace", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

int wah_compress_device_indexed(const uint32_t *d_in, uint64_t n_words, uint32_t *d_out, uint64_t out_capacity_words,
                                uint64_t *d_out_words, uint64_t *d_segment_offsets, void *d_workspace,
                                size_t workspace_bytes, void *stream) {
    return compress_device_impl(d_in, nullptr, 0, nullptr, n_words, d_out, out_capacity_words, d_out_words, d_segment_offsets,
                                d_workspace, workspace_bytes, stream, false);
}

int wah_compress_device_ex(const uint32_t *d_in, uint64_t n_words, uint32_t *d_out, uint64_t out_capacity_words,
                           uint64_t *d_out_words, unsigned flags, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (flags & ~(unsigned)(WAH_UNSEGMENTED | WAH_NO_WAIT)) {
        g_err[0] = 0;
        set_err("unknown flag");
        return WAH_ERR_ARG;
    }
    return compress_device_impl(d_in, nullptr, 0, nullptr, n_words, d_out, out_capacity_words, d_out_words, nullptr, d_workspace,
                                workspace_bytes, stream, false, nullptr, nullptr, (flags & WAH_UNSEGMENTED) != 0, (flags & WAH_NO_WAIT) != 0);
}

int wah_compress_device(const uint32_t *d_in, uint64_t n_words, uint32_t *d_out, uint64_t out_capacity_words,
                        uint64_t *d_out_words, void *d_workspace, size_t workspace_bytes, void *stream) {
    return wah_compress_device_indexed(d_in, n_words, d_out, out_capacity_words, d_out_words, nullptr, d_workspace,
                                       workspace_bytes, stream);
}

int wah_compress_status(void *d_workspace, void *stream) { return read_status(d_workspace, stream); }

// One host thread per shard: its device made current, a stream of its own, one launch over the shard's column matrix, the
// status read back.  Nothing crosses between the threads but the join (columns are unrelated bitmaps).
int wah_compress_columns_multi_device(int n_shards, const wah_column_shard *shards, uint64_t n_words_per_column, int *status) {
    g_err[0] = 0;
    if (n_shards < 0 || (n_shards > 0 && !shards)) {
        set_err("null shard list");
        return WAH_ERR_ARG;
    }
    if (n_words_per_column % WAH_SEGMENT_WORDS != 0) {
        set_err("columns of a shard must be whole 992-word segments (a fill must not cross into the next column)");
        return WAH_ERR_ARG;
    }
    std::vector<int> rc((size_t)n_shards, WAH_OK);
    std::vector<std::string> msg((size_t)n_shards);
    auto work = [&](int i) {
        const wah_column_shard &sh = shards[i];
        hipStream_t s = nullptr;
        if (hipSetDevice(sh.device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            rc[i] = WAH_ERR_HIP;
            msg[i] = "could not make the shard's device current / create its stream";
            return;
        }
        rc[i] = wah_compress_device_indexed(sh.d_in, sh.n_columns * n_words_per_column, sh.d_out, sh.out_capacity_words, sh.d_out_words,
                                            sh.d_segment_offsets, sh.d_workspace, sh.workspace_bytes, s);
        if (rc[i] == WAH_OK) rc[i] = wah_compress_status(sh.d_workspace, s); // (synchronises the shard's stream)
        if (rc[i] != WAH_OK) msg[i] = g_err; // (this thread's own error text)
        (void)hipStreamDestroy(s);
    };
    std::vector<std::thread> threads;
    threads.reserve((size_t)n_shards);
    for (int i = 0; i < n_shards; ++i) {
        try {
            threads.emplace_back(work, i);
        } catch (...) { // no thread to be had: this shard on the caller's thread
            int prev = 0;
            const bool have_prev = hipGetDevice(&prev) == hipSuccess;
            work(i);
            if (have_prev) (void)hipSetDevice(prev);
        }
    }
    for (std::thread &t : threads) t.join();
    int first = WAH_OK;
    for (int i = 0; i < n_shards; ++i) {
        if (status) status[i] = rc[i];
        if (first == WAH_OK && rc[i] != WAH_OK) {
            first = rc[i];
            set_err(msg[i].c_str());
        }
    }
    return first;
}
int wah_decompress_status(void *d_workspace, void *stream) { return read_status(d_workspace, stream); }

// clear_first: the workspace is scratch of unknown content (the bitop paths): zero its control block and scan area in
// front of the launch, which makes it a fresh workspace every time.
static int decode_common(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out, uint64_t out_capacity_words,
                         uint64_t *d_out_info, void *d_workspace, size_t workspace_bytes, void *stream, bool do_scan,
                         bool do_expand, bool clear_first = false, uint64_t *host_result = nullptr, bool no_wait = false,
                         int route = 0 /* 0: by the rule below, 1: one pass, 2: two launches */) {
    g_err[0] = 0;
    g_last_route = wah::kRouteNone;
    // WAH_FORCE_FALLBACK=1: every sums pass takes the no-wait route (tests; a GPU shared in ways that starve the scan
    // route's waits).  Read per call: it is a switch for a running process too.
    if (!no_wait && do_scan) {
        const char *f = std::getenv("WAH_FORCE_FALLBACK");
        no_wait = f && f[0] == '1';
    }
    if (!d_out_info || !d_workspace || (c_words && !d_comp) || (do_expand && out_capacity_words && !d_out)) {
        set_err("null pointer");
        return WAH_ERR_ARG;
    }
    if (c_words >= (1ull << 40) || (reinterpret_cast<uintptr_t>(d_comp) & 3u) || (reinterpret_cast<uintptr_t>(d_workspace) & 255u)) {
        set_err("size out of range or misaligned pointer");
        return WAH_ERR_ARG;
    }
    const DecodeLayout l = decode_layout(c_words, workspace_bytes);
    if (workspace_bytes < l.total) {
        set_err("workspace too small");
        return WAH_ERR_WORKSPACE;
    }
    if (l.n_tiles >= (1ull << 31)) {
        set_err("stream too long");
        return WAH_ERR_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(d_workspace);
    hipError_t e = hipSuccess;
    // ONE pass over the stream (decode_tile_kernel + the launch over its list) is CORRECT for every 16-byte aligned stream: the
    // kernel decides tile by tile, from the tile's own words, whether it expands the tile itself (up to about 7 groups per word)
    // or puts it on the list that the launch behind it shares out over work items.  It is also the FASTER route for streams of
    // up to 7 groups per word (by 13-19 % against the two launches, which read the stream twice), as fast within 1.5 % from about
    // 32 groups per word on (every tile on the list: the list's launch is the expand kernel) and 3-6 % faster from about 200; in
    // between -- 8 to 30 groups per word -- the two launches win by about 20 % (tools/decode_density_time.py, 992 MiB, one bit in
    // 2^9: 0.338 ms against 0.282, 2^10: 0.280 / 0.243 -- a stream of 60-130 MB is 2000-4000 workgroups of the tile kernel, whose
    // tickets and list entries come out of one address each at 86 per microsecond: 98 us where the sums kernel takes 41).  The library cannot look at the stream without a pass over it, so the default goes by what the
    // CAPACITY allows the stream to be: at most 7 words of output per word of stream, or more than 40 -- one pass; between --
    // the two launches.  A caller who knows better says so (WAH_ONE_PASS / WAH_TWO_LAUNCHES); decompress(), which has the stream
    // in host memory, samples it and does.  (WAH_DECODE_TWO_PASS=1, experiment builds: always the two launches.)
    static const bool two_pass_only = [] {
        const char *f = wah::experiment_env("WAH_DECODE_TWO_PASS");
        return f && f[0] == '1';
    }();
    // (7 words of output per word of stream: the tile kernel expands up to 7.5 groups = 7.27 words per word itself)
    const bool prefer_one_pass = route == 1 || (route == 0 && (out_capacity_words <= 7 * c_words || out_capacity_words > 40 * c_words));
    const bool one_pass = do_scan && do_expand && !no_wait && prefer_one_pass && !two_pass_only && c_words != 0 && aligned16(d_comp);
    if (do_expand || do_scan) g_last_route = one_pass ? wah::kRouteOnePass : no_wait ? wah::kRouteNoWait : wah::kRouteTwoLaunches;
    if (one_pass) {
        if (clear_first) e = wah::launch_clear(ws, l.base_off, s); // (control block -- with the deferred tiles' counters -- and scan area)
        if (e != hipSuccess) {
            set_err("clearing the workspace", e);
            return WAH_ERR_HIP;
        }
        wah::ScanArgs a{};
        a.comp = d_comp;
        a.c_words = c_words;
        a.n_tiles = l.n_tiles;
        a.info = d_out_info;
        a.tile_base = reinterpret_cast<uint64_t *>(ws + l.base_off);
        a.ctrl = reinterpret_cast<uint32_t *>(ws + l.ctrl_off);
        a.gen_desc = reinterpret_cast<uint32_t *>(ws + l.desc_off);
        a.scan_words = l.scan_bytes / sizeof(uint32_t);
        a.tile_flags = reinterpret_cast<uint8_t *>(ws + l.flags_off);
        a.aligned16 = 1;
        a.host_result = host_result;
        a.no_wait = 0;
        wah::ExpandArgs x{};
        x.comp = d_comp;
        x.c_words = c_words;
        x.out = d_out;
        x.out_capacity = out_capacity_words;
        x.info = d_out_info;
        x.tile_base = a.tile_base;
        x.tile_flags = a.tile_flags;
        x.ctrl = a.ctrl;
        x.aligned16 = 1;
        x.parts = 1;
        x.tile_buckets = reinterpret_cast<const uint32_t *>(ws + l.bucket_off);
        e = wah::launch_decode_tiles(a, x, reinterpret_cast<uint64_t *>(ws + l.defer_off), s);
        if (e != hipSuccess) {
            set_err("decode tile kernel launch", e);
            return WAH_ERR_HIP;
        }
        return WAH_OK;
    }
    if (do_scan) {
        if (clear_first) e = wah::launch_clear(ws, l.base_off, s);
        if (e == hipSuccess && c_words == 0) {
            e = wah::launch_clear(d_out_info, 2 * sizeof(uint64_t), s);
            if (e == hipSuccess && !clear_first) e = wah::launch_clear(ws + wah::kCtlError * sizeof(uint32_t), sizeof(uint32_t), s);
        }
        if (e != hipSuccess) {
            set_err("clearing the workspace", e);
            return WAH_ERR_HIP;
        }
        if (c_words) {
            wah::ScanArgs a{};
            a.comp = d_comp;
            a.c_words = c_words;
            a.n_tiles = l.n_tiles;
            a.info = d_out_info;
            a.tile_base = reinterpret_cast<uint64_t *>(ws + l.base_off);
            a.ctrl = reinterpret_cast<uint32_t *>(ws + l.ctrl_off);
            a.gen_desc = reinterpret_cast<uint32_t *>(ws + l.desc_off);
            a.scan_words = l.scan_bytes / sizeof(uint32_t);
            a.tile_flags = reinterpret_cast<uint8_t *>(ws + l.flags_off);
            a.aligned16 = aligned16(d_comp) ? 1 : 0;
            a.host_result = host_result;
            a.no_wait = no_wait ? 1 : 0;
            a.defer_list = reinterpret_cast<uint64_t *>(ws + l.defer_off);
            a.defer_capacity = wah::decode_defer_capacity(l.n_tiles, c_words);
            e = wah::launch_decode_sums(a, s);
            if (e != hipSuccess) {
                set_err("decode sums kernel launch", e);
                return WAH_ERR_HIP;
            }
        }
    }
    if (c_words == 0) return WAH_OK;
    if (do_expand) {
        wah::ExpandArgs x{};
        x.comp = d_comp;
        x.c_words = c_words;
        x.out = d_out;
        x.out_capacity = out_capacity_words;
        x.info = d_out_info;
        x.tile_base = reinterpret_cast<const uint64_t *>(ws + l.base_off);
        x.tile_flags = reinterpret_cast<const uint8_t *>(ws + l.flags_off);
        x.ctrl = reinterpret_cast<uint32_t *>(ws + l.ctrl_off);
        x.aligned16 = aligned16(d_comp) ? 1 : 0;
        x.parts = 1; // the launcher decides
        x.defer_list = reinterpret_cast<const uint64_t *>(ws + l.defer_off);
        x.defer_count = reinterpret_cast<const uint32_t *>(ws + l.ctrl_off) + wah::kCtlDefer;
        x.defer_capacity = wah::decode_defer_capacity(l.n_tiles, c_words);
        e = wah::launch_decode_expand(x, l.n_tiles, s);
        if (e != hipSuccess) {
            set_err("decode expand kernel launch", e);
            return WAH_ERR_HIP;
        }
    }
    return WAH_OK;
}

int wah_decompress_device(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out, uint64_t out_capacity_words,
                          uint64_t *d_out_info, void *d_workspace, size_t workspace_bytes, void *stream) {
    return decode_common(d_comp, c_words, d_out, out_capacity_words, d_out_info, d_workspace, workspace_bytes, stream,
                         true, true);
}

int wah_decompress_device_ex(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out, uint64_t out_capacity_words,
                             uint64_t *d_out_info, unsigned flags, void *d_workspace, size_t workspace_bytes, void *stream) {
    if ((flags & ~(unsigned)(WAH_NO_WAIT | WAH_TWO_LAUNCHES | WAH_ONE_PASS)) || ((flags & WAH_TWO_LAUNCHES) && (flags & WAH_ONE_PASS))) {
        g_err[0] = 0;
        set_err("unknown flag, or both WAH_ONE_PASS and WAH_TWO_LAUNCHES");
        return WAH_ERR_ARG;
    }
    return decode_common(d_comp, c_words, d_out, out_capacity_words, d_out_info, d_workspace, workspace_bytes, stream,
                         true, true, false, nullptr, (flags & WAH_NO_WAIT) != 0, (flags & WAH_ONE_PASS) ? 1 : (flags & WAH_TWO_LAUNCHES) ? 2 : 0);
}

int wah_last_decode_route(void) { return g_last_route; }
int wah_last_bitop_route(void) { return g_last_bitop_route; }

int wah_decompress_scan_device(const uint32_t *d_comp, uint64_t c_words, uint64_t *d_out_info, void *d_workspace,
                               size_t workspace_bytes, void *stream) {
    return decode_common(d_comp, c_words, nullptr, 0, d_out_info, d_workspace, workspace_bytes, stream, true, false);
}

int wah_decompress_expand_device(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out,
                                 uint64_t out_capacity_words, uint64_t *d_out_info, void *d_workspace,
                                 size_t workspace_bytes, void *stream) {
    return decode_common(d_comp, c_words, d_out, out_capacity_words, d_out_info, d_workspace, workspace_bytes, stream,
                         false, true);
}

int wah_validate_device(const uint32_t *d_comp, uint64_t c_words, uint64_t *d_report, void *d_workspace,
                        size_t workspace_bytes, void *stream) {
    if (!d_report) {
        g_err[0] = 0;
        set_err("null pointer");
        return WAH_ERR_ARG;
    }
    // the sums pass gives the tile bases and the totals ([words, groups] land in d_report[0..1] for a moment)
    const int rc = decode_common(d_comp, c_words, nullptr, 0, d_report, d_workspace, workspace_bytes, stream, true, false);
    if (rc != WAH_OK) return rc;
    const DecodeLayout l = decode_layout(c_words, workspace_bytes);
    char *ws = static_cast<char *>(d_workspace);
    // keep the totals aside (behind the tile bases), then build the report
    uint64_t *info = reinterpret_cast<uint64_t *>(ws + l.base_off) + (l.n_tiles + 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(info, d_report, 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess)
        e = wah::launch_validate(d_comp, c_words, reinterpret_cast<const uint64_t *>(ws + l.base_off), info, d_report, l.n_tiles, s);
    if (e != hipSuccess) {
        set_err("validate kernel launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

int wah_build_index_device(const uint32_t *d_comp, uint64_t c_words, uint64_t *d_segment_offsets, uint64_t offsets_capacity,
                           uint64_t *d_out_info, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (!d_segment_offsets || !d_out_info || offsets_capacity == 0) {
        g_err[0] = 0;
        set_err("null pointer or empty index");
        return WAH_ERR_ARG;
    }
    // the sums pass: tile bases in the workspace, [decoded words, groups] in d_out_info
    const int rc = decode_common(d_comp, c_words, nullptr, 0, d_out_info, d_workspace, workspace_bytes, stream, true, false);
    if (rc != WAH_OK) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (c_words == 0) { // no groups, no segments: the index is the single entry 0
        const hipError_t e = wah::launch_clear(d_segment_offsets, sizeof(uint64_t), s);
        if (e != hipSuccess) {
            set_err("clearing the index", e);
            return WAH_ERR_HIP;
        }
        return WAH_OK;
    }
    const DecodeLayout l = decode_layout(c_words, workspace_bytes);
    char *ws = static_cast<char *>(d_workspace);
    const hipError_t e = wah::launch_build_index(d_comp, c_words, reinterpret_cast<const uint64_t *>(ws + l.base_off), d_out_info,
                                                 d_segment_offsets, offsets_capacity, reinterpret_cast<uint32_t *>(ws + l.ctrl_off),
                                                 l.n_tiles, s);
    if (e != hipSuccess) {
        set_err("index kernel launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

// ---- merged (unsegmented) form -------------------------------------------------------------------------------
namespace {
struct MergeLayout {
    size_t decode_bytes, kept_off, pos_off, info_off, total;
};
MergeLayout merge_layout(uint64_t c_words) {
    MergeLayout l;
    const DecodeLayout d = decode_layout(c_words);
    l.decode_bytes = round256(d.total);
    l.kept_off = l.decode_bytes;
    l.info_off = round256(l.kept_off + (d.n_tiles + 2) * sizeof(uint64_t));
    l.pos_off = l.info_off + 256;
    l.total = round256(l.pos_off + (d.n_tiles + 2) * sizeof(uint64_t)); // (one position per TILE: round 3 kept one per word)
    return l;
}
} // namespace

size_t wah_merge_fills_workspace_bytes(uint64_t c_words) { return merge_layout(c_words).total; }

int wah_merge_fills_device(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out, uint64_t out_capacity_words,
                           uint64_t *d_out_words, void *d_workspace, size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!d_out_words || !d_workspace || (c_words && (!d_comp || !d_out))) {
        set_err("null pointer");
        return WAH_ERR_ARG;
    }
    const MergeLayout l = merge_layout(c_words);
    if (workspace_bytes < l.total) {
        set_err("workspace too small");
        return WAH_ERR_WORKSPACE;
    }
    char *ws = static_cast<char *>(d_workspace);
    uint64_t *info = reinterpret_cast<uint64_t *>(ws + l.info_off);
    // sums pass: tile bases and totals
    const int rc = decode_common(d_comp, c_words, nullptr, 0, info, d_workspace, l.decode_bytes, stream, true, false);
    if (rc != WAH_OK) return rc;
    const DecodeLayout d = decode_layout(c_words, l.decode_bytes);
    wah::MergeArgs a;
    a.comp = d_comp;
    a.c_words = c_words;
    a.n_tiles = d.n_tiles;
    a.tile_base = reinterpret_cast<const uint64_t *>(ws + d.base_off);
    a.info = info;
    a.tile_kept = reinterpret_cast<uint64_t *>(ws + l.kept_off);
    a.tile_first = reinterpret_cast<uint64_t *>(ws + l.pos_off);
    a.out = d_out;
    a.out_capacity = out_capacity_words;
    a.out_words = d_out_words;
    a.ctrl = reinterpret_cast<uint32_t *>(ws + d.ctrl_off);
    const hipError_t e = wah::launch_merge_fills(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_err("merge kernels launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

// ---- bitwise operations on two compressed bitmaps -------------------------------------------------------------
namespace {
struct BitopLayout {
    size_t bitmap_a, bitmap_b, info_a, info_b, ws_a, ws_b, ws_c, total;
    size_t ws_a_bytes, ws_b_bytes, ws_c_bytes;
    uint64_t decoded_capacity;
};
BitopLayout bitop_layout(uint64_t n_words, uint64_t a_words, uint64_t b_words) {
    BitopLayout l;
    l.decoded_capacity = n_words + 1; // ceil(31 G / 32) is n_words or n_words + 1
    const size_t bm = round256(l.decoded_capacity * sizeof(uint32_t));
    l.ws_a_bytes = wah_decompress_workspace_bytes(a_words, l.decoded_capacity);
    l.ws_b_bytes = wah_decompress_workspace_bytes(b_words, l.decoded_capacity);
    l.ws_c_bytes = wah_compress_workspace_bytes(n_words);
    l.bitmap_a = 0;
    l.bitmap_b = bm;
    l.info_a = 2 * bm;
    l.info_b = l.info_a + 256;
    l.ws_a = l.info_b + 256;
    l.ws_b = l.ws_a + round256(l.ws_a_bytes);
    l.ws_c = l.ws_b + round256(l.ws_b_bytes);
    l.total = l.ws_c + round256(l.ws_c_bytes);
    return l;
}
} // namespace

size_t wah_bitop_scratch_bytes(uint64_t n_words, uint64_t a_words, uint64_t b_words) {
    return bitop_layout(n_words, a_words, b_words).total;
}

int wah_bitop_device(int op, uint64_t n_words, const uint32_t *d_a, uint64_t a_words, const uint32_t *d_b,
                     uint64_t b_words, uint32_t *d_out, uint64_t out_capacity_words, uint64_t *d_out_words,
                     void *d_scratch, size_t scratch_bytes, void *stream) {
    g_err[0] = 0;
    if (op < WAH_OP_AND || op > WAH_OP_ANDNOT || !d_scratch || (reinterpret_cast<uintptr_t>(d_scratch) & 255u)) {
        set_err("bad operation or scratch pointer");
        return WAH_ERR_ARG;
    }
    const BitopLayout l = bitop_layout(n_words, a_words, b_words);
    if (scratch_bytes < l.total) {
        set_err("scratch too small");
        return WAH_ERR_WORKSPACE;
    }
    char *sc = static_cast<char *>(d_scratch);
    uint32_t *bm_a = reinterpret_cast<uint32_t *>(sc + l.bitmap_a), *bm_b = reinterpret_cast<uint32_t *>(sc + l.bitmap_b);
    uint64_t *info_a = reinterpret_cast<uint64_t *>(sc + l.info_a), *info_b = reinterpret_cast<uint64_t *>(sc + l.info_b);
    // (the decode workspaces inside the caller's scratch are cleared in front of every use: scratch needs no initialisation)
    int rc = decode_common(d_a, a_words, bm_a, l.decoded_capacity, info_a, sc + l.ws_a, l.ws_a_bytes, stream, true, true, true);
    if (rc == WAH_OK) rc = decode_common(d_b, b_words, bm_b, l.decoded_capacity, info_b, sc + l.ws_b, l.ws_b_bytes, stream, true, true, true);
    if (rc != WAH_OK) return rc;
    wah::PairCheck check;
    check.info_a = info_a;
    check.info_b = info_b;
    check.ctrl_a = reinterpret_cast<const uint32_t *>(sc + l.ws_a);
    check.ctrl_b = reinterpret_cast<const uint32_t *>(sc + l.ws_b);
    check.groups = wah_max_compressed_words(n_words);
    return compress_device_impl(bm_a, bm_b, op, &check, n_words, d_out, out_capacity_words, d_out_words, nullptr, sc + l.ws_c,
                                l.ws_c_bytes, stream, true);
}

int wah_bitop_status(void *d_scratch, uint64_t n_words, uint64_t a_words, uint64_t b_words, void *stream) {
    if (!d_scratch) return WAH_ERR_ARG;
    return read_status(static_cast<char *>(d_scratch) + bitop_layout(n_words, a_words, b_words).ws_c, stream);
}

int wah_gen_uniform_device(uint32_t *d_out, uint64_t n_words, uint64_t seed, uint64_t threshold, void *stream) {
    hipError_t e = wah::launch_gen_uniform(d_out, n_words, seed, threshold, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_err("generator launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

int wah_gen_clustered_device(uint32_t *d_out, uint64_t n_words, uint64_t seed, uint64_t threshold, void *stream) {
    hipError_t e = wah::launch_gen_clustered(d_out, n_words, seed, threshold, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_err("generator launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

int wah_copy_device(const uint32_t *d_in, uint32_t *d_out, uint64_t n_words, void *stream) {
    hipError_t e = wah::launch_copy(d_in, d_out, n_words, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_err("copy launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

size_t wah_decompress_segments_workspace_bytes(void) { return round256(wah::kCtlWords * sizeof(uint32_t)); }

int wah_decompress_segments_device(const uint32_t *d_comp, uint64_t c_words, const uint64_t *d_segment_offsets, uint64_t n_words,
                                   uint64_t first_segment, uint64_t n_segments, uint32_t *d_out, uint64_t out_capacity_words,
                                   void *d_workspace, size_t workspace_bytes, void *stream) {
    g_err[0] = 0;
    if (!d_workspace || !d_segment_offsets || (c_words && !d_comp) || (n_segments && !d_out)) {
        set_err("null pointer");
        return WAH_ERR_ARG;
    }
    if (n_words >= (1ull << 40) || c_words >= (1ull << 40) || (reinterpret_cast<uintptr_t>(d_comp) & 3u) ||
        (reinterpret_cast<uintptr_t>(d_out) & 3u) || (reinterpret_cast<uintptr_t>(d_workspace) & 255u)) {
        set_err("size out of range or misaligned pointer");
        return WAH_ERR_ARG;
    }
    if (workspace_bytes < wah_decompress_segments_workspace_bytes()) {
        set_err("workspace too small");
        return WAH_ERR_WORKSPACE;
    }
    const uint64_t groups = wah_max_compressed_words(n_words); // G = ceil(32 n / 31)
    const uint64_t all_segments = (groups + wah::kSegGroups - 1) / wah::kSegGroups;
    if (first_segment > all_segments || n_segments > all_segments - first_segment) {
        set_err("segment range outside the bitmap");
        return WAH_ERR_ARG;
    }
    const uint64_t out_words = wah_decoded_words(groups);
    const uint64_t range_end = (first_segment + n_segments) * wah::kSegWords < out_words ? (first_segment + n_segments) * wah::kSegWords : out_words;
    const uint64_t needed = n_segments ? range_end - first_segment * wah::kSegWords : 0;
    if (needed > out_capacity_words) {
        set_err("output capacity too small");
        return WAH_ERR_CAPACITY;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = wah::launch_clear(d_workspace, wah::kCtlWords * sizeof(uint32_t), s);
    if (e == hipSuccess) {
        wah::SegmentsArgs a = {};
        a.comp = d_comp;
        a.c_words = c_words;
        a.seg_offsets = d_segment_offsets;
        a.first_segment = first_segment;
        a.n_segments = n_segments;
        a.groups = groups;
        a.out_words = out_words;
        a.out = d_out;
        a.ctrl = static_cast<uint32_t *>(d_workspace);
        e = wah::launch_decode_segments(a, s);
    }
    if (e != hipSuccess) {
        set_err("segment decode launch", e);
        return WAH_ERR_HIP;
    }
    return WAH_OK;
}

namespace {
// scratch of wah_bitop_indexed_device: [control block of the combining pass][combined decoded bitmap][compress workspace]
constexpr uint64_t kRunsMaxWordsPerSeg = 112; // (bitop_runs_route below)
// (on the run-merge route the bitmap area holds the result's words per segment and per tile of segments instead)
struct BitopIndexedLayout {
    size_t bitmap, ws_c, total;
    size_t ws_c_bytes;
    uint64_t decoded_capacity;
    size_t runs_tiles, runs_temp; // run-merge route: the tile totals and the temporary, behind the segment counts at `bitmap`
};
BitopIndexedLayout bitop_indexed_layout(uint64_t n_words) {
    BitopIndexedLayout l;
    l.decoded_capacity = n_words + 1; // ceil(31 G / 32) is n_words or n_words + 1
    l.ws_c_bytes = wah_compress_workspace_bytes(n_words);
    l.bitmap = round256(wah::kCtlWords * sizeof(uint32_t));
    const uint64_t n_segments = ceil_div(wah_max_compressed_words(n_words), (uint64_t)wah::kSegGroups);
    l.runs_tiles = l.bitmap + round256(n_segments * sizeof(uint32_t));
    l.runs_temp = l.runs_tiles + round256((n_segments / 64 + 2) * sizeof(uint64_t));
    const size_t runs_end = l.runs_temp + round256((kRunsMaxWordsPerSeg * n_segments + 16) * sizeof(uint32_t));
    const size_t bitmap_end = l.bitmap + round256(l.decoded_capacity * sizeof(uint32_t));
    l.ws_c = runs_end > bitmap_end ? runs_end : bitmap_end;
    l.total = l.ws_c + round256(l.ws_c_bytes);
    return l;
}

// The run-merge route (wah_bitop_runs.hip) for operands of few words per segment: all operands together at most
// kRunsMaxWordsPerSeg words per segment on average -- the decode-based routes cost the same whatever the operands hold
// (about 0.25 ms per operand + 0.2 ms on a 1 GiB bitmap), a merge costs by the word (tools/bitop_density_time.py: two operands
// of 33 words per segment together 0.09 ms against 0.52, 63: 0.17, 122: 0.49-0.65 against 0.50-0.55; eight of 131 together:
// 1.1-1.3 against 1.6, of 250: 7.4).  false: not taken (the caller goes on with its own route); rc: what the call returns when taken.
bool bitop_runs_route(int op, uint64_t n_words, int n, const uint32_t *const *comp, const uint64_t *c_words, const uint64_t *const *offs,
                      uint32_t *d_out, uint64_t out_capacity_words, uint64_t *d_out_words, uint64_t *d_out_offsets, char *sc,
                      const BitopIndexedLayout &l, hipStream_t s, int *rc) {
    const uint64_t groups = wah_max_compressed_words(n_words);
    const uint64_t n_segments = ceil_div(groups, (uint64_t)wah::kSegGroups);
    uint64_t total = 0;
    for (int j = 0; j < n; ++j) total += c_words[j];
    g_last_bitop_route = WAH_BITOP_ROUTE_GROUPS;
    const char *force = wah::experiment_env("WAH_BITOP_ROUTE"); // (experiment builds: "runs" / "groups", tools/bitop_density_time.py)
    const uint64_t temp_words = (l.ws_c - l.runs_temp) / sizeof(uint32_t);
    if (n_words == 0 || total + 16 > temp_words || (force ? force[0] != 'r' : total > kRunsMaxWordsPerSeg * n_segments)) return false;
    g_last_bitop_route = WAH_BITOP_ROUTE_RUNS;
    if (!d_out_words || !d_out) {
        set_err("null pointer");
        *rc = WAH_ERR_ARG;
        return true;
    }
    wah::BitopRunsArgs a = {};
    for (int j = 0; j < n; ++j) {
        a.comp[j] = comp[j];
        a.c_words[j] = c_words[j];
        a.offs[j] = offs[j];
    }
    a.n = n;
    a.op = op;
    a.groups = groups;
    a.n_segments = n_segments;
    a.seg_count = reinterpret_cast<uint32_t *>(sc + l.bitmap);
    a.tile_total = reinterpret_cast<uint64_t *>(sc + l.runs_tiles);
    a.temp = reinterpret_cast<uint32_t *>(sc + l.runs_temp);
    a.out = d_out;
    a.out_capacity = out_capacity_words;
    a.out_words = d_out_words;
    a.out_offsets = d_out_offsets;
    a.ctrl = reinterpret_cast<uint32_t *>(sc);
    // (both control blocks wah_bitop_indexed_status reads: this route's, and the compress workspace's, which it does not use)
    hipError_t e = wah::launch_clear(sc + l.ws_c, wah::kCtlWords * sizeof(uint32_t), s);
    if (e == hipSuccess) e = wah::launch_bitop_runs(a, s);
    if (e != hipSuccess) {
        set_err("run-merge launch", e);
        *rc = WAH_ERR_HIP;
        return true;
    }
    *rc = WAH_OK;
    return true;
}
} // namespace

size_t wah_bitop_indexed_scratch_bytes(uint64_t n_words) { return bitop_indexed_layout(n_words).total; }

int wah_bitop_indexed_device(int op, uint64_t n_words, const uint32_t *d_a, uint64_t a_words, const uint64_t *d_a_offsets,
                             const uint32_t *d_b, uint64_t b_words, const uint64_t *d_b_offsets, uint32_t *d_out,
                             uint64_t out_capacity_words, uint64_t *d_out_words, uint64_t *d_out_offsets, void *d_scratch,
                             size_t scratch_bytes, void *stream) {
    g_err[0] = 0;
    if (op < WAH_OP_AND || op > WAH_OP_ANDNOT || !d_scratch || (reinterpret_cast<uintptr_t>(d_scratch) & 255u)) {
        set_err("bad operation or scratch pointer");
        return WAH_ERR_ARG;
    }
    if (!d_a_offsets || !d_b_offsets || (a_words && !d_a) || (b_words && !d_b) || n_words >= (1ull << 40) ||
        a_words >= (1ull << 40) || b_words >= (1ull << 40) || (reinterpret_cast<uintptr_t>(d_a) & 3u) ||
        (reinterpret_cast<uintptr_t>(d_b) & 3u)) {
        set_err("null or misaligned operand");
        return WAH_ERR_ARG;
    }
    const BitopIndexedLayout l = bitop_indexed_layout(n_words);
    if (scratch_bytes < l.total) {
        set_err("scratch too small");
        return WAH_ERR_WORKSPACE;
    }
    char *sc = static_cast<char *>(d_scratch);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // ONE kernel: both operands are walked segment by segment through their indexes, combined group by group in
    // registers, and the combined groups go straight into the compress passes -- no decoded bitmap is written or read
    // (the scratch's bitmap area stays unused on this route; the many-operand call still goes through it).
    hipError_t e = wah::launch_clear(sc, wah::kCtlWords * sizeof(uint32_t), s); // (read by wah_bitop_indexed_status)
    if (e != hipSuccess) {
        set_err("clearing the scratch", e);
        return WAH_ERR_HIP;
    }
    {
        const uint32_t *const comp[2] = {d_a, d_b};
        const uint64_t words[2] = {a_words, b_words};
        const uint64_t *const offs[2] = {d_a_offsets, d_b_offsets};
        int rc = WAH_OK;
        if (bitop_runs_route(op, n_words, 2, comp, words, offs, d_out, out_capacity_words, d_out_words, d_out_offsets, sc, l, s, &rc)) return rc;
    }
    wah::BitopOperands ops;
    ops.comp_a = d_a;
    ops.comp_b = d_b;
    ops.c_words_a = a_words;
    ops.c_words_b = b_words;
    ops.offs_a = d_a_offsets;
    ops.offs_b = d_b_offsets;
    ops.groups = wah_max_compressed_words(n_words);
    ops.op = (uint32_t)op;
    return compress_device_impl(nullptr, nullptr, 0, nullptr, n_words, d_out, out_capacity_words, d_out_words, d_out_offsets,
                                sc + l.ws_c, l.ws_c_bytes, stream, true, nullptr, &ops);
}

int wah_bitop_many_indexed_device(int op, uint64_t n_words, int n_operands, const uint32_t *const *d_streams,
                                  const uint64_t *stream_words, const uint64_t *const *d_offsets, uint32_t *d_out,
                                  uint64_t out_capacity_words, uint64_t *d_out_words, uint64_t *d_out_offsets, void *d_scratch,
                                  size_t scratch_bytes, void *stream) {
    g_err[0] = 0;
    if (op < WAH_OP_AND || op > WAH_OP_ANDNOT || !d_scratch || (reinterpret_cast<uintptr_t>(d_scratch) & 255u)) {
        set_err("bad operation or scratch pointer");
        return WAH_ERR_ARG;
    }
    if (n_operands < 1 || n_operands > wah::kMaxBitopOperands || !d_streams || !stream_words || !d_offsets ||
        n_words >= (1ull << 40)) {
        set_err("between 1 and 8 operands, each with its stream, length and index");
        return WAH_ERR_ARG;
    }
    wah::BitopManyArgs a = {};
    for (int j = 0; j < n_operands; ++j) {
        if (!d_offsets[j] || (stream_words[j] && !d_streams[j]) || stream_words[j] >= (1ull << 40) ||
            (reinterpret_cast<uintptr_t>(d_streams[j]) & 3u)) {
            set_err("null or misaligned operand");
            return WAH_ERR_ARG;
        }
        a.comp[j] = d_streams[j];
        a.c_words[j] = stream_words[j];
        a.offs[j] = d_offsets[j];
    }
    const BitopIndexedLayout l = bitop_indexed_layout(n_words);
    if (scratch_bytes < l.total) {
        set_err("scratch too small");
        return WAH_ERR_WORKSPACE;
    }
    char *sc = static_cast<char *>(d_scratch);
    uint32_t *combined = reinterpret_cast<uint32_t *>(sc + l.bitmap);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t groups = wah_max_compressed_words(n_words);
    hipError_t e = wah::launch_clear(sc, wah::kCtlWords * sizeof(uint32_t), s);
    if (e == hipSuccess) {
        int rc = WAH_OK;
        if (bitop_runs_route(op, n_words, n_operands, d_streams, stream_words, d_offsets, d_out, out_capacity_words, d_out_words,
                             d_out_offsets, sc, l, s, &rc))
            return rc;
    }
    if (e == hipSuccess) {
        a.g.first_segment = 0;
        a.g.n_segments = (groups + wah::kSegGroups - 1) / wah::kSegGroups;
        a.g.groups = groups;
        a.g.out_words = wah_decoded_words(groups);
        a.g.out = combined;
        a.g.ctrl = reinterpret_cast<uint32_t *>(sc);
        a.n = n_operands;
        a.op = op;
        e = wah::launch_bitop_many_segments(a, s);
    }
    if (e != hipSuccess) {
        set_err("combining pass launch", e);
        return WAH_ERR_HIP;
    }
    return compress_device_impl(combined, nullptr, 0, nullptr, n_words, d_out, out_capacity_words, d_out_words, d_out_offsets,
                                sc + l.ws_c, l.ws_c_bytes, stream, true);
}

int wah_bitop_indexed_status(void *d_scratch, uint64_t n_words, void *stream) {
    if (!d_scratch) return WAH_ERR_ARG;
    const int rc = read_status(d_scratch, stream); // the combining pass: operands that are not segmented streams of n_words
    return rc != WAH_OK ? rc : read_status(static_cast<char *>(d_scratch) + bitop_indexed_layout(n_words).ws_c, stream);
}

// ---------------------------------------------------------------------------
// host-pointer entry points (the reference's API)
// ---------------------------------------------------------------------------
uint32_t *wah_compress(const uint32_t *data_host, uint64_t n_words, uint64_t *out_words, float *t_to_device_ms,
                       float *t_device_ms, float *t_from_device_ms) {
    g_err[0] = 0;
    if (n_words && !data_host) {
        set_err("null input");
        std::fprintf(stderr, "wah: compress() called with a null input\n");
        return nullptr;
    }
    HostCall hc;
    if (!hc.init()) {
        set_err("hipEventCreate failed (no usable GPU?)");
        std::fprintf(stderr, "wah: %s\n", g_err);
        return nullptr;
    }
    float t_in = 0.f, t_dev = 0.f, t_out = 0.f;

    // phase 1: allocate + H2D (compress.cu:57-120)
    hc.start();
    const uint64_t cap = wah_max_compressed_words(n_words);
    void *d_in = nullptr, *d_out = nullptr, *d_ws = nullptr;
    // (one word more than the data: the buffer is decompress()'s output buffer next, and a bitmap whose length is not a
    //  multiple of 31 decodes to one padding word more, decompress.cu:84-93)
    if (!hc.alloc(0, &d_in, (n_words + 1) * sizeof(uint32_t), "space for the data")) return nullptr;
    if (!hc.alloc(1, &d_out, cap * sizeof(uint32_t), "space for the compressed output")) return nullptr;
    bool fresh_ws = false;
    if (!hc.alloc(2, &d_ws, wah_compress_workspace_bytes(n_words), "workspace", &fresh_ws)) return nullptr;
    // a kept workspace is named with ITS size, call after call, whatever this call needs of it: where a launch keeps its
    // entries follows from the size it is given (include/wah.h), and entries that moved with the input's size would let
    // one call's tables be read as another call's published entries
    const size_t ws_bytes = hc.cache.cap[2];
    // the compress workspace is zeroed once; from then on the kernel keeps it up itself (launch epochs)
    if (fresh_ws && wah_workspace_init_device(d_ws, ws_bytes, nullptr) != WAH_OK) return nullptr;
    uint64_t *d_cnt = reinterpret_cast<uint64_t *>(static_cast<uint32_t *>(d_ws) + wah::kCtlResult); // beside the error word
    if (n_words && !hip_ok(hipMemcpy(d_in, data_host, n_words * sizeof(uint32_t), hipMemcpyHostToDevice), "copy input"))
        return nullptr;
    t_in = hc.stop();

    // phase 2: device work (compress.cu:125-172)
    hc.start();
    uint64_t *const host_result = reinterpret_cast<uint64_t *>(hc.cache.pinned) + 4; // behind the copy's landing area
    host_result[0] = 0;
    int rc = compress_device_impl(static_cast<uint32_t *>(d_in), nullptr, 0, nullptr, n_words, static_cast<uint32_t *>(d_out), cap, d_cnt,
                                  nullptr, d_ws, ws_bytes, nullptr, false, n_words ? host_result : nullptr);
    hc.mark();
    uint64_t c = 0;
    if (rc == WAH_OK) rc = wait_host_result(host_result, d_ws, hc.cache.pinned, &c, 1); // status + size: one wait, no copy
    if (rc == WAH_OK && test_timeout_hook()) rc = WAH_ERR_TIMEOUT;
    if (rc == WAH_ERR_TIMEOUT) {
        // A bounded wait inside the kernel expired: a workgroup this launch depended on did not get to run in time (a GPU
        // shared in a way the arrival tickets do not cover, a preempted queue).  The no-wait route has no such
        // dependency: three launches, the bitmap read twice, the same stream.
        std::fprintf(stderr, "wah: compress: in-kernel wait expired, taking the no-wait route\n");
        host_result[0] = 0;
        rc = compress_device_impl(static_cast<uint32_t *>(d_in), nullptr, 0, nullptr, n_words, static_cast<uint32_t *>(d_out), cap, d_cnt,
                                  nullptr, d_ws, ws_bytes, nullptr, false, host_result, nullptr, false, true);
        hc.mark(); // the device phase ends with this route's last launch
        if (rc == WAH_OK) rc = wait_host_result(host_result, d_ws, hc.cache.pinned, &c, 1);
        // (the abandoned launch left its tickets and its epoch half way: the next call starts from a fresh workspace,
        //  whatever became of this one)
        if (wah_workspace_init_device(d_ws, ws_bytes, nullptr) != WAH_OK && rc == WAH_OK) rc = WAH_ERR_HIP;
    }
    if (rc != WAH_OK) {
        std::fprintf(stderr, "wah: compress failed: %s\n", g_err);
        return nullptr;
    }
    t_dev = hc.since_mark();

    // phase 3: D2H + free (compress.cu:177-202)
    hc.start();
    bool huge = false;
    uint32_t *host = static_cast<uint32_t *>(host_result_alloc(c * sizeof(uint32_t), &huge));
    if (!host) {
        set_err("host malloc failed");
        return nullptr;
    }
    if (c && !copy_to_fresh_host(host, d_out, c * sizeof(uint32_t), huge, "copy final output")) {
        std::free(host);
        return nullptr;
    }
    hc.release();
    t_out = hc.stop();

    if (out_words) *out_words = c;
    if (t_to_device_ms) *t_to_device_ms = t_in;
    if (t_device_ms) *t_device_ms = t_dev;
    if (t_from_device_ms) *t_from_device_ms = t_out;
    return host;
}

uint32_t *wah_decompress(const uint32_t *comp_host, uint64_t c_words, uint64_t *out_words, float *t_to_device_ms,
                         float *t_device_ms, float *t_from_device_ms) {
    g_err[0] = 0;
    if (c_words && !comp_host) {
        set_err("null input");
        std::fprintf(stderr, "wah: decompress() called with a null input\n");
        return nullptr;
    }
    HostCall hc;
    if (!hc.init()) {
        set_err("hipEventCreate failed (no usable GPU?)");
        std::fprintf(stderr, "wah: %s\n", g_err);
        return nullptr;
    }
    float t_in = 0.f, t_dev = 0.f, t_out = 0.f;

    // phase 1: allocate + H2D (decompress.cu:34-54)
    hc.start();
    void *d_comp = nullptr, *d_ws0 = nullptr;
    if (!hc.alloc(1, &d_comp, c_words * sizeof(uint32_t), "space for the compressed data")) return nullptr;
    bool fresh_ws = false;
    if (!hc.alloc(4, &d_ws0, wah_decompress_workspace_bytes(c_words, 0), "scan workspace", &fresh_ws)) return nullptr;
    const size_t ws0 = hc.cache.cap[4]; // (named with its own size every time: see wah_compress)
    // zeroed once; from then on the sums kernel keeps it up itself (launch epochs)
    if (fresh_ws && wah_workspace_init_device(d_ws0, ws0, nullptr) != WAH_OK) return nullptr;
    uint64_t *d_info = reinterpret_cast<uint64_t *>(static_cast<uint32_t *>(d_ws0) + wah::kCtlResult); // beside the error word
    // which decoder: a look at the stream itself, which lies in host memory -- the group counts of up to 65 536 words spread
    // evenly over it.  One pass up to 7 groups per word; the two launches from there to 128 (tools/report.py, 992 MiB, device
    // phase: one bit in 2^11 = 32 groups per word 0.248 ms in one pass against 0.223, 2^12 0.223 / 0.210, 2^14 0.209 / 0.203 --
    // through this boundary the one-pass kernel's longer workgroups show on a stream of a few megabytes); above: the same.
    // (BEFORE the copy: a device left idle for the millisecond this takes starts its next kernel slower)
    // The same sample says how large the bitmap will be, to a percent or so (exactly, for a stream of up to 65 536 words): see
    // phase 2.
    int route = 1;
    uint64_t expect_words = 0; // decoded words the sample predicts (0: no prediction)
    if (c_words) {
        uint64_t sampled = 0, sample_groups = 0;
        const uint64_t stride = c_words > 65536 ? c_words / 65536 : 1;
        for (uint64_t i = 0; i < c_words; i += stride, ++sampled)
            sample_groups += (comp_host[i] & wah::kFillZero) ? (comp_host[i] & wah::kCountMask) : 1u;
        const uint64_t per_word = sample_groups / (sampled ? sampled : 1);
        route = (per_word <= 7 || per_word >= 128) ? 1 : 2;
        const unsigned __int128 groups = (unsigned __int128)sample_groups * c_words / (sampled ? sampled : 1);
        if (groups < ((unsigned __int128)1 << 40)) expect_words = ((uint64_t)groups * 31u + 31u) / 32u;
    }
    if (c_words && !hip_ok(hipMemcpy(d_comp, comp_host, c_words * sizeof(uint32_t), hipMemcpyHostToDevice), "copy input"))
        return nullptr;
    t_in = hc.stop();

    // phase 2: device work (decompress.cu:56-122): size scan, allocate, scan + expand
    hc.start();
    uint64_t info[2] = {0, 0};
    void *d_out = nullptr;
    int rc = WAH_OK;
    bool expanded = false, no_wait = false;
    // An output buffer kept from an earlier call (the reference's callers decompress in loops, source.cpp:70): expand
    // right behind the scan, into what is there -- the expand kernel compares the decoded size with the capacity on the
    // device and writes nothing if it does not fit.  One host round trip for the whole phase instead of two.
    // Without one -- a process's first call, a bitmap larger than any before -- the reference's order (scan, read the size back,
    // allocate, expand: decompress.cu:72-100) would read the stream twice and wait for the device twice: instead a buffer of
    // the size the SAMPLE predicts, plus a sixteenth (+ 64 Ki words: a sample of 65 536 words is good to about a percent), and
    // the same single pass.  A prediction that was too small costs what a kept buffer that is too small costs: the decoder
    // reports WAH_ERR_CAPACITY and the call goes by the book below; a buffer the device cannot give: by the book, too.
    size_t kept_words = hc.cache.buf[0] ? hc.cache.cap[0] / sizeof(uint32_t) : 0;
    if (c_words && expect_words && kept_words < expect_words) {
        const uint64_t want = expect_words + expect_words / 16u + 65536u;
        void *grown = nullptr;
        if (hc.alloc(0, &grown, want * sizeof(uint32_t), "space for the result", nullptr, true)) kept_words = want;
        else kept_words = 0; // (alloc has dropped the smaller buffer)
    }
    if (c_words && kept_words) {
        rc = decode_common(static_cast<uint32_t *>(d_comp), c_words, static_cast<uint32_t *>(hc.cache.buf[0]), kept_words, d_info, d_ws0,
                           ws0, nullptr, true, true, false, nullptr, false, route);
        hc.mark();
        if (rc == WAH_OK) rc = read_status_packed(d_ws0, hc.cache.pinned, info, 2); // status + sizes in one copy
        if (rc == WAH_OK && test_timeout_hook()) rc = WAH_ERR_TIMEOUT;
        if (rc == WAH_OK) {
            d_out = hc.cache.buf[0];
            expanded = true;
        } else if (rc == WAH_ERR_TIMEOUT) {
            no_wait = true; // (by the book below, without waits)
        } else if (rc != WAH_ERR_CAPACITY) {
            std::fprintf(stderr, "wah: decompress failed: %s\n", g_err);
            return nullptr;
        } // (too small: by the book below)
    }
    if (!expanded) {
        uint64_t *const host_result = reinterpret_cast<uint64_t *>(hc.cache.pinned) + 4; // behind the copy's landing area
        host_result[0] = 0;
        rc = decode_common(static_cast<uint32_t *>(d_comp), c_words, nullptr, 0, d_info, d_ws0, ws0, nullptr, true, false, false,
                           c_words ? host_result : nullptr, no_wait);
        if (rc == WAH_OK) rc = wait_host_result(host_result, d_ws0, hc.cache.pinned, info, 2); // status + sizes: one wait, no copy
        if (rc == WAH_OK && !no_wait && test_timeout_hook()) rc = WAH_ERR_TIMEOUT;
        if (rc == WAH_ERR_TIMEOUT && !no_wait) {
            // A bounded wait inside the sums kernel expired (see wah_compress): the no-wait route has no such dependency --
            // per-tile totals, then one scan launch, the reference's own shape (decompress.cu:66-80).
            std::fprintf(stderr, "wah: decompress: in-kernel wait expired, taking the no-wait route\n");
            no_wait = true;
            host_result[0] = 0;
            rc = decode_common(static_cast<uint32_t *>(d_comp), c_words, nullptr, 0, d_info, d_ws0, ws0, nullptr, true, false, false,
                               c_words ? host_result : nullptr, true);
            if (rc == WAH_OK) rc = wait_host_result(host_result, d_ws0, hc.cache.pinned, info, 2);
        }
        if (rc != WAH_OK) {
            std::fprintf(stderr, "wah: decompress failed: %s\n", g_err);
            return nullptr;
        }
        if (!hc.alloc(0, &d_out, info[0] * sizeof(uint32_t), "space for the result")) return nullptr;
        // the tile bases of the scan are still in the workspace: expand only
        rc = wah_decompress_expand_device(static_cast<uint32_t *>(d_comp), c_words, static_cast<uint32_t *>(d_out), info[0], d_info,
                                          d_ws0, ws0, nullptr);
        hc.mark();
        if (rc == WAH_OK) rc = read_status_packed(d_ws0, hc.cache.pinned, nullptr, 0);
        if (rc != WAH_OK) {
            std::fprintf(stderr, "wah: decompress failed: %s\n", g_err);
            return nullptr;
        }
    }
    const uint64_t n_out = info[0], groups = info[1];
    t_dev = hc.since_mark();
    // (after a timeout the abandoned launch left its tickets and its epoch half way: the next call starts from a fresh workspace)
    if (no_wait && wah_workspace_init_device(d_ws0, ws0, nullptr) != WAH_OK) return nullptr;

    // phase 3: D2H + free (decompress.cu:124-131).  The reference hands back a buffer of G words
    // (one per group) of which ceil(31 G / 32) are meaningful; we keep the size and zero the rest.
    hc.start();
    bool huge = false;
    uint32_t *host = static_cast<uint32_t *>(host_result_alloc(groups * sizeof(uint32_t), &huge));
    if (!host) {
        set_err("host malloc failed");
        return nullptr;
    }
    if (groups > n_out) std::memset(host + n_out, 0, (groups - n_out) * sizeof(uint32_t));
    if (groups == 0) host[0] = 0;
    if (n_out && !copy_to_fresh_host(host, d_out, n_out * sizeof(uint32_t), huge, "copy final output")) {
        std::free(host);
        return nullptr;
    }
    hc.release();
    t_out = hc.stop();

    if (out_words) *out_words = n_out;
    if (t_to_device_ms) *t_to_device_ms = t_in;
    if (t_device_ms) *t_device_ms = t_dev;
    if (t_from_device_ms) *t_from_device_ms = t_out;
    return host;
}

} // extern "C"

// ---------------------------------------------------------------------------
// The reference's own symbols, C++ linkage (compress.h:12-18, decompress.h:11-17):
// _Z8compressPjyPyPfS1_S1_ and _Z10decompressPjyPyPfS1_S1_.
// ---------------------------------------------------------------------------
unsigned int *compress(unsigned int *data_cpu, unsigned long long int dataSize, unsigned long long int *outputSize,
                       float *pTransferToDeviceTime, float *pCompressionTime, float *ptranserFromDeviceTime) {
    uint64_t c = 0;
    uint32_t *r = wah_compress(data_cpu, dataSize, &c, pTransferToDeviceTime, pCompressionTime, ptranserFromDeviceTime);
    if (r && outputSize) *outputSize = c;
    return r;
}

unsigned int *decompress(unsigned int *data, unsigned long long int dataSize, unsigned long long int *outSize,
                         float *pTransferToDeviceTime, float *pCompressionTime, float *ptranserFromDeviceTime) {
    uint64_t n = 0;
    uint32_t *r = wah_decompress(data, dataSize, &n, pTransferToDeviceTime, pCompressionTime, ptranserFromDeviceTime);
    if (r && outSize) *outSize = n;
    return r;
}
