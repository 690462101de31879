// wah_device.hpp -- wavefront / LDS / scan helpers shared by the kernel files (gfx950, wave64).
// Everything lives in an anonymous namespace: every translation unit gets its own copy, nothing is exported.
#ifndef WAH_DEVICE_HPP_
#define WAH_DEVICE_HPP_

#include "wah_internal.hpp"

namespace wah {
namespace {

using u32 = uint32_t;
using u64 = uint64_t;

// Diagnostic build only (make diag): per-phase cycle totals of each workgroup's thread 0, added into the unused
// tail of the control block.  The product build contains no stamps.
#ifdef WAH_DIAG
#define WAH_STAMP_DECL u64 dg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 dg_big[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 dg_prev = __builtin_readcyclecounter();
#define WAH_STAMP(i)                                      \
    do {                                                  \
        const u64 dg_now = __builtin_readcyclecounter();  \
        dg_acc[i] += dg_now - dg_prev;                    \
        dg_big[i] += (dg_now - dg_prev) > 4000u;          \
        dg_prev = dg_now;                                 \
    } while (0)
#define WAH_STAMP_FLUSH(ctrl)                                                                      \
    do {                                                                                           \
        if (threadIdx.x == 0)                                                                      \
            for (int i = 0; i < 8; ++i) {                                                          \
                atomicAdd(reinterpret_cast<unsigned long long *>(ctrl + 192) + i, (unsigned long long)dg_acc[i]); \
                atomicAdd(reinterpret_cast<unsigned long long *>(ctrl + 192) + 16 + i, (unsigned long long)dg_big[i]); \
            }                                                                                      \
    } while (0)
#else
#define WAH_STAMP_DECL
#define WAH_STAMP(i) asm volatile("; WAH_MARK " #i ::: "memory")
#define WAH_STAMP_FLUSH(ctrl)
#endif

constexpr u32 kMaxSpins = 1u << 21; // bound of every in-kernel wait (polls with a sleep in between: seconds)

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
// wave-uniform by construction; readfirstlane tells the compiler so (keeps masks and offsets in SGPRs)
__device__ __forceinline__ u32 wave_id() { return (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ u64 wave_sum(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ u32 wave_sum32(u32 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ u64 wave_scan_incl(u64 v, u32 lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u64 t = __shfl_up(v, off);
        if (lane >= (u32)off) v += t;
    }
    return v;
}

// inclusive prefix sum of a u32 across the 64 lanes with DPP only (no LDS crossbar): Hillis-Steele inside each
// row of 16 lanes (row_shr 1,2,4,8, zero fill), then row_bcast:15 into rows 1 and 3, then row_bcast:31 into rows 2-3
__device__ __forceinline__ u32 wave_scan_incl32(u32 v) {
    v += __builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false);
    return v;
}

// sum of a u32 over the 64 lanes as a wave-uniform value, DPP only (wave_sum32's six trips through the LDS crossbar are
// a chain of six LDS latencies)
__device__ __forceinline__ u32 wave_total32(u32 v) { return (u32)__builtin_amdgcn_readlane((int)wave_scan_incl32(v), 63); }

// inclusive prefix MAXIMUM of a u32 across the 64 lanes, same DPP pattern (missing sources read as 0)
__device__ __forceinline__ u32 wave_scan_max32(u32 v) {
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0u, v, 0x111, 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0u, v, 0x112, 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0u, v, 0x114, 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0u, v, 0x118, 0xf, 0xf, true));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0u, v, 0x142, 0xa, 0xf, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0u, v, 0x143, 0xc, 0xf, false));
    return v;
}

// Values read back from LDS are wave-uniform here by construction; readfirstlane tells the compiler so, which
// keeps everything derived from them (segment numbers, masks, offsets, branches) on the scalar unit.
__device__ __forceinline__ u32 uniform32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uniform64(u64 v) {
    return ((u64)uniform32((u32)(v >> 32)) << 32) | uniform32((u32)v);
}


typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

using lds_u32_ptr = __attribute__((address_space(3))) u32 *;
using lds_u64_ptr = __attribute__((address_space(3))) u64 *;
using lds_u16_ptr = __attribute__((address_space(3))) unsigned short *;
using lds_u8_ptr = __attribute__((address_space(3))) unsigned char *;


// Buffer descriptor over `bytes` bytes at p (raw, stride 0): loads past the end return 0, stores past the end are
// dropped -- the hardware does the tail padding (F5) and the capacity clipping, and addresses become
// descriptor + 32-bit lane offset + immediate, with no 64-bit vector arithmetic per access.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x27000);
}

// ---------------------------------------------------------------------------
// Launch epochs: how the scan areas of the tile kernels (compress_tile_kernel, decode_sums_kernel) get by without
// being cleared.  Everything a launch publishes is stamped with the launch epoch kept in the control block: read by
// every workgroup at its start, advanced by the LAST tile once its scan is complete -- by then every other tile has
// published, hence started.  A zeroed workspace is epoch 0 = "nothing valid".  When the epoch space is used up, the
// next launch has tile 0 clear the scan area while the others wait for it (tile 0 has the smallest blockIdx: it is
// running).  Called by ALL threads of the workgroup (it contains barriers in the wrap case).
// ---------------------------------------------------------------------------
struct LaunchEpoch {
    u32 epoch;
    bool wrap, bad;
};

// Tile number of a workgroup of a tile kernel = the order in which workgroups START RUNNING (one agent-scope atomic per
// workgroup), never blockIdx: a tile only ever waits for tiles with smaller numbers, and those have been drawn, so they
// are running (or done) -- whatever else shares the GPU.  With tile = blockIdx two such kernels running side by side
// (two streams, two processes) can starve each other for good: the XCDs dispatch their shares of a grid independently,
// so kernel A's waiting tiles can fill one XCD while its lowest tile needs a slot on another that kernel B's waiting
// tiles fill, and vice versa (seen: multi-second stalls with two ranks on one card).  Called by ALL threads.
__device__ __forceinline__ u32 draw_tile(u32 *ctrl, u32 *s_tile) {
    if (threadIdx.x == 0) *s_tile = __hip_atomic_fetch_add(ctrl + kCtlStart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return uniform32(*s_tile);
}

__device__ __forceinline__ LaunchEpoch launch_epoch_begin(u32 *ctrl, u32 tile, u32 n_tiles, u32 *scan_area, u64 scan_words, int keep_error) {
    LaunchEpoch le;
    const u32 stored = uniform32(ctrl[kCtlEpoch]);
    const u32 magic = uniform32(ctrl[kCtlMagic]);
    // neither a zeroed nor a used workspace: a foreign magic word, an epoch no launch can have left, or a ticket counter
    // that did not start at zero -- a tile number outside the grid, checked BEFORE anything is indexed with it (recycled
    // memory often reads as magic == 0 with garbage elsewhere)
    le.bad = (magic != 0u && magic != kWorkspaceMagic) || stored > kEpochWrap || tile >= n_tiles;
    le.wrap = stored >= kEpochWrap;
    le.epoch = (stored == 0u || le.wrap) ? 1u : stored;
    if (le.bad) {
        if (threadIdx.x == 0) atomicOr(ctrl + kCtlError, kErrWorkspace);
        return le;
    }
    if (le.wrap) {
        const u32 wraps = uniform32(ctrl[kCtlWraps]);
        if (tile == 0) {
            for (u64 k = threadIdx.x; k < scan_words; k += blockDim.x)
                __hip_atomic_store(scan_area + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(ctrl + kCtlClearDone, wraps + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) {
                u32 spins = 0;
                while (__hip_atomic_load(ctrl + kCtlClearDone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != wraps + 1u) {
                    if (++spins > kMaxSpins) {
                        atomicOr(ctrl + kCtlError, kErrTimeout);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
        }
    }
    if (tile == 0 && threadIdx.x == 0 && !keep_error) {
        // a new launch: forget the previous one's status.  Completed before this tile publishes anything, and every
        // error of this launch is raised by a tile that has seen something published.
        __hip_atomic_store(ctrl + kCtlError, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return le;
}

// one lane of the last tile, after its scan: every other tile has published its granule, so it has read the epoch
__device__ __forceinline__ void launch_epoch_end(u32 *ctrl, const LaunchEpoch &le) {
    if (le.wrap) __hip_atomic_store(ctrl + kCtlWraps, ctrl[kCtlWraps] + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ctrl + kCtlStart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // every tile number has been drawn
    __hip_atomic_store(ctrl + kCtlMagic, kWorkspaceMagic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(ctrl + kCtlEpoch, le.epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kAuxSc1 = 16; // buffer load cache policy: sc1 = agent scope (served past the XCD-private caches)

// groups a compressed word expands to
__device__ __forceinline__ u32 word_groups(u32 w) {
    return (w & kFillZero) ? (w & kCountMask) : 1u; // kernels.cu:298-304
}

// expand / checker / merge kernels: one workgroup of four wavefronts per 4096-word tile
constexpr int kExpandThreads = kExpandWaves * 64;                 // 256
constexpr int kExpandWordsPerThread = kScanTileWords / kExpandThreads; // 16
constexpr u32 kCoarse = kScanTileWords / 64;                      // coarse prefix: one entry per 64 words

} // namespace
} // namespace wah

#endif // WAH_DEVICE_HPP_
