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

constexpr u32 kMaxSpins = 1u << 21; // bounded look-back wait

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


// Arrival ticket: the order in which workgroups actually start running.  Tiles are dealt round robin in THIS
// order (never in blockIdx order, which says nothing about dispatch), so a workgroup only ever waits for
// workgroups that are already running.
__device__ __forceinline__ u32 draw_arrival(u32 *ctrl) {
    return __hip_atomic_fetch_add(ctrl + kCtlStart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// Generation scan: the one-hop offset resolution of the persistent compress kernel.
//
// With the static round robin (tile = slot + generation * G) the G tiles of a generation are in flight together,
// so a chained look-back needs several store->poll hops per generation and every workgroup stalls for all of them.
// Here each tile publishes ONE 4-byte granule {valid, words} in its generation's row, and reads
//     row[gen][0 .. slot)     -> words in front of it inside its generation            (poll until all valid)
//     row[gen-1](slot .. G)   -> the rest of the previous generation's total          (normally valid already)
// Every workgroup carries the running total of all earlier generations in registers (GenScan), so there is no
// prefix descriptor, no chain and exactly one hop: publish, poll once, done.  Granules are naturally aligned
// 4-byte words written and read with agent-scope relaxed atomics (sc1), the data is the flag.
// ---------------------------------------------------------------------------
constexpr u32 kGenValid = 0x80000000u;

struct GenScan {
    u64 gen_base;   // words of all generations before the current one
    u32 below_prev; // previous generation: words of slots below mine
    u32 own_prev;   // previous generation: my own words
};

__device__ __forceinline__ void publish_generation(u32 *gdesc, u32 gen, u32 slot, u32 row_stride, u32 aggregate) {
    __hip_atomic_store(gdesc + (u64)gen * row_stride + slot, kGenValid | aggregate, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// Whole wavefront.  Returns the number of words in front of tile (gen, slot); `aggregate` is that tile's own count
// (already published).
__device__ __forceinline__ u64 resolve_generation(const u32 *gdesc, u32 gen, u32 slot, u32 G, u32 row_stride,
                                                  u32 aggregate, GenScan &st, u32 lane, u32 *ctrl) {
    const u32 *cur = gdesc + (u64)gen * row_stride;
    const u32 *prv = cur - row_stride; // only dereferenced when gen > 0

    bool need_prev = gen > 0 && slot + 1 < G, need_cur = slot > 0;
    u32 above = 0, below = 0, spins = 0;
    while (need_prev || need_cur) {
        u32 sum_cur = 0, sum_prev = 0;
        bool bad_cur = false, bad_prev = false;
        // lane l looks at entries 2l, 2l+1 (+128 per trip) as one 8-byte load per row
        for (u32 k0 = 2u * lane; k0 < G; k0 += 128u) {
            if (need_cur && k0 < slot) {
                const u64 v = __hip_atomic_load(reinterpret_cast<const u64 *>(cur + k0), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
                const u32 e0 = (u32)v, e1 = (u32)(v >> 32);
                bad_cur |= !(e0 & kGenValid);
                sum_cur += e0 & ~kGenValid;
                if (k0 + 1 < slot) {
                    bad_cur |= !(e1 & kGenValid);
                    sum_cur += e1 & ~kGenValid;
                }
            }
            if (need_prev && k0 + 1 > slot) {
                const u64 v = __hip_atomic_load(reinterpret_cast<const u64 *>(prv + k0), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
                const u32 e0 = (u32)v, e1 = (u32)(v >> 32);
                if (k0 > slot) {
                    bad_prev |= !(e0 & kGenValid);
                    sum_prev += e0 & ~kGenValid;
                }
                if (k0 + 1 < G) {
                    bad_prev |= !(e1 & kGenValid);
                    sum_prev += e1 & ~kGenValid;
                }
            }
        }
        bool progressed = false;
        if (need_cur && !__any(bad_cur)) {
            below = uniform32(wave_sum32(sum_cur));
            need_cur = false;
            progressed = true;
        }
        if (need_prev && !__any(bad_prev)) {
            above = uniform32(wave_sum32(sum_prev));
            need_prev = false;
            progressed = true;
        }
        if (!progressed) {
            if (++spins > kMaxSpins) {
                if (lane == 0) atomicOr(ctrl + kCtlError, kErrTimeout);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    if (gen > 0) st.gen_base += (u64)st.below_prev + st.own_prev + above;
    st.below_prev = below;
    st.own_prev = aggregate;
    return st.gen_base + below;
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

// Tile assignment is a static round robin over the workgroups in ARRIVAL order: the workgroup that draws arrival
// ticket v processes tiles v, v + G, v + 2G, ... (G = grid size).  Every generation of G consecutive tiles is
// then in flight at once and no workgroup ever holds a tile that sits below a tile somebody else is already
// waiting behind (dynamic tickets drawn ahead of time do exactly that, and serialise the scan).  It needs all
// G workgroups to be resident together: the host sizes G from a residency census of this very kernel
// (census mode below), and every wait is bounded, so a lost workgroup ends in WAH_ERR_TIMEOUT, never in a hang.
// Hand-offs inside the workgroup go through LDS words, not s_barrier: a wave only ever waits for the one thing it
// needs.  LDS operations of a wave execute in order and the LDS is coherent inside the CU, so "write data, then
// write flag" / "see flag, then read data" is enough; the waits below only drain the LDS counter (lgkmcnt),
// never the vector-memory counter -- the prefetched loads stay in flight.
// (explicit LDS address space + relaxed workgroup atomics: a volatile access through a generic pointer would be
//  emitted as a FLAT instruction, which counts on the vector-memory counter as well and forces vmcnt(0) waits)
__device__ __forceinline__ u32 lds_ld(const u32 *p) {
    return __hip_atomic_load((lds_u32_ptr)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ u64 lds_ld64(const u64 *p) {
    return __hip_atomic_load((lds_u64_ptr)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(u32 *p, u32 v) {
    __hip_atomic_store((lds_u32_ptr)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_publish(u32 *flag, u32 value) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_st(flag, value);
}
// A waiting wave must not compete with the working ones: the hardware favours the OLDEST wave of a SIMD, and the
// oldest waves are exactly the ones that finish first and wait (measured: a busy spin made the youngest worker of a
// SIMD take 1.75x as long as the oldest).  So: lowest priority and a short sleep between polls (128 cycles; longer ones only delay the hand-over).
__device__ __forceinline__ bool lds_wait(const u32 *flag, u32 value, u32 *ctrl, u32 lane) {
    if (lds_ld(flag) != value) {
        __builtin_amdgcn_s_setprio(0);
        for (u32 spins = 0; lds_ld(flag) != value;) {
            if (++spins > kMaxSpins) {
                if (lane == 0) atomicOr(ctrl + kCtlError, kErrTimeout);
                __builtin_amdgcn_s_setprio(1);
                return false;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_s_setprio(1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return true;
}

// same, for a monotonic progress counter: wait until it has reached `value`
__device__ __forceinline__ bool lds_wait_reached(const u32 *counter, u32 value, u32 *ctrl) {
    if ((int)(lds_ld(counter) - value) < 0) {
        __builtin_amdgcn_s_setprio(0);
        for (u32 spins = 0; (int)(lds_ld(counter) - value) < 0;) {
            if (++spins > kMaxSpins) {
                atomicOr(ctrl + kCtlError, kErrTimeout);
                return false;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_s_setprio(1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return true;
}

// groups a compressed word expands to
__device__ __forceinline__ u32 word_groups(u32 w) {
    return (w & kFillZero) ? (w & kCountMask) : 1u; // kernels.cu:298-304
}

// expand / checker / merge kernels: one workgroup of four wavefronts per 4096-word tile
constexpr int kExpandThreads = kExpandWaves * 64;                 // 256
constexpr int kExpandWordsPerThread = kScanTileWords / kExpandThreads; // 16
constexpr u32 kCoarse = kScanTileWords / 64;                      // coarse prefix: one entry per 64 words

// The census counts what was resident at one moment; near the edge that depends on how the dispatcher happened to
// place the wavefronts (measured: census 1184, runs above ~1060 workgroups lost a workgroup).  Keep a margin: only
// whole multiples of the CU count are used, i.e. what EVERY compute unit can hold.
[[maybe_unused]] inline int whole_per_cu(int resident) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 1;
    return resident >= cus ? (resident / cus) * cus : resident;
}

[[maybe_unused]] inline int persistent_grid(const void *kernel, int threads, u64 n_tiles) {
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    u64 g = (u64)cus * (u64)per_cu;
    if (g > n_tiles) g = n_tiles;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace
} // namespace wah

#endif // WAH_DEVICE_HPP_
