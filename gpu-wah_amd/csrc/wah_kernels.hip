// wah_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the WAH path.
//
// What the reference does in five kernels, two thrust scans and four blocking 8-byte D2H copies
// (compress.cu:129-166, decompress.cu:66-115, kernels.cu), is done here in
//   compress   : ONE persistent kernel            (reads 4N, writes 4C, nothing else)
//   decompress : streaming sums kernel + expand   (reads 4C twice, writes 4N')
// built on these CDNA4 idioms:
//   * a wavefront (64 lanes) owns a whole 1024-group segment; its 992 words are staged once in wave-private LDS
//     with 16-byte coalesced loads and re-read as 31-bit groups by a funnel shift (v_alignbit) -- the regroup of
//     kernels.cu:72-79 without idle lanes and without the shift-by-32;
//   * zero/ones classification produces 64-lane masks straight from v_cmp, "same as the next group" is one DPP
//     compare, so run detection, run lengths and the cross-warp merge (kernels.cu:126-229) collapse into a couple
//     of scalar mask operations per 64 groups plus one v_mbcnt rank per lane;
//   * run-end words are compacted in LDS and leave the chip as dense 256-byte stores;
//   * output offsets come from a one-hop "generation scan" over 4-byte {valid,count} granules written and polled
//     with agent-scope relaxed atomics (correct across the 8 non-coherent XCD L2s), instead of
//     thrust::exclusive_scan + moveData (compress.cu:133-166, kernels.cu:273-280);
//   * tiles are assigned round robin to the workgroups in arrival order, the grid is sized from a residency census
//     of the kernel itself, and every wait is bounded: a lost workgroup ends in WAH_ERR_TIMEOUT, never in a hang.
//
// Sections of this file, in order: wavefront helpers and the generation scan; compress (staging, the classify block
// -- csrc/classify_block*.inc --, compress_kernel); decompress (decode_sums_kernel, the expand routines,
// decode_expand_kernel); stream checker; generators / copy used by the bench; launchers; workspace clearing; the
// merge pass of wah_merge_fills_device.
#include "wah_internal.hpp"

#include "../../include/wah_gen.h"

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

// ===========================================================================
// compress
//
// Workgroup = W worker wavefronts + 1 scan wavefront, persistent, no barrier after start-up.
//   worker w : owns segment tile*W + w.  Per iteration g: the 4 x 16-byte loads of its segment were issued an
//              iteration earlier (software prefetch); it stages them in its private 4 KiB STAGE buffer, issues the
//              next tile's loads, classifies, compacts the run-end words in place, delivers their count, and turns
//              them into final WAH words (fill length = distance between consecutive run ends) written to its
//              private 4 KiB RING behind the words of earlier tiles that still wait for their output offset.
//              Whenever the offset of the oldest tile in the ring is known it is streamed out with dense 256-byte
//              stores (kernels.cu:256 + moveData).
//   scan wave: never touches bitmap data, so its memory queue only holds granule traffic.  Takes the tile's word
//              count from the last worker to deliver, resolves the tile's offset with the one-hop generation scan
//              above and hands it to the workers through LDS.
// Offsets are therefore needed two to four iterations after the counts were published (one for incompressible
// data, where a tile fills the ring), which absorbs the resolve latency and the jitter between 256 workgroups.
// ===========================================================================
constexpr u32 kStageWords = 1024; // staged segment (992 words + look-ahead) / compacted output words (<= 1024), aliased
constexpr u32 kOutWords = kStageWords + 4; // + one dump dword (non-end lanes), kept 16-byte aligned
constexpr u32 kPosEntries = 1032; // pos[0] = -1 sentinel, pos[k+1] = group position of run end k (u16)

typedef u32 u32x4 __attribute__((ext_vector_type(4)));

using lds_u32_ptr = __attribute__((address_space(3))) u32 *;
using lds_u64_ptr = __attribute__((address_space(3))) u64 *;
using lds_u16_ptr = __attribute__((address_space(3))) unsigned short *;
using lds_u8_ptr = __attribute__((address_space(3))) unsigned char *;

struct Prefetch {
    u32x4 v[4];
};

// Buffer descriptor over `bytes` bytes at p (raw, stride 0): loads past the end return 0, stores past the end are
// dropped -- the hardware does the tail padding (F5) and the capacity clipping, and addresses become
// descriptor + 32-bit lane offset + immediate, with no 64-bit vector arithmetic per access.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, u32 bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x27000);
}

// issue the four coalesced 16-byte loads of one segment (3968 B = 248 x 16 B; lanes 56..63 of the fourth load and
// everything past the end of the bitmap read as zero)
__device__ __forceinline__ void prefetch_segment(const CompressArgs &a, u32 seg, u32 lane, Prefetch &p) {
    // whole segments: 3968 bytes; the (one) partial segment at the end of the bitmap: what is left of it
    const u32 bytes = seg < a.full_segments ? kSegWords * 4u : a.tail_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.in + (u64)seg * kSegWords, bytes);
    const u32 off = lane * 16u;
    p.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    p.v[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 1024u, 0, 0);
    p.v[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 2048u, 0, 0);
    p.v[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 3072u, 0, 0);
}

// pair mode: the same four loads from the second bitmap
__device__ __forceinline__ void prefetch_segment2(const CompressArgs &a, u32 seg, u32 lane, Prefetch &p) {
    const u32 bytes = seg < a.full_segments ? kSegWords * 4u : a.tail_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.in2 + (u64)seg * kSegWords, bytes);
    const u32 off = lane * 16u;
    p.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    p.v[1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 1024u, 0, 0);
    p.v[2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 2048u, 0, 0);
    p.v[3] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 3072u, 0, 0);
}
// ... and the word-by-word combination (include/wah.h: WAH_OP_*); words behind the bitmap stay zero for every op
__device__ __forceinline__ void combine_pair(Prefetch &p, const Prefetch &q, u32 op) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        switch (op) {
        case 0: p.v[k] = p.v[k] & q.v[k]; break;
        case 1: p.v[k] = p.v[k] | q.v[k]; break;
        case 2: p.v[k] = p.v[k] ^ q.v[k]; break;
        default: p.v[k] = p.v[k] & ~q.v[k]; break;
        }
    }
}

// the fourth store also zeroes word 992, the look-ahead word of the last group (and the unused words up to 1023)
__device__ __forceinline__ void stage_prefetched(const Prefetch &p, u32 *lds, u32 lane) {
    u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
    dst[lane] = p.v[0];
    dst[lane + 64] = p.v[1];
    dst[lane + 128] = p.v[2];
    dst[lane + 192] = p.v[3];
}

// input that is only 4-byte aligned: bounds-checked scalar staging, zero padded (F5)
__device__ __forceinline__ void stage_slow(const CompressArgs &a, u32 seg, u32 *lds, u32 lane) {
    const u64 w0 = (u64)seg * kSegWords;
    const u64 left = a.n_words - w0;
    const u32 have = left < kSegWords ? (u32)left : kSegWords; // wave-uniform
    const u32 *src = a.in + w0;
    for (u32 i = lane; i < kSegWords + 64; i += 64)
        if (i <= kSegWords) lds[i] = i < have ? src[i] : 0u;
}

// Classify + run detect + compact one staged segment (wave-private LDS), returns the number of words produced.
//   classify  (kernels.cu:93-112): group = funnel shift of two staged words; zero / ones kinds by v_cmp, whose
//             result IS the 64-lane mask.
//   run ends  (kernels.cu:126-141 + the cross-warp merge of :188-229): a group does NOT end a run iff it is a
//             fill and the next group of the segment has the same value.  "Same as next" is one DPP compare
//             against the neighbouring lane (lane 63 is patched with lane 0 of the following step), so the
//             scalar side is three mask operations per 64 groups.  The group after the last one never matches,
//             so every segment closes its last run (tests.cpp:166-172).
//   compact   : step s-1 is finished once step s is classified; its run-end words go to LDS at rank = running
//             count + mbcnt, written over staged words that every later step has already left behind
//             (rank < 64 s <= 62 (s+1), the lowest word still to be read), with the group position beside it
//             (fill lengths are position differences, see the emit loop).
// kFull = all 1024 groups exist (every segment but possibly the last one of the bitmap).
// v_bcnt_u32_b32: acc + popcount(mask half).  Spelled out because the compiler would do a uniform popcount on the
// scalar unit and then needs a scalar add, a v_mov back and hazard nops around them: keeping the wave-uniform running
// count in a vector register makes the whole step a straight run of vector instructions (the scalar unit is shared
// by the four SIMDs of a CU and already carries the loop control and hand-off code).
__device__ __forceinline__ u32 add_popcount(u32 acc, u64 mask) {
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "s"((u32)mask));
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "s"((u32)(mask >> 32)));
    return acc;
}

template <bool kFull>
__device__ __forceinline__ u32 classify_compact(const u32 *sp, u32 *lds, unsigned short *pos, u32 r, u32 lane_v,
                                                u32 nvalid, bool long_fills, bool &any_fill) {
    // phase 1: all 16 LDS reads, then the funnel shifts: every staged word is in registers before the first
    // compacted word overwrites the staging buffer
    u32 x[kSteps + 1];
#pragma unroll
    for (int s = 0; s < (int)kSteps; ++s) {
        const u32 lo = sp[62 * s];
        const u32 hi = sp[62 * s + 1];
        x[s] = __builtin_amdgcn_alignbit(hi, lo, r) & kOnes31;
    }
    x[kSteps] = 0xFFFFFFFFu; // "group after the last one": a value no 31-bit group can equal
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // phase 2: one straight-line block of vector instructions per step.
    //   next   : value of the following group = lane l+1 (DPP wave_shl:1); lane 63 has no source lane and keeps the
    //            `old` operand, which a wave_rol:1 of the NEXT step's register has loaded with that step's lane 0
    //   z      : (x ^ next) | ((x + 1) & 0x7FFFFFFE) is zero  <=>  x is 0 or 0x7FFFFFFF AND the next group equals it
    //            <=>  the group does NOT end a run (kernels.cu:93-141 and the merge of :188-229 in three operations)
    //   ends   : v_cmp_ne z, 0 -- the 64-lane mask comes out of the compare itself
    //   rank   : v_mbcnt pair seeded with the running count; count += v_bcnt pair
    //   write  : every lane stores; lanes that end no run store to a dump slot (cheaper than masking EXEC, which
    //            is scalar work)
    u32 count_v = 0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(count_v)); // a VECTOR zero: keeps the running count off the scalar unit
    u32 min_t = 0xFFFFFFFFu;
    // LDS byte addresses: value k at vbase + 4 k, its position at pbase + 2 k; k = kStageWords is the dump slot
    const u32 vbase = (u32)(uintptr_t)(lds_u32_ptr)lds;
    const u32 pbase = (u32)(uintptr_t)(lds_u16_ptr)pos + 2u;
    u32 dump_slot;
    asm volatile("v_mov_b32 %0, 0x400" : "=v"(dump_slot)); // kStageWords, in a vector register (v_cndmask cannot take a literal)
    static_assert(kStageWords == 0x400, "dump slot literal");
    if (kFull) {
        // Hand-scheduled block for the 16 steps (csrc/classify_block.inc, generated by tools/gen_classify_block.py):
        // 13.5 vector + 2 LDS instructions per step, software-pipelined by one step so that no hazard needs a wait
        // state (a DPP source or a v_cmp mask read as data must be two instructions old; the compiler pads with
        // s_nop, also around every asm statement).  The kernel is bound by vector issue, so
        // every instruction here is ~0.25 % of its run time.
        u32 na, ta, nb, tb, ps;
        const u32 lane2 = lane_v * 0x10001u; // the lane id in both halves: position words are built two at a time
        // two schedules of the same block: `long_fills` (the wave's previous segment compressed to a few words) takes
        // the one in which a step without any run end branches over the ranking and the stores
#define WAH_CLASSIFY_OPERANDS                                                                                                  \
    : [na] "=&v"(na), [ta] "=&v"(ta), [nb] "=&v"(nb), [tb] "=&v"(tb), [ps] "=&v"(ps), [cn] "+&v"(count_v)                      \
    : [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]), [x8] "v"(x[8]), [x9] "v"(x[9]), [x10] "v"(x[10]), [x11] "v"(x[11]), [x12] "v"(x[12]), [x13] "v"(x[13]), [x14] "v"(x[14]), [x15] "v"(x[15]), [x16] "v"(x[16]),                                                                                                                  \
      [ln2] "v"(lane2), [vb] "s"(vbase), [pb] "s"(pbase), [dm] "v"(dump_slot)                                                  \
    : "vcc", "memory"
        if (long_fills) {
            asm volatile(
#include "classify_block_skip.inc"
                WAH_CLASSIFY_OPERANDS);
        } else {
            asm volatile(
#include "classify_block.inc"
                WAH_CLASSIFY_OPERANDS);
        }
#undef WAH_CLASSIFY_OPERANDS
        const u32 count = uniform32(count_v);
        // Some emitted word is a fill iff some group is one.  Fewer words than groups: certainly.  As many words as
        // groups (incompressible data): only fills of length 1 are possible, look for an all-zero / all-one group.
        any_fill = true;
        if (count == kSegGroups) {
            u32 lo = x[0], hi = x[0];
#pragma unroll
            for (int s = 1; s < (int)kSteps; s += 2) {
                lo = s + 1 < (int)kSteps ? min(lo, min(x[s], x[s + 1])) : min(lo, x[s]);
                hi = s + 1 < (int)kSteps ? max(hi, max(x[s], x[s + 1])) : max(hi, x[s]);
            }
            any_fill = __ballot(lo == 0u || hi == kOnes31) != 0;
        }
        return count;
    }
#pragma unroll
    for (int s = 0; s < (int)kSteps; ++s) {
        const u32 carry = (u32)__builtin_amdgcn_mov_dpp((int)x[s + 1], 0x134 /* wave_rol:1 */, 0xf, 0xf, true);
        const u32 nxt = __builtin_amdgcn_update_dpp(carry, x[s], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
        const u32 t = (x[s] + 1u) & 0x7FFFFFFEu; // zero <=> x is all zeros or all ones
        const u32 z = __builtin_amdgcn_bitop3_b32(x[s], nxt, t, 0xbe); // (x ^ next) | t
        u64 e = __ballot(z != 0u);
        const int rem = (int)nvalid - 64 * s;
        const u64 valid = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
        const u64 last = (rem >= 1 && rem <= 64) ? (1ull << (rem - 1)) : 0ull; // the last existing group closes its run
        e = (e | last) & valid;
        if (rem > 0) min_t = min(min_t, (lane_v < (u32)rem) ? t : 0xFFFFFFFFu);
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(e >> 32), __builtin_amdgcn_mbcnt_lo((u32)e, count_v));
        const u32 slot = __builtin_amdgcn_inverse_ballot_w64(e) ? rank : kStageWords;
        *(lds_u32_ptr)(uintptr_t)(vbase + (slot << 2)) = x[s];
        *(lds_u16_ptr)(uintptr_t)(pbase + (slot << 1)) = (unsigned short)(64 * s + (int)lane_v);
        count_v = add_popcount(count_v, e);
    }
    any_fill = __ballot(min_t == 0u) != 0; // some group is a fill, so some emitted word is one
    return uniform32(count_v);
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

constexpr u32 kDepth = 8;      // generations a workgroup keeps bookkeeping for (power of two)
constexpr u32 kMaxPending = 4; // finished tiles a worker may hold in LDS while their offsets resolve (< kDepth - 2)

template <int W, bool kPair = false>
__global__ __launch_bounds__((W + 1) * 64) void compress_kernel(const CompressArgs a) {
    __shared__ __attribute__((aligned(16))) u32 s_out[2][W][kOutWords];
    __shared__ unsigned short s_pos[W][kPosEntries];
    __shared__ u32 s_count[kDepth][W];    // words per worker of tile (gen % kDepth)
    __shared__ u32 s_prefix[kDepth][W];   // ... and the words of the workers before it
    __shared__ u32 s_arrived[kDepth];     // workers that have delivered their count for tile (gen % kDepth)
    __shared__ u32 s_total[kDepth];       // words of tile (gen % kDepth) ...
    __shared__ u32 s_total_flag[kDepth];  // ... valid when == gen + 1
    __shared__ u64 s_base[kDepth];        // output offset of tile (gen % kDepth) ...
    __shared__ u32 s_base_flag[kDepth];   // ... valid when == gen + 1
    __shared__ u32 s_arrival;

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const bool worker = wave < (u32)W;

    if (threadIdx.x < kDepth) {
        s_arrived[threadIdx.x] = 0;
        s_total_flag[threadIdx.x] = 0;
        s_base_flag[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) s_arrival = draw_arrival(a.ctrl);
    __syncthreads();
    const u32 arrival = uniform32(s_arrival);

    if (a.census) {
        // residency census: how many workgroups of this kernel are running together?  Everybody that is resident
        // arrives within about a microsecond; whoever is not cannot start before a resident one exits.
        // EVERY wave stays for the whole census (the barrier below): a wave that left early would give back its
        // slot and registers, and more workgroups would fit than in the real run.
        if (threadIdx.x == 0) {
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < 3000) __builtin_amdgcn_s_sleep(8); // 30 us (100 MHz)
            if (arrival == 0)
                a.ctrl[kCtlCensus] = __hip_atomic_load(a.ctrl + kCtlStart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        return;
    }

    const u32 stride = gridDim.x;
    const u32 row_stride = (stride + 3u) & ~3u;
    WAH_STAMP_DECL

    if (worker)
        __builtin_amdgcn_s_setprio(1);
    else
        __builtin_amdgcn_s_setprio(2);
    if (!worker) {
        // ---------------- scan wave: resolve output offsets, tile after tile, as the counts come in ------------
        GenScan scan = {0, 0, 0};
        u32 gen = 0;
        for (u32 tile = arrival; tile < a.n_tiles; tile += stride, ++gen) {
            const u32 q = gen & (kDepth - 1u);
            if (!lds_wait(&s_total_flag[q], gen + 1u, a.ctrl, lane)) break;
            const u32 aggregate = uniform32(lds_ld(&s_total[q]));
            WAH_STAMP(0);
            const u64 excl = resolve_generation(a.gen_desc, gen, arrival, stride, row_stride, aggregate, scan, lane, a.ctrl);
            WAH_STAMP(1);
#ifdef WAH_DIAG
            if (lane == 0 && a.seg_offsets) {
                a.seg_offsets[(u64)tile * 4 + 2] = __builtin_amdgcn_s_memrealtime();
                a.seg_offsets[(u64)tile * 4 + 3] = ((u64)blockIdx.x << 32) | gen;
            }
#endif
            if (lane == 0) {
                __hip_atomic_store((lds_u64_ptr)&s_base[q], excl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lds_publish(&s_base_flag[q], gen + 1u);
                if (tile == a.n_tiles - 1) {
                    *a.out_words = excl + aggregate;
#ifndef WAH_DIAG
                    if (a.seg_offsets) a.seg_offsets[a.n_segments] = excl + aggregate;
#endif
                }
                if (excl + aggregate > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
            }
#ifdef WAH_DIAG
            dg_acc[7] += 1;
#endif
        }
#ifdef WAH_DIAG
        if (lane == 0)
            for (int i = 0; i < 2; ++i)
                atomicAdd(reinterpret_cast<unsigned long long *>(a.ctrl + 192) + 8 + i, (unsigned long long)dg_acc[i]);
#endif
        return;
    }

    // ---------------- worker waves -------------------------------------------------------------------------
    // Each worker owns two 4 KiB LDS buffers.  One is the STAGE: the segment is staged, classified and compacted
    // there.  The other is a RING of finished output words that wait for their global offset: up to kMaxPending
    // tiles (as many as fit 1024 words), oldest first.  So the offset of a tile is not needed one iteration after
    // it was published (the resolve latency across the chip is about one iteration of work, measured) but only
    // when the ring runs out of room -- two or three iterations later for compressible data.  A segment that does
    // not fit beside what is pending (incompressible data) waits for the ring to drain and then the two buffers
    // simply swap roles, without copying.
    // regroup constants: group g = 64*step + lane starts at stream bit 31*g; 64 groups = 1984 bits = 62 words
    // exactly, so the in-word shift is fixed per lane and the word index advances by 62 per step
    const u32 r = (31u * lane) & 31u;
    unsigned short *const pos = s_pos[wave];

    Prefetch pre, pre2; // pre2: pair mode only (wah_bitop_device), the second bitmap's words
    pre.v[0] = pre.v[1] = pre.v[2] = pre.v[3] = u32x4{0, 0, 0, 0};
    pre2 = pre;
    // wave-uniform: `pre` holds the current tile's segment (always, unless the input is only 4-byte aligned)
    bool pre_valid = false;
    {
        const u32 seg = arrival * W + wave;
        if (arrival < a.n_tiles && seg < a.n_segments && a.fast_segments) {
            prefetch_segment(a, seg, lane, pre);
            if (kPair) prefetch_segment2(a, seg, lane, pre2);
            pre_valid = true;
        }
    }

    u32 *stage = s_out[0][wave];
    u32 *ring = s_out[1][wave];
    u32 pend = 0;                       // tiles in the ring: generations gen - pend .. gen - 1
    u32 ring_head = 0;                  // ring index of the oldest pending word
    u32 used = 0;                       // pending words
    u32 pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0; // their word counts, oldest first
    bool ok = true;

    // stream out the oldest pending tile (kernels.cu:256 + moveData, kernels.cu:273-280); `block`: wait for its offset
    auto emit_oldest = [&](u32 gen_now, bool block, u32 lane_v) -> bool {
        const u32 pgen = gen_now - pend;
        const u32 q = pgen & (kDepth - 1u);
        if (lds_ld(&s_base_flag[q]) != pgen + 1u) {
            if (!block) return false;
            if (!lds_wait(&s_base_flag[q], pgen + 1u, a.ctrl, lane)) {
                ok = false;
                return false;
            }
        }
        WAH_STAMP(3);
        const u32 pseg = (arrival + pgen * stride) * W + wave;
        const u32 cnt = pc0;
        if (pseg < a.n_segments) {
            const u64 base = uniform64(lds_ld64(&s_base[q])) + uniform32(lds_ld(&s_prefix[q][wave]));
#ifndef WAH_DIAG
            if (lane == 0 && a.seg_offsets) a.seg_offsets[pseg] = base;
#endif
            if (base < a.out_capacity && cnt != 0u) {
                // descriptor over this segment's slice of the output (clipped to the capacity: words past it
                // are dropped by the hardware, and the scan wave has already raised the capacity error)
                const u64 room = a.out_capacity - base;
                const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + base, (room < cnt ? (u32)room : cnt) * 4u);
                const u32 off = lane_v * 4u;
                if (ring_head + ((cnt + 255u) & ~255u) <= kStageWords) {
                    // no wrap inside the trips (reads behind the last word stay inside the buffer): plain addressing
                    const u32 *const r0 = ring + ring_head + lane_v;
                    for (u32 t = 0; t < cnt; t += 256u) { // four LDS reads in flight, then four dense stores
                        u32 v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = r0[t + 64u * k];
#pragma unroll
                        for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b32(v[k], rsrc, off + 256u * k, t * 4u, 0);
                    }
                } else {
                    for (u32 t = 0; t < cnt; t += 256u) {
                        u32 v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = ring[(ring_head + t + lane_v + 64u * k) & (kStageWords - 1u)];
#pragma unroll
                        for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b32(v[k], rsrc, off + 256u * k, t * 4u, 0);
                    }
                }
            }
        }
        ring_head = (ring_head + cnt) & (kStageWords - 1u);
        used -= cnt;
        pc0 = pc1;
        pc1 = pc2;
        pc2 = pc3;
        pc3 = 0;
        --pend;
        // later iterations overwrite these words: order the reads before those writes
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        WAH_STAMP(4);
        return true;
    };

    // the last segment compressed to a handful of words: the next one probably consists of long fills too
    constexpr u32 kLongFillsBelow = 256;
    bool long_fills = false, whole_run = false;
    u32 gen = 0;
    for (u32 tile = arrival; tile < a.n_tiles && ok; tile += stride, ++gen) {
        const u32 seg = tile * W + wave;
        u32 count = 0;
        // Opaque copy of the lane id, renewed every iteration: per-step constants derived from it (group
        // positions, LDS addresses) are then recomputed next to their use instead of being hoisted out of the
        // persistent loop, where 16 + 16 of them would be kept live and spilled.
        u32 lane_v = lane;
        asm volatile("" : "+v"(lane_v));
#ifdef WAH_DIAG
        if (threadIdx.x == 0 && a.seg_offsets) a.seg_offsets[(u64)tile * 4 + 0] = __builtin_amdgcn_s_memrealtime();
#endif
        const bool has_seg = seg < a.n_segments;
        if (kPair && pre_valid) combine_pair(pre, pre2, a.op); // from here on `pre` is the combined bitmap
        // Inside a very long run (the wave's last segment was one or two words) the whole segment is probably one
        // fill: decide that from the prefetched registers -- all 992 words zero, or all ones -- and skip staging and
        // classification.  (Lanes 56..63 of the fourth load lie behind the segment and read as zero.)
        u32 uniform_kind = 0; // 1: all zero, 2: all ones
        if (has_seg && pre_valid && whole_run && seg + 1u < a.n_segments) {
            const u32x4 o = pre.v[0] | pre.v[1] | pre.v[2] | pre.v[3];
            const u32x4 tail_fix = lane >= 56u ? u32x4{~0u, ~0u, ~0u, ~0u} : u32x4{0, 0, 0, 0};
            const u32x4 n = pre.v[0] & pre.v[1] & pre.v[2] & (pre.v[3] | tail_fix);
            if (__ballot((o.x | o.y | o.z | o.w) != 0u) == 0)
                uniform_kind = 1;
            else if (__ballot((n.x & n.y & n.z & n.w) != ~0u) == 0)
                uniform_kind = 2;
        }
        if (has_seg && !uniform_kind) {
            if (pre_valid)
                stage_prefetched(pre, stage, lane);
            else
                stage_slow(a, seg, stage, lane);
        }
        // the wave re-reads other lanes' words: order the LDS traffic at wavefront scope (no barrier needed)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        WAH_STAMP(0);

        // software prefetch of the next tile's segment: in flight during everything below
        {
            const u32 next_tile = tile + stride;
            const u32 nseg = next_tile * W + wave;
            pre_valid = next_tile < a.n_tiles && nseg < a.n_segments && a.fast_segments;
            if (pre_valid) {
                prefetch_segment(a, nseg, lane, pre);
                if (kPair) prefetch_segment2(a, nseg, lane, pre2);
            }
        }

        bool any_fill = false;
        if (uniform_kind) {
            // one run end, at the last group: what classify_compact would have left in the stage buffer
            if (lane == 0) {
                stage[0] = uniform_kind == 1 ? 0u : kOnes31;
                pos[0] = 0xFFFFu;
                pos[1] = (unsigned short)(kSegGroups - 1u);
            }
            count = 1;
            any_fill = true;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else if (has_seg) {
            const u32 nvalid = (seg == a.n_segments - 1) ? a.last_segment_groups : kSegGroups;
            if (lane == 0) pos[0] = 0xFFFFu; // position "-1": the run before the first one ends there
            const u32 *sp = stage + ((31u * lane_v) >> 5);
            count = nvalid == kSegGroups ? classify_compact<true>(sp, stage, pos, r, lane_v, nvalid, long_fills, any_fill)
                                         : classify_compact<false>(sp, stage, pos, r, lane_v, nvalid, long_fills, any_fill);
            long_fills = count < kLongFillsBelow;
            whole_run = count <= 2u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        WAH_STAMP(1);

        // deliver the count (nothing else of this tile is needed to resolve offsets); the last worker to arrive
        // publishes the tile's total to the other workgroups (one 4-byte granule, see resolve_generation) and to
        // the scan wave
        {
            const u32 q = gen & (kDepth - 1u);
            u32 last = 0;
            if (lane == 0) {
                lds_st(&s_count[q][wave], count);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                last = __hip_atomic_fetch_add((lds_u32_ptr)&s_arrived[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == (u32)W - 1u;
            }
            if (uniform32(last)) {
                // lane w: words of worker w -> DPP scan -> every worker's offset inside the tile, and the total
                const u32 mine = lane < (u32)W ? lds_ld(&s_count[q][lane]) : 0u;
                const u32 incl = wave_scan_incl32(mine);
                if (lane < (u32)W) lds_st(&s_prefix[q][lane], incl - mine);
                const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
                if (lane == 0) {
#ifdef WAH_DIAG
                    if (a.seg_offsets) a.seg_offsets[(u64)tile * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
                    publish_generation(a.gen_desc, gen, arrival, row_stride, total);
                    lds_st(&s_arrived[q], 0u);
                    lds_st(&s_total[q], total);
                    lds_publish(&s_total_flag[q], gen + 1u);
                }
            }
        }
        WAH_STAMP(2);

        // room in the ring for this tile's words (every wait here is for an offset published >= 1 iteration ago)
        while (ok && pend != 0u && (pend == kMaxPending || used + count > kStageWords)) (void)emit_oldest(gen, true, lane_v);
        if (!ok) break;
        const bool in_place = pend == 0u; // ring empty: the stage buffer BECOMES the ring, nothing is copied
        if (in_place) {
            u32 *const t = stage;
            stage = ring;
            ring = t;
            ring_head = 0;
        }
        if (any_fill || !in_place) {
            // final words (kernels.cu:244-249): fill length = distance between consecutive run ends; written to the
            // ring behind what is pending (or in place).  Four batches (256 words) per trip: 12 LDS reads in flight,
            // then the arithmetic, then 4 writes; every lane rewrites its word (unchanged if a literal).
            const u32 *const src = in_place ? ring : stage;
            const u32 tail = (ring_head + used) & (kStageWords - 1u);
            const u32 padded = (count + 63u) & ~63u; // whole 64-word batches
            // one word: fill -> type | length, literal -> itself
            auto final_word = [](u32 v, u32 p1, u32 p0) {
                const u32 len = (p1 - p0) & 0xFFFFu;
                return v - 1u >= 0x7FFFFFFEu ? ((v ? kFillOne : kFillZero) | len) : v;
            };
            if (tail + padded <= kStageWords && used + padded <= kStageWords) {
                // usual case: the batches neither wrap around the ring nor reach the oldest pending words, so whole
                // batches are written (the up to 63 words behind the last real one land on free ring space): no
                // predicates, no wrap arithmetic, every address is one register + an immediate
                const u32 *const s0 = src + lane_v;
                const unsigned short *const q0 = pos + lane_v;
                u32 *const d0 = ring + tail + lane_v;
                for (u32 t = 0; t < padded; t += 256u) {
                    const u32 left = padded - t; // 64, 128, 192 or >= 256
                    u32 v[4], p1[4], p0[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (64u * k < left) {
                            v[k] = s0[t + 64u * k];
                            p1[k] = q0[t + 64u * k + 1u];
                            p0[k] = q0[t + 64u * k];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (64u * k < left) d0[t + 64u * k] = final_word(v[k], p1[k], p0[k]);
                }
            } else {
                for (u32 t = 0; t < count; t += 256u) {
                    u32 v[4], p1[4], p0[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32 j = t + lane_v + 64u * k;
                        v[k] = src[j];
                        p1[k] = pos[j + 1u];
                        p0[k] = pos[j];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32 j = t + lane_v + 64u * k;
                        if (j < count) ring[(tail + j) & (kStageWords - 1u)] = final_word(v[k], p1[k], p0[k]);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        // push
        if (pend == 0u) pc0 = count;
        else if (pend == 1u) pc1 = count;
        else if (pend == 2u) pc2 = count;
        else pc3 = count;
        ++pend;
        used += count;
        WAH_STAMP(5);

        // stream out whatever has its offset already
        while (ok && pend != 0u && emit_oldest(gen + 1u, false, lane_v)) {}
#ifdef WAH_DIAG
        dg_acc[7] += 1;
#endif
    }
    while (ok && pend != 0u) { // drain
        u32 lane_v = lane;
        asm volatile("" : "+v"(lane_v));
        (void)emit_oldest(gen, true, lane_v);
    }
    WAH_STAMP_FLUSH(a.ctrl);
#ifdef WAH_DIAG
    if (lane == 0 && a.seg_offsets && blockIdx.x < 64) { // per-wave phase totals of the first 64 workgroups
        for (int i = 0; i < 8; ++i)
            a.seg_offsets[(u64)a.n_tiles * 4 + (u64)gridDim.x * 10 + ((u64)blockIdx.x * 16 + wave) * 8 + i] = dg_acc[i];
    }
    if (threadIdx.x == 0 && a.seg_offsets) {
        u32 xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        for (int i = 0; i < 8; ++i) a.seg_offsets[(u64)a.n_tiles * 4 + (u64)blockIdx.x * 10 + i] = dg_acc[i];
        a.seg_offsets[(u64)a.n_tiles * 4 + (u64)blockIdx.x * 10 + 8] = xcc;
        a.seg_offsets[(u64)a.n_tiles * 4 + (u64)blockIdx.x * 10 + 9] = arrival;
    }
#endif
}

// ===========================================================================
// decompress
//
// The reference runs getCounts -> thrust::exclusive_scan over one u64 PER COMPRESSED WORD -> decompressWords (a
// serial fill loop per thread into a 4-byte-per-group intermediate) -> mergeWords (kernels.cu:291-385,
// decompress.cu:66-115).  Here:
//   pass 1  decode_sums_kernel   : streaming reduce.  Tiles of 4096 compressed words; per tile the number of 31-bit
//                                   groups it expands to, turned into exclusive tile bases by the same one-hop
//                                   generation scan as compress.  Reads C once, writes 8 bytes per tile.
//   pass 2  decode_expand_kernel : one workgroup per tile, tile words resident in LDS.  A tile OWNS the output
//                                   segments (1024 groups -> 992 words) whose first group falls into it; each of its
//                                   wavefronts expands whole segments: group -> source word by RANK (mbcnt over a
//                                   1024-bit mask of word starts), fill / literal decode, 31 -> 32 repack in
//                                   registers with two DPP shifts, dense 248-byte stores.
// Any stream the reference decoder accepts is handled (arbitrary 30-bit counts, fills across segment boundaries).
// ===========================================================================
__device__ __forceinline__ u32 word_groups(u32 w) {
    return (w & kFillZero) ? (w & kCountMask) : 1u; // kernels.cu:298-304
}

constexpr u32 kGenEscape = 0x7FFFFFFFu; // granule value: "total does not fit 31 bits, read the 64-bit side entry"

// generation scan with 64-bit totals (a tile of fills can expand to more than 2^31 groups)
__device__ __forceinline__ u64 resolve_generation64(const u32 *gdesc, const u64 *big, u32 gen, u32 slot, u32 G,
                                                    u32 row_stride, u64 aggregate, GenScan &st, u64 &own_prev64,
                                                    u64 &below_prev64, u32 lane, u32 *ctrl) {
    const u32 *cur = gdesc + (u64)gen * row_stride;
    const u32 *prv = cur - row_stride;
    bool need_prev = gen > 0 && slot + 1 < G, need_cur = slot > 0;
    u64 above = 0, below = 0;
    u32 spins = 0;
    while (need_prev || need_cur) {
        u64 sum_cur = 0, sum_prev = 0;
        bool bad_cur = false, bad_prev = false;
        for (u32 k = lane; k < G; k += 64u) {
            if (need_cur && k < slot) {
                const u32 e = __hip_atomic_load(cur + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad_cur |= !(e & kGenValid);
                u64 v = e & ~kGenValid;
                if (v == kGenEscape && (e & kGenValid))
                    v = __hip_atomic_load(big + ((u64)gen * G + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sum_cur += v;
            }
            if (need_prev && k > slot) {
                const u32 e = __hip_atomic_load(prv + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad_prev |= !(e & kGenValid);
                u64 v = e & ~kGenValid;
                if (v == kGenEscape && (e & kGenValid))
                    v = __hip_atomic_load(big + ((u64)(gen - 1) * G + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sum_prev += v;
            }
        }
        bool progressed = false;
        if (need_cur && !__any(bad_cur)) {
            below = uniform64(wave_sum(sum_cur));
            need_cur = false;
            progressed = true;
        }
        if (need_prev && !__any(bad_prev)) {
            above = uniform64(wave_sum(sum_prev));
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
    if (gen > 0) st.gen_base += below_prev64 + own_prev64 + above;
    below_prev64 = below;
    own_prev64 = aggregate;
    return st.gen_base + below;
}

// Workgroup = kSumWorkers worker wavefronts + 1 scan wave.  A worker sums one whole expand tile (4096 words) per
// iteration in four rolling 4 KiB rounds; the workgroup's tile (kSumWorkers expand tiles, 128 KiB) costs ONE
// granule, so the scan traffic stays below 1 % of the stream.
constexpr int kSumWorkers = kSumTilesPerGroup;

__global__ __launch_bounds__((kSumWorkers + 1) * 64) void decode_sums_kernel(const ScanArgs a) {
    __shared__ u64 s_part[4][kSumWorkers];
    __shared__ u32 s_arrived[4];
    __shared__ u64 s_total[4];
    __shared__ u32 s_total_flag[4];
    __shared__ u32 s_scanned; // tiles the scan wave has consumed (flow control of the 4-deep hand-off ring)
    __shared__ u32 s_arrival;

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const bool worker = wave < (u32)kSumWorkers;

    if (threadIdx.x < 4) {
        s_arrived[threadIdx.x] = 0;
        s_total_flag[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) {
        s_scanned = 0;
        s_arrival = draw_arrival(a.ctrl);
    }
    __syncthreads();
    const u32 arrival = uniform32(s_arrival);
    if (a.census) {
        if (threadIdx.x == 0) { // residency census, see compress_kernel
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < 3000) __builtin_amdgcn_s_sleep(8);
            if (arrival == 0)
                a.ctrl[kCtlCensus] = __hip_atomic_load(a.ctrl + kCtlStart, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        return;
    }
    const u32 stride = gridDim.x;
    const u32 row_stride = (stride + 3u) & ~3u;
    const u32 n_tiles = (u32)a.n_tiles;                                      // expand tiles (4096 words)
    const u32 n_wg_tiles = (n_tiles + (u32)kSumWorkers - 1u) / (u32)kSumWorkers; // workgroup tiles

    if (!worker) {
        // scan wave: workgroup-tile totals -> exclusive bases; then the bases of the expand tiles inside
        GenScan scan = {0, 0, 0};
        u64 own_prev = 0, below_prev = 0;
        u32 gen = 0;
        for (u32 wt = arrival; wt < n_wg_tiles; wt += stride, ++gen) {
            const u32 q = gen & 3u;
            if (!lds_wait(&s_total_flag[q], gen + 1u, a.ctrl, lane)) break;
            const u64 total = uniform64(lds_ld64(&s_total[q]));
            const u64 part = lane < (u32)kSumWorkers ? lds_ld64(&s_part[q][lane]) : 0ull;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_st(&s_scanned, gen + 1u); // the ring slot may be reused
            const u64 excl = resolve_generation64(a.gen_desc, a.big, gen, arrival, stride, row_stride, total, scan, own_prev,
                                                  below_prev, lane, a.ctrl);
            // lane w: groups in front of expand tile wt * kSumWorkers + w
            const u64 incl_part = wave_scan_incl(part, lane);
            const u32 et = wt * (u32)kSumWorkers + lane;
            if (lane < (u32)kSumWorkers && et < n_tiles) a.tile_base[et] = excl + (incl_part - part);
            if (lane == 0 && wt == n_wg_tiles - 1) {
                const u64 groups = excl + total;
                a.tile_base[n_tiles] = groups;
                a.info[1] = groups;
                a.info[0] = (31ull * groups + 31ull) / 32ull; // decompress.cu:84-93
            }
        }
        return;
    }

    // worker waves: stream one expand tile per iteration, sum the group counts (getCounts, kernels.cu:291-309)
    uint4 pre[4];
    // A fill word of count 0 expands to nothing; the reference decoder steps over it (kernels.cu:332-354).  The
    // expand kernel's rank arithmetic assumes that every word owns at least one group, so every tile is checked here
    // and expand takes its index-map route for the tiles concerned (tile_flags).  `pre_whole`: the prefetched round
    // consists of real words only (no padding past the end, which is written as empty fills); `pre_empty`: a round
    // that needed bounds checks contains an empty fill among its real words.
    bool pre_empty = false, pre_whole = true;
    // round `rd` (0..3) of expand tile `et`: 1024 words as four fully coalesced 1 KiB loads (order is irrelevant)
    auto load_round = [&](u32 et, u32 rd) {
        const u64 w0 = (u64)et * kScanTileWords + (u64)rd * 1024u;
        if (a.aligned16 && w0 + 1024u <= a.c_words) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + w0);
#pragma unroll
            for (int k = 0; k < 4; ++k) pre[k] = src[k * 64 + (int)lane];
            pre_whole = true;
            pre_empty = false;
        } else {
            u32 t[16];
            pre_whole = false;
            pre_empty = false;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const u64 i = w0 + (u64)(k / 4) * 256u + (u64)lane * 4u + (u64)(k % 4);
                t[k] = i < a.c_words ? a.comp[i] : 0x80000000u; // past the end: a fill of zero groups
                pre_empty |= i < a.c_words && word_groups(t[k]) == 0u;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) pre[k] = make_uint4(t[4 * k], t[4 * k + 1], t[4 * k + 2], t[4 * k + 3]);
        }
    };
    if (arrival < n_wg_tiles) load_round(arrival * (u32)kSumWorkers + wave, 0);
    u32 gen = 0;
    for (u32 wt = arrival; wt < n_wg_tiles; wt += stride, ++gen) {
        const u32 et = wt * (u32)kSumWorkers + wave;
        u64 mine = 0;
        bool tile_empty = false;
#pragma unroll
        for (u32 rd = 0; rd < 4; ++rd) {
            uint4 cur[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = pre[k];
            const bool cur_whole = pre_whole;
            tile_empty |= pre_empty;
            // rolling prefetch: next round of this tile, or round 0 of this wave's next tile
            if (rd < 3)
                load_round(et, rd + 1);
            else if (wt + stride < n_wg_tiles)
                load_round((wt + stride) * (u32)kSumWorkers + wave, 0);
            u32 round_min = 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) { // four counts of < 2^30 each fit 32 bits; words past the end are empty fills
                const u32 nx = word_groups(cur[k].x), ny = word_groups(cur[k].y), nz = word_groups(cur[k].z), nw = word_groups(cur[k].w);
                mine += (u64)(nx + ny + nz + nw);
                round_min = min(min(round_min, min(nx, ny)), min(nz, nw));
            }
            if (cur_whole) tile_empty |= round_min == 0u;
        }
        {
            const bool any_empty = __any(tile_empty);
            if (lane == 0 && (u64)et < a.n_tiles) a.tile_flags[et] = any_empty ? 1 : 0;
        }
        const u64 wave_total = uniform64(wave_sum(mine));
        const u32 q = gen & 3u;
        u32 last = 0;
        if (lane == 0) {
            // the ring slot is free once the scan wave has consumed the tile that used it 4 generations ago
            if (gen >= 4) lds_wait_reached(&s_scanned, gen - 3u, a.ctrl);
            __hip_atomic_store((lds_u64_ptr)&s_part[q][wave], wave_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            last = __hip_atomic_fetch_add((lds_u32_ptr)&s_arrived[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ==
                   (u32)kSumWorkers - 1u;
        }
        if (uniform32(last)) {
            if (lane == 0) {
                u64 total = 0;
#pragma unroll
                for (int w = 0; w < kSumWorkers; ++w) total += lds_ld64(&s_part[q][w]);
                // publish: one 4-byte granule; totals of 2^31 - 1 groups or more go through the 64-bit side entry
                if (total >= kGenEscape) {
                    __hip_atomic_store(a.big + ((u64)gen * stride + arrival), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    publish_generation(a.gen_desc, gen, arrival, row_stride, kGenEscape);
                } else {
                    publish_generation(a.gen_desc, gen, arrival, row_stride, (u32)total);
                }
                lds_st(&s_arrived[q], 0u);
                __hip_atomic_store((lds_u64_ptr)&s_total[q], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                lds_publish(&s_total_flag[q], gen + 1u);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// pass 2
// ---------------------------------------------------------------------------
constexpr int kExpandThreads = kExpandWaves * 64;                 // 256
constexpr int kExpandWordsPerThread = kScanTileWords / kExpandThreads; // 16
constexpr u32 kCoarse = kScanTileWords / 64;                      // coarse prefix: one entry per 64 words

// word `idx` (tile-local index) of the stream: LDS inside the tile, global memory past its end
__device__ __forceinline__ u32 tile_word(const u32 *s_words, const ExpandArgs &a, u64 tile_w0, u32 idx) {
    if (idx < (u32)kScanTileWords) return s_words[idx];
    const u64 g = tile_w0 + idx;
    return g < a.c_words ? a.comp[g] : 0u;
}

// Flags of one output segment: the byte for group p lives at (p % 64) * 16 + p / 64, so that ONE 16-byte LDS read
// hands a lane the flags of its group in all 16 steps.  Behind the 1024 flags a dump area takes the stores of lanes
// that have nothing to flag (cheaper than masking them off).
constexpr u32 kFlagBytes = kSegGroups + 64;
__device__ __forceinline__ u32 flag_slot(u32 p) { return ((p & 63u) << 4) | (p >> 6); }

// The 16 steps of one output segment: 64 groups -> 62 output words each.  kWhole: all 1024 groups exist and the
// whole segment lies inside the output; kLocal: every source word is inside the LDS-resident tile.  (Template
// parameters so that the step body is straight-line code.)
template <bool kWhole, bool kLocal>
__device__ __forceinline__ void expand_steps(const ExpandArgs &a, const u32 *s_words, const unsigned char *flag, u64 tile_w0,
                                             __amdgpu_buffer_rsrc_t rsrc, u32 first_word, u32 nvalid, u32 lane) {
    // 31 -> 32 repack (mergeWords, kernels.cu:375): output word 62 s + l takes stream bits [32 (62 s + l), +32) =
    // groups 64 s + l + (l >= 31) and the next one, shifted by l mod 31.  Lane L decodes group 64 s + L; lanes 0..30
    // build words 0..30 and lanes 32..62 words 31..61 from their own group and the neighbour's (one DPP shift),
    // lanes 31 and 63 only lend their group.
    const u32 o = lane & 31u;
    const u32 up = 31u - ((lane - 1u) & 31u);             // what my group is shifted up by in my LEFT neighbour's word
    const u32 soff = o != 31u ? (lane - (lane >> 5)) * 4u : 0xFFFFF000u; // lanes 31, 63: out of range, dropped
    const uint4 fq = reinterpret_cast<const uint4 *>(flag)[lane];
    const u32 f[4] = {fq.x, fq.y, fq.z, fq.w};
    u32 before4 = (first_word - 1u) * 4u;                 // byte offset of (first_word + flags in earlier steps - 1)
#pragma unroll
    for (int s = 0; s < (int)kSteps; ++s) {
        const u32 fb = (f[s >> 2] >> (8 * (s & 3))) & 0xFFu;  // 1: a word starts at my group
        const u64 m = __ballot(fb != 0u);
        // inclusive rank among this step's flags (the count is seeded with my own flag), plus all earlier ones
        const u32 r4 = (__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, fb)) << 2) + before4;
        before4 = (u32)__builtin_amdgcn_readlane((int)r4, 63); // the last lane's rank counts every flag so far
        const u32 src_word = kLocal ? *reinterpret_cast<const u32 *>(reinterpret_cast<const unsigned char *>(s_words) + r4)
                                    : tile_word(s_words, a, tile_w0, r4 >> 2);
        // fill -> 31 copies of bit 30, literal -> itself (kernels.cu:332-354)
        const u32 fill_val = (u32)((int)(src_word << 1) >> 31) & kOnes31;
        u32 grp = (int)src_word < 0 ? fill_val : src_word;
        if (!kWhole && (u32)(64 * s) + lane >= nvalid) grp = 0u;
        const u32 hi_part = (u32)__builtin_amdgcn_mov_dpp((int)(grp << up), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        const u32 word = (grp >> o) | hi_part;
        __builtin_amdgcn_raw_buffer_store_b32(word, rsrc, soff + 248u * s, 0, 0);
    }
}

// flags are in place: expand the segment
__device__ __forceinline__ void expand_emit(const ExpandArgs &a, const u32 *s_words, const unsigned char *flag, u64 tile_w0,
                                            u32 first_word, u32 nvalid, u64 out_words, u64 seg, bool local, u32 lane) {
    const bool whole = nvalid == kSegGroups && (seg + 1) * kSegWords <= out_words;   // wave-uniform
    const u64 seg_w0 = seg * kSegWords;
    const u32 seg_words = whole ? kSegWords : (out_words > seg_w0 ? (u32)(out_words - seg_w0 < kSegWords ? out_words - seg_w0 : kSegWords) : 0u);
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + seg_w0, seg_words * 4u); // stores past the end are dropped
    if (whole) {
        if (local)
            expand_steps<true, true>(a, s_words, flag, tile_w0, rsrc, first_word, nvalid, lane);
        else
            expand_steps<true, false>(a, s_words, flag, tile_w0, rsrc, first_word, nvalid, lane);
    } else {
        expand_steps<false, false>(a, s_words, flag, tile_w0, rsrc, first_word, nvalid, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ---- one output segment, general version: any counts, 64-bit positions (foreign streams with giant fills) ---------
__device__ __forceinline__ void expand_segment_general(const ExpandArgs &a, const u32 *s_words, const u64 *s_coarse,
                                                       unsigned char *flag, u64 tile_w0, u64 base, u64 groups,
                                                       u64 out_words, u64 seg, u32 lane) {
    const u64 target = seg * kSegGroups - base; // tile-relative position of the segment's first group
    const u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
    // 64-ary search over the coarse prefix: last 64-word bucket that starts at or before the target
    const u64 c = lane < kCoarse ? s_coarse[lane] : ~0ull;
    const u32 bucket = (u32)__popcll(__ballot(c <= target)) - 1u;
    const u64 drop = target - uniform64(s_coarse[bucket]); // groups of the bucket in front of the segment

    reinterpret_cast<uint4 *>(flag)[lane] = make_uint4(0, 0, 0, 0); // 64 lanes x 16 B = the 1024 flags
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // mark the first group of every word that contributes to the segment (clipped at the segment start)
    u32 first_word = 0; // tile-local index of the word that covers the segment's first group
    bool have_first = false;
    u64 seen = 0;       // groups of the words looked at so far, from the bucket start
    u32 wi = bucket * 64u;
    while (seen < drop + nvalid && tile_w0 + wi < a.c_words) {
        const u32 idx = wi + lane;
        const bool in = tile_w0 + idx < a.c_words;
        const u32 ww = in ? tile_word(s_words, a, tile_w0, idx) : 0u;
        const u32 n = in ? word_groups(ww) : 0u;
        bool contributes;
        u32 p = 0;
        u64 batch_total;
        if (__ballot(n > (1u << 20)) == 0 && drop - (seen < drop ? seen : drop) < (1ull << 27)) {
            // common case: everything fits 32 bits relative to `seen`
            const u32 incl_n = wave_scan_incl32(n);
            const u32 lead = (u32)(drop - (seen < drop ? seen : drop)); // groups still to drop in this batch
            const u32 past = seen > drop ? (u32)(seen - drop) : 0u;     // segment groups already covered
            const u32 lo = incl_n - n, hi = incl_n;
            contributes = n != 0 && hi > lead && lo + past < lead + nvalid;
            p = (lo > lead ? lo - lead : 0u) + past;
            batch_total = (u32)__builtin_amdgcn_readlane((int)incl_n, 63);
        } else {
            const u64 incl_n = wave_scan_incl((u64)n, lane);
            const u64 lo = seen + (incl_n - n), hi = seen + incl_n; // the word covers [lo, hi) from the bucket start
            contributes = n != 0 && hi > drop && lo < drop + nvalid;
            p = lo > drop ? (u32)(lo - drop) : 0u;
            batch_total = uniform64(__shfl(incl_n, 63));
        }
        if (contributes) flag[flag_slot(p)] = 1; // distinct groups: plain byte stores, no atomics
        const u64 cmask = __ballot(contributes);
        if (!have_first && cmask) {
            first_word = uniform32(wi + (u32)__ffsll((long long)cmask) - 1u);
            have_first = true;
        }
        seen += batch_total;
        wi += 64u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (seen < drop + nvalid) { // the stream ended inside the segment: cannot happen for a consistent scan
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, false, lane);
}

// ---- one output segment, streams with fill words of count 0 (foreign streams only; the reference decoder steps over
// such words, kernels.cu:332-354).  The rank arithmetic of the routines above assumes that consecutive contributing
// words are consecutive in the stream, which an empty word in between breaks.  Here every contributing word writes
// its own index at the group it starts at, and a group's source is the last index written at or before it (a
// running maximum: indices grow with position).  One wavefront, 4 KiB of LDS (`src`), any counts. ----------------
__device__ __forceinline__ void expand_segment_with_empties(const ExpandArgs &a, const u32 *s_words, const u64 *s_coarse,
                                                            u32 *src, u64 tile_w0, u64 base, u64 groups, u64 out_words,
                                                            u64 seg, u32 lane) {
    const u64 target = seg * kSegGroups - base; // tile-relative position of the segment's first group
    const u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
    const u64 c = lane < kCoarse ? s_coarse[lane] : ~0ull;
    const u32 bucket = (u32)__popcll(__ballot(c <= target)) - 1u;
    const u64 drop = target - uniform64(s_coarse[bucket]); // groups of the bucket in front of the segment
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<uint4 *>(src)[k * 64 + (int)lane] = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    u64 seen = 0; // groups of the words looked at so far, from the bucket start
    u32 wi = bucket * 64u;
    while (seen < drop + nvalid && tile_w0 + wi < a.c_words) {
        const u32 idx = wi + lane;
        const bool in = tile_w0 + idx < a.c_words;
        const u32 n = in ? word_groups(tile_word(s_words, a, tile_w0, idx)) : 0u;
        const u64 incl_n = wave_scan_incl((u64)n, lane);
        const u64 lo = seen + (incl_n - n), hi = seen + incl_n; // the word covers [lo, hi) from the bucket start
        if (n != 0u && hi > drop && lo < drop + nvalid) src[lo > drop ? (u32)(lo - drop) : 0u] = idx + 1u;
        seen += uniform64(__shfl(incl_n, 63));
        wi += 64u;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (seen < drop + nvalid) { // the stream ended inside the segment: cannot happen for a consistent scan
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    const u64 seg_w0 = seg * kSegWords;
    const u32 seg_words = out_words > seg_w0 ? (u32)(out_words - seg_w0 < kSegWords ? out_words - seg_w0 : kSegWords) : 0u;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + seg_w0, seg_words * 4u); // stores past the end are dropped
    const u32 o = lane & 31u;
    const u32 up = 31u - ((lane - 1u) & 31u);
    const u32 soff = o != 31u ? (lane - (lane >> 5)) * 4u : 0xFFFFF000u;
    u32 carry = 0;
    for (u32 s = 0; s < kSteps; ++s) {
        const u32 m = max(wave_scan_max32(src[64u * s + lane]), carry); // index + 1 of the word my group belongs to
        carry = (u32)__builtin_amdgcn_readlane((int)m, 63);
        const u32 src_word = tile_word(s_words, a, tile_w0, m ? m - 1u : 0u);
        const u32 fill_val = (u32)((int)(src_word << 1) >> 31) & kOnes31;
        u32 grp = (int)src_word < 0 ? fill_val : src_word;
        if (64u * s + lane >= nvalid) grp = 0u;
        const u32 hi_part = (u32)__builtin_amdgcn_mov_dpp((int)(grp << up), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        __builtin_amdgcn_raw_buffer_store_b32((grp >> o) | hi_part, rsrc, soff + 248u * s, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// ---- one output segment, fast version: the tile expands to fewer than 2^31 groups, so every position relative to the
// segment start fits a signed 32-bit integer; bookkeeping stays in vector registers (see compress_kernel) ----------

// One batch of the mark phase: the next 128 words, two per lane.  `rel` = where the batch starts, seen from the
// segment start (<= 0 at first).  Flags the group at which every contributing word starts (clipped at the segment
// start).  kFirst: returns the tile-local index of the first contributing word.
template <bool kLocal, bool kFirst>
__device__ __forceinline__ u32 mark_pairs(const ExpandArgs &a, const u32 *s_words, unsigned char *flag, u64 tile_w0,
                                          u32 left_in_stream, u32 nvalid, u32 lane, int &rel, u32 &wi) {
    const u32 i0 = wi + 2u * lane;
    u32 w0, w1;
    if (kLocal) {
        const uint2 q = *reinterpret_cast<const uint2 *>(s_words + i0);
        w0 = q.x;
        w1 = q.y;
    } else { // past the tile: global memory; past the stream: empty fills
        w0 = i0 < left_in_stream ? a.comp[tile_w0 + i0] : 0x80000000u;
        w1 = i0 + 1u < left_in_stream ? a.comp[tile_w0 + i0 + 1u] : 0x80000000u;
    }
    const u32 n0 = word_groups(w0), n1 = word_groups(w1);
    // all literals (dense data): consecutive positions, no scan; otherwise one DPP scan over the pair sums
    const u32 incl = __ballot((int)(w0 | w1) < 0) == 0 ? 2u * lane + 2u : wave_scan_incl32(n0 + n1);
    const int hi1 = rel + (int)incl, lo1 = hi1 - (int)n1, lo0 = lo1 - (int)n0; // words cover [lo0, lo1) and [lo1, hi1)
    // clip to the segment [0, nvalid): a word contributes iff something is left of it
    const int s0 = lo0 > 0 ? lo0 : 0, e0 = lo1 < (int)nvalid ? lo1 : (int)nvalid;
    const int s1 = lo1 > 0 ? lo1 : 0, e1 = hi1 < (int)nvalid ? hi1 : (int)nvalid;
    const bool c0 = e0 > s0, c1 = e1 > s1;
    // distinct groups: plain byte stores, no atomics.  slot(p) + base = base + 16 p - 1023 (p / 64): three instructions
    const u32 fbase = (u32)(uintptr_t)(lds_u8_ptr)flag;
    const u32 dump = fbase + kSegGroups + lane;
    const u32 a0 = (u32)__mul24(s0 >> 6, -1023) + (((u32)s0 << 4) + fbase);
    const u32 a1 = (u32)__mul24(s1 >> 6, -1023) + (((u32)s1 << 4) + fbase);
    *(lds_u8_ptr)(uintptr_t)(c0 ? a0 : dump) = 1;
    *(lds_u8_ptr)(uintptr_t)(c1 ? a1 : dump) = 1;
    u32 first = 0;
    if (kFirst) {
        const u64 m0 = __ballot(c0), m1 = __ballot(c1);
        const u32 l = (u32)__ffsll((long long)(m0 | m1)) - 1u;
        first = wi + 2u * l + (((m0 >> l) & 1ull) ? 0u : 1u);
    }
    rel += (int)(u32)__builtin_amdgcn_readlane((int)incl, 63);
    wi += 128u;
    return first;
}

__device__ __forceinline__ void expand_segment_tame(const ExpandArgs &a, const u32 *s_words, const u32 *s_coarse32,
                                                    unsigned char *flag, u64 tile_w0, u32 target, u32 nvalid,
                                                    u64 out_words, u64 seg, u32 lane) {
    // 64-ary search over the coarse prefix: last 64-word bucket that starts at or before the target
    const u32 c = lane <= kCoarse ? s_coarse32[lane] : 0xFFFFFFFFu;
    const u32 bucket = (u32)__popcll(__ballot(lane < kCoarse && c <= target)) - 1u;
    int rel = (int)(uniform32(s_coarse32[bucket]) - target); // <= 0: where the bucket starts, seen from the segment

    reinterpret_cast<uint4 *>(flag)[lane] = make_uint4(0, 0, 0, 0); // 64 lanes x 16 B = the 1024 flags
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const u32 left_in_stream = a.c_words - tile_w0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)(a.c_words - tile_w0);
    u32 wi = bucket * 64u;
    constexpr u32 kLastLocal = (u32)kScanTileWords - 128u; // batches starting up to here come out of the LDS tile
    // the word that covers the segment's first group is in the first batch (that is how the bucket was chosen)
    const u32 first_word = wi <= kLastLocal ? mark_pairs<true, true>(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi)
                                            : mark_pairs<false, true>(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi);
    while (rel < (int)nvalid && wi <= kLastLocal)
        (void)mark_pairs<true, false>(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi);
    while (rel < (int)nvalid && wi < left_in_stream)
        (void)mark_pairs<false, false>(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (rel < (int)nvalid) { // the stream ended inside the segment: cannot happen for a consistent scan
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    // group g belongs to the r-th contributing word, r = (flags at positions <= g) - 1
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, wi <= (u32)kScanTileWords, lane);
}

__global__ __launch_bounds__(kExpandThreads) void decode_expand_kernel(const ExpandArgs a) {
    __shared__ __attribute__((aligned(16))) u32 s_words[kScanTileWords];
    __shared__ u64 s_coarse[kCoarse + 1]; // groups in front of word 64 c, relative to the tile start
    __shared__ u32 s_coarse32[kCoarse + 1]; // the same in 32 bits (valid when the tile total is below 2^31)
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ __attribute__((aligned(16))) unsigned char s_flag[kExpandWaves][kFlagBytes]; // 1: a word starts at this group

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    // a stream of few tiles (highly compressed data) expands to many segments per tile: `parts` workgroups share a tile
    const u32 tile = blockIdx.x / a.parts;
    const u32 part = blockIdx.x % a.parts;
    const u64 tile_w0 = (u64)tile * kScanTileWords;
    const u64 groups = a.info[1];
    const u64 out_words = a.info[0];
    // (asked for up front, used after the prologue: this tile or the next one contains fill words of count 0)
    const u64 all_tiles = (a.c_words + kScanTileWords - 1) / kScanTileWords;
    const bool has_empties = (a.tile_flags[tile] | ((u64)tile + 1 < all_tiles ? a.tile_flags[tile + 1] : 0)) != 0;
    if (out_words > a.out_capacity) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(a.ctrl + kCtlError, kErrCapacity);
        return;
    }

    // ---- stage the tile and build the coarse prefix of group counts --------------------------------------------
    constexpr int kVec = kExpandWordsPerThread / 4;
    if (a.aligned16 && tile_w0 + kScanTileWords <= a.c_words) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + tile_w0);
        uint4 *dst = reinterpret_cast<uint4 *>(s_words);
        uint4 v[kVec];
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] = src[k * kExpandThreads + (int)threadIdx.x]; // coalesced 16-byte loads
#pragma unroll
        for (int k = 0; k < kVec; ++k) dst[k * kExpandThreads + (int)threadIdx.x] = v[k];
    } else {
        for (u32 i = threadIdx.x; i < (u32)kScanTileWords; i += kExpandThreads)
            s_words[i] = tile_w0 + i < a.c_words ? a.comp[tile_w0 + i] : 0x80000000u; // past the end: empty fill
    }
    __syncthreads();
    // every thread sums the counts of its own 16 consecutive words
    u64 mine = 0;
    {
        const uint4 *my = reinterpret_cast<const uint4 *>(s_words + threadIdx.x * kExpandWordsPerThread);
#pragma unroll
        for (int k = 0; k < kVec; ++k) {
            const uint4 q = my[k];
            mine += (u64)(word_groups(q.x) + word_groups(q.y) + word_groups(q.z) + word_groups(q.w));
        }
    }
    const u64 incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_wave_sum[wave] = incl;
    __syncthreads();
    u64 excl = incl - mine;
    for (u32 k = 0; k < wave; ++k) excl += s_wave_sum[k];
    constexpr u32 kThreadsPer64 = 64 / kExpandWordsPerThread;
    if (threadIdx.x % kThreadsPer64 == 0) { // first thread of each 64 words
        s_coarse[threadIdx.x / kThreadsPer64] = excl;
        s_coarse32[threadIdx.x / kThreadsPer64] = (u32)excl;
    }
    if (threadIdx.x == kExpandThreads - 1) {
        s_coarse[kCoarse] = excl + mine;
        s_coarse32[kCoarse] = (u32)(excl + mine);
    }
    __syncthreads();

    // ---- segments owned by this tile: those whose first group lies in [base, base + total) ----------------------
    const u64 base = a.tile_base[tile];
    const u64 total = uniform64(s_coarse[kCoarse]);
    const u64 n_seg = (groups + kSegGroups - 1) / kSegGroups;
    const u64 k_begin = (base + kSegGroups - 1) / kSegGroups;
    u64 k_end = (base + total + kSegGroups - 1) / kSegGroups;
    if (k_end > n_seg) k_end = n_seg;

    if (has_empties) {
        // this tile, or the next one (a segment reads at most 1024 + 128 words past its tile), contains fill words of
        // count 0 (found by the sums pass): index-map route, one wavefront per workgroup, the four flag areas together
        // hold its 1024-entry index map
        static_assert(sizeof(s_flag) >= kSegGroups * sizeof(u32), "index map must fit the flag areas");
        if (wave == 0)
            for (u64 seg = k_begin + part; seg < k_end; seg += a.parts)
                expand_segment_with_empties(a, s_words, s_coarse, reinterpret_cast<u32 *>(&s_flag[0][0]), tile_w0, base, groups,
                                            out_words, seg, lane);
        return;
    }
    unsigned char *flag = s_flag[wave];
    const bool tame = total < (1ull << 31); // wave-uniform: positions inside this tile fit 32 bits
    for (u64 seg = k_begin + wave + (u64)kExpandWaves * part; seg < k_end; seg += (u64)kExpandWaves * a.parts) {
        if (tame) {
            const u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
            expand_segment_tame(a, s_words, s_coarse32, flag, tile_w0, (u32)(seg * kSegGroups - base), nvalid, out_words, seg, lane);
        } else {
            expand_segment_general(a, s_words, s_coarse, flag, tile_w0, base, groups, out_words, seg, lane);
        }
    }
}

// ===========================================================================
// stream checker (include/wah.h: wah_validate_device).  One workgroup per 4096-word tile, after the sums pass: the
// tile bases give every word its group position, so the per-word properties that depend on position (a fill crossing
// a 1024-group boundary, two mergeable fills inside one segment) can be told from the ones that do not.
// ===========================================================================
__global__ __launch_bounds__(kExpandThreads) void validate_kernel(const u32 *comp, u64 c_words, const u64 *tile_base,
                                                                  const u64 *info, u64 *report) {
    __shared__ u64 s_wave_sum[kExpandWaves];
    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const u32 tile = blockIdx.x;
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread; // my 16 consecutive words
    u32 w[kExpandWordsPerThread];
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        w[k] = w0 + k < c_words ? comp[w0 + k] : 0x80000000u; // past the end: nothing
        mine += word_groups(w[k]);
    }
    const u64 incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_wave_sum[wave] = incl;
    __syncthreads();
    u64 p = tile_base[tile] + (incl - mine); // group position of my first word
    for (u32 k = 0; k < wave; ++k) p += s_wave_sum[k];

    u32 prev = w0 > 0 && w0 - 1 < c_words ? comp[w0 - 1] : 0u; // the word in front of mine (a literal 0 if none)
    const bool have_prev = w0 > 0;
    u32 n_empty = 0, n_litfill = 0, n_cross = 0, n_unmerged = 0;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        const u32 x = w[k];
        if (w0 + k < c_words) {
            const bool fill = (x & kFillZero) != 0;
            const u32 cnt = x & kCountMask;
            n_empty += fill && cnt == 0u;
            n_litfill += !fill && (x == 0u || x == kOnes31);
            n_cross += fill && cnt != 0u && (p & (kSegGroups - 1u)) + cnt > kSegGroups;
            const bool prev_fill = (k > 0 || have_prev) && (prev & kFillZero) && (prev & kCountMask) != 0u;
            n_unmerged += fill && cnt != 0u && prev_fill && ((prev ^ x) & 0x40000000u) == 0u && (p & (kSegGroups - 1u)) != 0u;
        }
        p += word_groups(x);
        prev = x;
    }
    const u32 t_empty = wave_sum32(n_empty), t_lit = wave_sum32(n_litfill), t_cross = wave_sum32(n_cross), t_unm = wave_sum32(n_unmerged);
    if (lane == 0) {
        if (t_empty) atomicAdd(reinterpret_cast<unsigned long long *>(report + 2), (unsigned long long)t_empty);
        if (t_lit) atomicAdd(reinterpret_cast<unsigned long long *>(report + 3), (unsigned long long)t_lit);
        if (t_cross) atomicAdd(reinterpret_cast<unsigned long long *>(report + 4), (unsigned long long)t_cross);
        if (t_unm) atomicAdd(reinterpret_cast<unsigned long long *>(report + 5), (unsigned long long)t_unm);
        if (t_empty | t_lit | t_cross | t_unm) report[6] = 0; // (set to 1 by validate_init_kernel)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        report[0] = info[1];
        report[1] = info[0];
    }
}

__global__ void validate_init_kernel(u64 *report) {
    if (threadIdx.x < 8) report[threadIdx.x] = threadIdx.x == 6 ? 1ull : 0ull;
}

// ===========================================================================
// bench support
// ===========================================================================
__global__ void gen_uniform_kernel(u32 *out, u64 n, u64 seed, u64 thr) {
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n; w += (u64)gridDim.x * blockDim.x)
        out[w] = wah_gen_uniform_word(seed, w, thr);
}

__global__ void gen_clustered_kernel(u32 *out, u64 n, u64 seed, u64 thr) {
    const u64 chunks = (n + WAH_GEN_CHUNK_WORDS - 1) / WAH_GEN_CHUNK_WORDS;
    for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (u64)gridDim.x * blockDim.x) {
        const u64 w0 = c * WAH_GEN_CHUNK_WORDS;
        const u64 left = n - w0;
        wah_gen_clustered_chunk(seed, c, thr, out + w0, (u32)(left < WAH_GEN_CHUNK_WORDS ? left : WAH_GEN_CHUNK_WORDS));
    }
}

__global__ __launch_bounds__(256) void copy_kernel(const uint4 *in, uint4 *out, u64 n16) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) out[i] = in[i];
}

// The census counts what was resident at one moment; near the edge that depends on how the dispatcher happened to
// place the wavefronts (measured: census 1184, runs above ~1060 workgroups lost a workgroup).  Keep a margin: only
// whole multiples of the CU count are used, i.e. what EVERY compute unit can hold.
int whole_per_cu(int resident) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus < 1) cus = 1;
    return resident >= cus ? (resident / cus) * cus : resident;
}

int persistent_grid(const void *kernel, int threads, u64 n_tiles) {
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

// Grid of the persistent compress kernel = how many of its workgroups are resident together, measured once per
// device by a census launch of the same kernel (the occupancy API is advisory: MI355X_MICROARCH residency notes).
hipError_t launch_clear(void *p, size_t bytes, hipStream_t s);

template <int W>
int compress_grid_for(u32 *d_ctrl, hipStream_t s) {
    static int cached[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] > 0) return cached[dev];
    const int upper = persistent_grid(reinterpret_cast<const void *>(&compress_kernel<W>), (W + 1) * 64, ~0ull);
    CompressArgs a = {};
    a.ctrl = d_ctrl;
    a.census = 1;
    int resident = 0;
    if (launch_clear(d_ctrl, kCtlWords * sizeof(u32), s) == hipSuccess) {
        hipLaunchKernelGGL(compress_kernel<W>, dim3(upper), dim3((W + 1) * 64), 0, s, a);
        u32 seen = 0;
        if (hipGetLastError() == hipSuccess &&
            hipMemcpyAsync(&seen, d_ctrl + kCtlCensus, sizeof seen, hipMemcpyDeviceToHost, s) == hipSuccess &&
            hipStreamSynchronize(s) == hipSuccess)
            resident = (int)seen;
    }
    if (resident < 1) return -1;
    if (resident > upper) resident = upper;
    resident = whole_per_cu(resident);
    cached[dev] = resident;
    return resident;
}

int compress_grid(int workers, u32 *d_ctrl, hipStream_t s) {
    return workers == 15 ? compress_grid_for<15>(d_ctrl, s) : compress_grid_for<7>(d_ctrl, s);
}

hipError_t launch_compress(int workers, const CompressArgs &a, int grid, hipStream_t s) {
    if (workers == 15)
        hipLaunchKernelGGL(compress_kernel<15>, dim3(grid), dim3(16 * 64), 0, s, a);
    else
        hipLaunchKernelGGL(compress_kernel<7>, dim3(grid), dim3(8 * 64), 0, s, a);
    return hipGetLastError();
}

// wah_bitop_device: both operands must have expanded to the bitmap length the caller named, without errors of their own
__global__ void bitop_check_kernel(const u64 *info_a, const u64 *info_b, const u32 *ctrl_a, const u32 *ctrl_b, u64 groups, u32 *ctrl) {
    if (threadIdx.x == 0) {
        u32 err = ctrl_a[kCtlError] | ctrl_b[kCtlError];
        if (info_a[1] != groups || info_b[1] != groups) err |= kErrStream;
        if (err) atomicOr(ctrl + kCtlError, err);
    }
}
hipError_t launch_bitop_check(const u64 *info_a, const u64 *info_b, const u32 *ctrl_a, const u32 *ctrl_b, u64 groups, u32 *ctrl,
                              hipStream_t s) {
    hipLaunchKernelGGL(bitop_check_kernel, dim3(1), dim3(64), 0, s, info_a, info_b, ctrl_a, ctrl_b, groups, ctrl);
    return hipGetLastError();
}

// pair mode (wah_bitop_device): same kernel, two inputs combined while they are staged; needs the fast path
hipError_t launch_compress_pair(const CompressArgs &a, int grid, hipStream_t s) {
    hipLaunchKernelGGL((compress_kernel<15, true>), dim3(grid), dim3(16 * 64), 0, s, a);
    return hipGetLastError();
}

int decode_sums_grid(u32 *d_ctrl, hipStream_t s) {
    static int cached[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] > 0) return cached[dev];
    const int upper = persistent_grid(reinterpret_cast<const void *>(&decode_sums_kernel), (kSumWorkers + 1) * 64, ~0ull);
    ScanArgs a = {};
    a.ctrl = d_ctrl;
    a.census = 1;
    int resident = 0;
    if (launch_clear(d_ctrl, kCtlWords * sizeof(u32), s) == hipSuccess) {
        hipLaunchKernelGGL(decode_sums_kernel, dim3(upper), dim3((kSumWorkers + 1) * 64), 0, s, a);
        u32 seen = 0;
        if (hipGetLastError() == hipSuccess &&
            hipMemcpyAsync(&seen, d_ctrl + kCtlCensus, sizeof seen, hipMemcpyDeviceToHost, s) == hipSuccess &&
            hipStreamSynchronize(s) == hipSuccess)
            resident = (int)seen;
    }
    if (resident < 1) return -1;
    if (resident > upper) resident = upper;
    resident = whole_per_cu(resident);
    cached[dev] = resident;
    return resident;
}

hipError_t launch_decode_sums(const ScanArgs &a, int grid, hipStream_t s) {
    hipLaunchKernelGGL(decode_sums_kernel, dim3(grid), dim3((kSumWorkers + 1) * 64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_decode_expand(const ExpandArgs &a0, u64 n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    // One workgroup per tile fills the chip when the stream is long.  A short stream that expands a lot (highly
    // compressed bitmaps: thousands of output segments per tile) is shared out: `parts` workgroups per tile, each
    // taking every parts-th group of kExpandWaves segments.  The true output size is only known on the device; the
    // capacity bounds it, and a part with nothing to do costs one 16 KiB tile read.
    ExpandArgs a = a0;
    const u64 want = 4096; // workgroups: 256 CUs x 7 resident x ~2
    const u64 segs_per_tile = a.out_capacity / kSegWords / n_tiles;
    u64 parts = (want + n_tiles - 1) / n_tiles;
    if (parts > segs_per_tile / (2 * kExpandWaves)) parts = segs_per_tile / (2 * kExpandWaves);
    if (parts < 1) parts = 1;
    if (parts > 1024) parts = 1024;
    a.parts = (u32)parts;
    hipLaunchKernelGGL(decode_expand_kernel, dim3((unsigned)(n_tiles * parts)), dim3(kExpandThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_validate(const u32 *comp, u64 c_words, const u64 *tile_base, const u64 *info, u64 *report, u64 n_tiles,
                           hipStream_t s) {
    hipLaunchKernelGGL(validate_init_kernel, dim3(1), dim3(64), 0, s, report);
    if (n_tiles) hipLaunchKernelGGL(validate_kernel, dim3((unsigned)n_tiles), dim3(kExpandThreads), 0, s, comp, c_words, tile_base, info, report);
    return hipGetLastError();
}

// Zero `bytes` bytes (a multiple of 4) at p (4-byte aligned).  A kernel of our own rather than hipMemsetAsync: the
// device-pointer API is meant to be captured into HIP graphs, and a captured memset node did not reliably write
// zeros when the graph was replayed (ROCm 7.2, observed on MI355X: tests/test_gpu_parity.py graph test).
__global__ void clear_kernel(u32 *p, u64 n_words) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (u64)gridDim.x * blockDim.x) p[i] = 0u;
}
hipError_t launch_clear(void *p, size_t bytes, hipStream_t s) {
    const u64 n = bytes / 4;
    if (n == 0) return hipSuccess;
    const u64 blocks = (n + 255) / 256;
    hipLaunchKernelGGL(clear_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, static_cast<u32 *>(p), n);
    return hipGetLastError();
}

// ===========================================================================
// wah_merge_fills_device (include/wah.h): adjacent fills of the same kind become one word, empty fills disappear.
// Tile-based like the checker: the sums pass gives every tile its group position, an in-workgroup scan gives every word
// its own.  A word is DROPPED if it is an empty fill, or a fill whose predecessor is a non-empty fill of the same kind
// and both lie inside one block of 2^29 groups (so that no merged count can outgrow 30 bits).  Kept words move up by
// the number of dropped words in front of them; a kept fill's new count is the distance to the next kept word.
// ===========================================================================
constexpr u32 kMergeBlockShift = 29;

struct MergeTile {
    u32 w[kExpandWordsPerThread];
    u64 p;          // group position of w[0]
    u32 prev;       // the word in front of w[0]
    bool have_prev;
};

__device__ __forceinline__ void merge_load_tile(const MergeArgs &a, u32 tile, u64 *s_wave_sum, u32 lane, u32 wave, MergeTile &t) {
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread;
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        t.w[k] = w0 + k < a.c_words ? a.comp[w0 + k] : 0x80000000u; // past the end: nothing
        mine += word_groups(t.w[k]);
    }
    const u64 incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_wave_sum[wave] = incl;
    __syncthreads();
    t.p = a.tile_base[tile] + (incl - mine);
    for (u32 k = 0; k < wave; ++k) t.p += s_wave_sum[k];
    t.have_prev = w0 > 0;
    t.prev = w0 > 0 && w0 - 1 < a.c_words ? a.comp[w0 - 1] : 0u;
}

// is word x (at group position p, behind word prev) dropped?
__device__ __forceinline__ bool merge_dropped(u32 x, u32 prev, bool have_prev, u64 p) {
    if (!(x & kFillZero)) return false;
    const u32 cnt = x & kCountMask;
    if (cnt == 0u) return true;
    if (!have_prev || !(prev & kFillZero)) return false;
    const u32 pcnt = prev & kCountMask;
    if (pcnt == 0u || ((prev ^ x) & 0x40000000u)) return false;
    return ((p - pcnt) >> kMergeBlockShift) == ((p + cnt - 1u) >> kMergeBlockShift);
}

__global__ __launch_bounds__(kExpandThreads) void merge_count_kernel(const MergeArgs a) {
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ u32 s_kept[kExpandWaves];
    const u32 lane = lane_id(), wave = wave_id(), tile = blockIdx.x;
    MergeTile t;
    merge_load_tile(a, tile, s_wave_sum, lane, wave, t);
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread;
    u32 kept = 0, prev = t.prev;
    bool have_prev = t.have_prev;
    u64 p = t.p;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        if (w0 + k < a.c_words) kept += !merge_dropped(t.w[k], prev, have_prev, p);
        p += word_groups(t.w[k]);
        prev = t.w[k];
        have_prev = true;
    }
    const u32 wk = wave_sum32(kept);
    if (lane == 0) s_kept[wave] = wk;
    __syncthreads();
    if (threadIdx.x == 0) a.tile_kept[tile] = (u64)s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3];
}

// exclusive scan of tile_kept[0 .. n_tiles) in place, total into tile_kept[n_tiles] and *out_words (one workgroup)
__global__ __launch_bounds__(1024) void merge_scan_kernel(const MergeArgs a) {
    __shared__ u64 s_part[16];
    __shared__ u64 s_carry;
    const u32 lane = lane_id(), wave = wave_id();
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < a.n_tiles; base += 1024) {
        const u64 i = base + threadIdx.x;
        const u64 v = i < a.n_tiles ? a.tile_kept[i] : 0;
        const u64 incl = wave_scan_incl(v, lane);
        if (lane == 63) s_part[wave] = incl;
        __syncthreads();
        u64 excl = incl - v + s_carry;
        for (u32 k = 0; k < wave; ++k) excl += s_part[k];
        if (i < a.n_tiles) a.tile_kept[i] = excl;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const u64 total = s_carry;
        a.tile_kept[a.n_tiles] = total;
        *a.out_words = total;
        if (total > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
    }
}

__global__ __launch_bounds__(kExpandThreads) void merge_scatter_kernel(const MergeArgs a) {
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ u32 s_kept[kExpandWaves];
    const u32 lane = lane_id(), wave = wave_id(), tile = blockIdx.x;
    MergeTile t;
    merge_load_tile(a, tile, s_wave_sum, lane, wave, t);
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread;
    bool keep[kExpandWordsPerThread];
    u64 pos[kExpandWordsPerThread];
    u32 kept = 0, prev = t.prev;
    bool have_prev = t.have_prev;
    u64 p = t.p;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        keep[k] = w0 + k < a.c_words && !merge_dropped(t.w[k], prev, have_prev, p);
        pos[k] = p;
        kept += keep[k];
        p += word_groups(t.w[k]);
        prev = t.w[k];
        have_prev = true;
    }
    const u32 incl = wave_scan_incl32(kept);
    if (lane == 63) s_kept[wave] = incl;
    __syncthreads();
    u64 idx = a.tile_kept[tile] + (incl - kept);
    for (u32 k = 0; k < wave; ++k) idx += s_kept[k];
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        if (keep[k]) {
            if (idx < a.out_capacity) {
                a.out[idx] = t.w[k];
                a.positions[idx] = pos[k];
            }
            ++idx;
        }
    }
}

// a kept fill covers everything up to the next kept word
__global__ void merge_fix_kernel(const MergeArgs a) {
    const u64 kept = a.tile_kept[a.n_tiles] < a.out_capacity ? a.tile_kept[a.n_tiles] : a.out_capacity;
    const u64 groups = a.info[1];
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < kept; i += (u64)gridDim.x * blockDim.x) {
        const u32 x = a.out[i];
        if ((x & kFillZero) && (x & kCountMask)) {
            const u64 next = i + 1 < a.tile_kept[a.n_tiles] && i + 1 < a.out_capacity ? a.positions[i + 1] : groups;
            const u64 cnt = next - a.positions[i];
            if (cnt > kCountMask)
                atomicOr(a.ctrl + kCtlError, kErrStream); // cannot happen: runs do not cross 2^29-group blocks
            else if ((u32)cnt != (x & kCountMask))
                a.out[i] = (x & ~kCountMask) | (u32)cnt;
        }
    }
}

hipError_t launch_merge_fills(const MergeArgs &a, hipStream_t s) {
    if (a.n_tiles == 0) {
        hipLaunchKernelGGL(clear_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<u32 *>(a.out_words), (u64)2);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(merge_count_kernel, dim3((unsigned)a.n_tiles), dim3(kExpandThreads), 0, s, a);
    hipLaunchKernelGGL(merge_scan_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(merge_scatter_kernel, dim3((unsigned)a.n_tiles), dim3(kExpandThreads), 0, s, a);
    hipLaunchKernelGGL(merge_fix_kernel, dim3(2048), dim3(256), 0, s, a);
    return hipGetLastError();
}


hipError_t launch_gen_uniform(u32 *out, u64 n, u64 seed, u64 thr, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gen_uniform_kernel, dim3(4096), dim3(256), 0, s, out, n, seed, thr);
    return hipGetLastError();
}

hipError_t launch_gen_clustered(u32 *out, u64 n, u64 seed, u64 thr, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(gen_clustered_kernel, dim3(1024), dim3(64), 0, s, out, n, seed, thr);
    return hipGetLastError();
}

hipError_t launch_copy(const u32 *in, u32 *out, u64 n, hipStream_t s) {
    const u64 n16 = n / 4;
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, s, reinterpret_cast<const uint4 *>(in),
                       reinterpret_cast<uint4 *>(out), n16);
    return hipGetLastError();
}

} // namespace wah
