// wah_compress.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the WAH path: overview, and the compress kernel.
// (decompress: wah_decode.hip; checker, merge pass, bench support: wah_aux.hip; shared helpers: wah_device.hpp)
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
#include "wah_device.hpp"

namespace wah {
namespace {

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

struct Prefetch {
    u32x4 v[4];
};

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

        // (Pending tiles are emitted only when the ring needs the room, above: trying here as well -- "stream out
        // whatever has its offset already" -- costs an LDS poll per iteration that mostly fails and moves the emission in
        // front of the next tile's loads; without it the sparse GiB takes 0.349 instead of 0.373 ms, clustered 0.255
        // instead of 0.270, dense the same.)
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

} // namespace

// Grid of the persistent compress kernel = how many of its workgroups are resident together, measured once per
// device by a census launch of the same kernel (the occupancy API is advisory: MI355X_MICROARCH residency notes).

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

} // namespace wah
