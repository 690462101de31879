// wah_segdecode.hpp -- one 1024-group segment of an INDEXED compressed stream back into its 31-bit groups, in registers:
// the building block of the index decode (decode_segments_kernel, wah_decode.hip), of the combining passes over indexed
// operands, and of the bit operation that feeds the compress passes directly (bitop_tile_kernel, wah_compress.hip).
// Everything lives in an anonymous namespace: every translation unit gets its own copy, nothing is exported.
#ifndef WAH_SEGDECODE_HPP_
#define WAH_SEGDECODE_HPP_

#include "wah_device.hpp"

namespace wah {
namespace {

constexpr int kSegBatches = kSegGroups / 128;

// where segment `seg` of the bitmap lies in the stream
struct SegRange {
    u64 w0;
    u32 cnt, nvalid;
    u32 bad, spare; // (no padding bytes: a copy of the struct stays in registers instead of going through scratch)
};
__device__ __forceinline__ SegRange seg_range(const SegmentsArgs &a, u64 seg, u64 w0, u64 w1) {
    SegRange r;
    const u64 g0 = seg * kSegGroups;
    r.nvalid = a.groups - g0 < kSegGroups ? (u32)(a.groups - g0) : kSegGroups;
    // every word of a compress() stream covers at least one group
    r.bad = (w1 < w0 || w1 > a.c_words || w1 - w0 > r.nvalid) ? 1u : 0u;
    r.spare = 0;
    r.cnt = r.bad ? 0u : (u32)(w1 - w0);
    r.w0 = w0;
    return r;
}
// the segment's words: 128 per batch, two per lane (reads past the range return 0)
__device__ __forceinline__ void seg_load_words(const SegmentsArgs &a, const SegRange &r, u32 (&x0)[kSegBatches], u32 (&x1)[kSegBatches], u32 lane) {
    const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc(a.comp + r.w0, r.cnt * 4u);
    x0[0] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, 2u * lane * 4u, 0, 0);
    x1[0] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, 2u * lane * 4u + 4u, 0, 0);
    // (the batches behind the first under wave-uniform tests -- 128, 512 words: a segment of a clustered bitmap has 16 words,
    //  fourteen of the sixteen loads would come back as zeros; of the sparse bitmap 476: eight)
#pragma unroll
    for (int b = 1; b < kSegBatches; ++b) x0[b] = x1[b] = 0u;
    if (r.cnt > 128u) {
#pragma unroll
        for (int b = 1; b < 4; ++b) {
            x0[b] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (128u * b + 2u * lane) * 4u, 0, 0);
            x1[b] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (128u * b + 2u * lane) * 4u + 4u, 0, 0);
        }
        if (r.cnt > 512u) {
#pragma unroll
            for (int b = 4; b < kSegBatches; ++b) {
                x0[b] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (128u * b + 2u * lane) * 4u, 0, 0);
                x1[b] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (128u * b + 2u * lane) * 4u + 4u, 0, 0);
            }
        }
    }
}

// Mark phase of one segment: parks its words (x0/x1, two per lane and batch) in `words` and flags the group at which
// every word starts (as mark_pairs).  Returns false when the range is not exactly this segment: the words must add up
// to nvalid groups, none of them empty -- then the r-th flag is the r-th word.
__device__ __forceinline__ bool seg_mark(const SegRange &rg, const u32 (&x0)[kSegBatches], const u32 (&x1)[kSegBatches],
                                         unsigned char *flag, u32 *words, u32 lane) {
    const u32 cnt = rg.cnt, nvalid = rg.nvalid;
    reinterpret_cast<uint4 *>(flag)[lane] = make_uint4(0, 0, 0, 0); // 64 lanes x 16 B = the 1024 flags
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const u32 fbase = (u32)(uintptr_t)(lds_u8_ptr)flag;
    const u32 wbase = (u32)(uintptr_t)(lds_u8_ptr)reinterpret_cast<unsigned char *>(words);
    u32 pos = 0; // groups covered by the batches so far
    bool empty_word = false;
#pragma unroll
    for (int b = 0; b < kSegBatches; ++b) {
        const u32 wi = 128u * b;
        if (wi < cnt) { // wave-uniform
            // a lane without a word stores its flag byte into its own word slot instead, which is past the segment's
            // words and never read (no dump area: 20 KiB of LDS per workgroup, eight workgroups per CU)
            const u32 dump = wbase + (wi + 2u * lane) * 4u;
            const u32 i0 = wi + 2u * lane;
            const bool in0 = i0 < cnt, in1 = i0 + 1u < cnt;
            reinterpret_cast<uint2 *>(words)[64 * b + (int)lane] = make_uint2(x0[b], x1[b]);
            // counts are clamped so that a corrupt word cannot wrap the 32-bit sums; anything above 1024 fails the total
            const u32 n0 = in0 ? min(word_groups(x0[b]), 2u * kSegGroups) : 0u, n1 = in1 ? min(word_groups(x1[b]), 2u * kSegGroups) : 0u;
            empty_word |= (in0 && n0 == 0u) || (in1 && n1 == 0u);
            // a full batch of literals (dense data): consecutive positions, no scan
            const u32 incl = (wi + 128u <= cnt && __ballot((int)(x0[b] | x1[b]) < 0) == 0) ? 2u * lane + 2u : wave_scan_incl32(n0 + n1);
            const u32 lo1 = pos + incl - n1, lo0 = lo1 - n0;
            const bool c0 = in0 && lo0 < nvalid, c1 = in1 && lo1 < nvalid;
            const u32 a0 = (u32)__mul24((int)(lo0 >> 6), -1023) + ((lo0 << 4) + fbase); // flag_slot(lo0), three instructions
            const u32 a1 = (u32)__mul24((int)(lo1 >> 6), -1023) + ((lo1 << 4) + fbase);
            *(lds_u8_ptr)(uintptr_t)(c0 ? a0 : dump) = 1;
            *(lds_u8_ptr)(uintptr_t)(c1 ? a1 : dump + 4u) = 1;
            pos += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return !(rg.bad || pos != nvalid || __ballot(empty_word) != 0);
}

// Step s of the expansion (expand_steps()): the 31-bit group 64 s + lane of a marked segment.  f: the lane's 16 flag
// bytes; before: flags in earlier steps - 1 (carried from step to step).
__device__ __forceinline__ u32 seg_group(int s, const u32 (&f)[4], u32 &before, const u32 *words, u32 cnt, u32 nvalid, u32 lane) {
    const u32 fb = (f[s >> 2] >> (8 * (s & 3))) & 0xFFu;
    const u64 m = __ballot(fb != 0u);
    // flags at or below my lane = bits of (m >> 1) below my lane (v_mbcnt) + bit 0; the running count is a scalar
    const u64 m1 = m >> 1;
    const u32 r = __builtin_amdgcn_mbcnt_hi((u32)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m1, 0u)) + (before + ((u32)m & 1u));
    before += (u32)__builtin_popcountll(m);
    const u32 src_word = words[min(r, cnt - 1u)];
    const u32 fill_val = (u32)__builtin_amdgcn_sbfe((int)src_word, 30, 1) >> 1; // 31 copies of bit 30
    u32 grp = (int)src_word < 0 ? fill_val : src_word;
    if ((u32)(64 * s) + lane >= nvalid) grp = 0u;
    return grp;
}


} // namespace
} // namespace wah

#endif // WAH_SEGDECODE_HPP_
