p='gpu-wah_amd/csrc/wah_kernels.hip'
s=open(p).read()

# --- slow routine: dynamic tail + unified clipping
old='''    const u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
    const u64 c = lane < kCoarse ? s_coarse[lane] : ~0ull;
    const u32 bucket = (u32)__popcll(__ballot(c <= target)) - 1u;
    const u64 drop = target - uniform64(s_coarse[bucket]); // groups of the bucket in front of the segment
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<uint4 *>(src)[k * 64 + (int)lane] = make_uint4(0, 0, 0, 0);'''
new='''    u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
    const u64 c = lane < kCoarse ? s_coarse[lane] : ~0ull;
    const u32 bucket = (u32)__popcll(__ballot(c <= target)) - 1u;
    const u64 drop = target - uniform64(s_coarse[bucket]); // groups of the bucket in front of the segment
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<uint4 *>(src)[k * 64 + (int)lane] = make_uint4(0, 0, 0, 0);'''
assert old in s
s=s.replace(old,new,1)
old='''    if (seen < drop + nvalid) { // the stream ended inside the segment: cannot happen for a consistent scan
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    const u64 seg_w0 = seg * kSegWords;
    const u32 seg_words = out_words > seg_w0 ? (u32)(out_words - seg_w0 < kSegWords ? out_words - seg_w0 : kSegWords) : 0u;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + seg_w0, seg_words * 4u); // stores past the end are dropped
    const u32 o = lane & 31u;'''
new='''    if (seen < drop + nvalid) {
        if (!a.dynamic_tail) { // the stream ended inside the segment: cannot happen for a consistent scan
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
            return;
        }
        nvalid = (u32)(seen - drop); // single pass: this is the last segment of the stream, and that is its length
    }
    const u64 seg_w0 = seg * kSegWords;
    const u32 full = nvalid == kSegGroups ? kSegWords : (31u * nvalid + 31u) / 32u;
    const u32 seg_words = out_words > seg_w0 ? (u32)(out_words - seg_w0 < full ? out_words - seg_w0 : full) : 0u;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + seg_w0, seg_words * 4u); // stores past the end are dropped
    const u32 o = lane & 31u;'''
assert old in s
s=s.replace(old,new,1)

# --- mark_pairs: report empty words met in the spill region
old='''template <bool kLocal, bool kFirst>
__device__ __forceinline__ u32 mark_pairs(const ExpandArgs &a, const u32 *s_words, unsigned char *flag, u64 tile_w0,
                                          u32 left_in_stream, u32 nvalid, u32 lane, int &rel, u32 &wi) {'''
new='''template <bool kLocal, bool kFirst>
__device__ __forceinline__ u32 mark_pairs(const ExpandArgs &a, const u32 *s_words, unsigned char *flag, u64 tile_w0,
                                          u32 left_in_stream, u32 nvalid, u32 lane, int &rel, u32 &wi, bool &spill_empty) {'''
assert old in s
s=s.replace(old,new,1)
old='''    const u32 n0 = word_groups(w0), n1 = word_groups(w1);
    // all literals (dense data): consecutive positions, no scan; otherwise one DPP scan over the pair sums'''
new='''    const u32 n0 = word_groups(w0), n1 = word_groups(w1);
    // words past the tile have not been checked for empty fills by anybody (single pass): say so
    if (!kLocal) spill_empty |= __ballot((i0 < left_in_stream && n0 == 0u) || (i0 + 1u < left_in_stream && n1 == 0u)) != 0;
    // all literals (dense data): consecutive positions, no scan; otherwise one DPP scan over the pair sums'''
assert old in s
s=s.replace(old,new,1)
for tf in ("<true, true>","<false, true>","<true, false>","<false, false>"):
    s=s.replace(f"mark_pairs{tf}(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi)",f"mark_pairs{tf}(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi, spill_empty)")

# --- tame: returns false when the segment must be redone by the routine for streams with empties
old='''__device__ __forceinline__ void expand_segment_tame(const ExpandArgs &a, const u32 *s_words, const u32 *s_coarse32,'''
new='''__device__ __forceinline__ bool expand_segment_tame(const ExpandArgs &a, const u32 *s_words, const u32 *s_coarse32,'''
assert old in s
s=s.replace(old,new,1)
old='''    u32 wi = bucket * 64u;
    constexpr u32 kLastLocal = (u32)kScanTileWords - 128u; // batches starting up to here come out of the LDS tile'''
new='''    u32 wi = bucket * 64u;
    bool spill_empty = false;
    constexpr u32 kLastLocal = (u32)kScanTileWords - 128u; // batches starting up to here come out of the LDS tile'''
assert old in s
s=s.replace(old,new,1)
old='''    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (rel < (int)nvalid) {
        if (!a.dynamic_tail) { // the stream ended inside the segment: cannot happen for a consistent scan
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
            return;
        }
        nvalid = (u32)rel; // single pass: this is the last segment of the stream, and that is its length
    }
    // group g belongs to the r-th contributing word, r = (flags at positions <= g) - 1
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, wi <= (u32)kScanTileWords, lane);
}'''
new='''    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (a.dynamic_tail && spill_empty) return false; // empty fill words behind the tile: not for the rank arithmetic
    if (rel < (int)nvalid) {
        if (!a.dynamic_tail) { // the stream ended inside the segment: cannot happen for a consistent scan
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
            return true;
        }
        nvalid = (u32)rel; // single pass: this is the last segment of the stream, and that is its length
    }
    // group g belongs to the r-th contributing word, r = (flags at positions <= g) - 1
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, wi <= (u32)kScanTileWords, lane);
    return true;
}'''
assert old in s
s=s.replace(old,new,1)

# --- general: same
old='''__device__ __forceinline__ void expand_segment_general(const ExpandArgs &a, const u32 *s_words, const u64 *s_coarse,'''
new='''__device__ __forceinline__ bool expand_segment_general(const ExpandArgs &a, const u32 *s_words, const u64 *s_coarse,'''
assert old in s
s=s.replace(old,new,1)
old='''        const u32 ww = in ? tile_word(s_words, a, tile_w0, idx) : 0u;
        const u32 n = in ? word_groups(ww) : 0u;
        bool contributes;'''
new='''        const u32 ww = in ? tile_word(s_words, a, tile_w0, idx) : 0u;
        const u32 n = in ? word_groups(ww) : 0u;
        if (a.dynamic_tail && __ballot(in && idx >= (u32)kScanTileWords && n == 0u) != 0) return false; // see expand_segment_tame
        bool contributes;'''
assert old in s
s=s.replace(old,new,1)
old='''    if (seen < drop + nvalid) {
        if (!a.dynamic_tail) { // the stream ended inside the segment: cannot happen for a consistent scan
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
            return;
        }
        nvalid = (u32)(seen - drop); // single pass: this is the last segment of the stream, and that is its length
    }
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, false, lane);
}'''
new='''    if (seen < drop + nvalid) {
        if (!a.dynamic_tail) { // the stream ended inside the segment: cannot happen for a consistent scan
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
            return true;
        }
        nvalid = (u32)(seen - drop); // single pass: this is the last segment of the stream, and that is its length
    }
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, false, lane);
    return true;
}'''
assert old in s
s=s.replace(old,new,1)
# two-pass kernel call sites ignore the result
s=s.replace("            expand_segment_tame(a, s_words, s_coarse32, flag, tile_w0, (u32)(seg * kSegGroups - base), nvalid, out_words, seg, lane);","            (void)expand_segment_tame(a, s_words, s_coarse32, flag, tile_w0, (u32)(seg * kSegGroups - base), nvalid, out_words, seg, lane);")
s=s.replace("            expand_segment_general(a, s_words, s_coarse, flag, tile_w0, base, groups, out_words, seg, lane);","            (void)expand_segment_general(a, s_words, s_coarse, flag, tile_w0, base, groups, out_words, seg, lane);")
s=s.replace("    hipLaunchKernelGGL(decode_expand_kernel, dim3((unsigned)(n_tiles * parts)), dim3(kExpandThreads), 0, s, a);","    a.dynamic_tail = 0;\n    hipLaunchKernelGGL(decode_expand_kernel, dim3((unsigned)(n_tiles * parts)), dim3(kExpandThreads), 0, s, a);")
open(p,'w').write(s)
