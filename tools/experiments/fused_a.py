p='gpu-wah_amd/csrc/wah_kernels.hip'
s=open(p).read()

# 1. expand_emit: unified clipping
old=s[s.index("    const bool whole = nvalid == kSegGroups && (seg + 1) * kSegWords <= out_words;   // wave-uniform"):s.index("    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + seg_w0, seg_words * 4u); // stores past the end are dropped")]
new='''    // words this segment contributes: 992, or ceil(31 nvalid / 32) for the last one of the stream; never past
    // `out_words` (the decoded length, or the capacity when the length is not known yet)
    const u64 seg_w0 = seg * kSegWords;
    const u32 full = nvalid == kSegGroups ? kSegWords : (31u * nvalid + 31u) / 32u;
    const u32 seg_words = out_words > seg_w0 ? (u32)(out_words - seg_w0 < full ? out_words - seg_w0 : full) : 0u;
    const bool whole = nvalid == kSegGroups && seg_words == kSegWords;   // wave-uniform
'''
s=s.replace(old,new)

# 2. general path: dynamic tail
s=s.replace('''    const u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
    // 64-ary search over the coarse prefix: last 64-word bucket that starts at or before the target
    const u64 c = lane < kCoarse ? s_coarse[lane] : ~0ull;''','''    u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
    // 64-ary search over the coarse prefix: last 64-word bucket that starts at or before the target
    const u64 c = lane < kCoarse ? s_coarse[lane] : ~0ull;''')
s=s.replace('''    if (seen < drop + nvalid) { // the stream ended inside the segment: cannot happen for a consistent scan
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, false, lane);''','''    if (seen < drop + nvalid) {
        if (!a.dynamic_tail) { // the stream ended inside the segment: cannot happen for a consistent scan
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
            return;
        }
        nvalid = (u32)(seen - drop); // single pass: this is the last segment of the stream, and that is its length
    }
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, false, lane);''')

# 3. tame path
s=s.replace('''                                                    unsigned char *flag, u64 tile_w0, u32 target, u32 nvalid,
                                                    u64 out_words, u64 seg, u32 lane) {''','''                                                    unsigned char *flag, u64 tile_w0, u32 target, u32 nvalid_in,
                                                    u64 out_words, u64 seg, u32 lane) {
    u32 nvalid = nvalid_in;''')
s=s.replace('''    if (rel < (int)nvalid) { // the stream ended inside the segment: cannot happen for a consistent scan
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    // group g belongs to the r-th contributing word''','''    if (rel < (int)nvalid) {
        if (!a.dynamic_tail) { // the stream ended inside the segment: cannot happen for a consistent scan
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
            return;
        }
        nvalid = (u32)rel; // single pass: this is the last segment of the stream, and that is its length
    }
    // group g belongs to the r-th contributing word''')

open(p,'w').write(s)
p='gpu-wah_amd/csrc/wah_internal.hpp'
s=open(p).read()
s=s.replace("    uint32_t parts; // workgroups that share one tile's output segments (set by the launcher)\n};","    uint32_t parts; // workgroups that share one tile's output segments (set by the launcher)\n    // single-pass decode only (launch_decode_stream):\n    uint32_t dynamic_tail; // the stream's length is not known up front: the last segment finds its own end\n    uint32_t n_tiles;\n    uint64_t *tile_desc;   // 8-byte granule per tile\n    uint64_t *group_desc;  // 2 x 8-byte granules per group of 64 tiles (aggregates, then inclusive prefixes)\n    uint64_t *info_out;    // [0] decoded words, [1] groups: written by the last tile\n};")
s=s.replace("hipError_t launch_decode_expand(const ExpandArgs &a, uint64_t n_tiles, hipStream_t s);","hipError_t launch_decode_expand(const ExpandArgs &a, uint64_t n_tiles, hipStream_t s);\nhipError_t launch_decode_stream(const ExpandArgs &a, uint64_t n_tiles, hipStream_t s);")
open(p,'w').write(s)
