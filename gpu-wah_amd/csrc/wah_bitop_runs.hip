// wah_bitop_runs.hip -- bit operations on indexed compressed bitmaps IN THE COMPRESSED DOMAIN (the
// reference has no counterpart: its README.md:10 names such operations as what bitmap indexes use WAH for).
//
// wah_bitop_indexed_device / wah_bitop_many_indexed_device decode every operand's segment into its 1024 groups, combine
// them and run the compress passes over the result: about 1 300 instructions per output segment whatever the operands hold --
// two clustered operands of 17 MB each took 0.53 ms, 0.012 of the roofline.  For operands of FEW words per segment this file
// merges their run lists instead, the way WAH operations are done on a CPU, but with one LANE per segment (segments of a
// stream of compress() are independent: no fill crosses one; the index says where each starts):
//   a lane walks its segment in all operands at once; a step takes the shortest of the operands' current runs (a literal is
//   a run of one group), combines the runs' group values, and appends the result to the segment's output with the encoder's
//   rules -- zero / one groups extend or start a fill, anything else is a literal (kernels.cu:96-142 says the same of a
//   bitmap's groups).  The result is word for word what compress() gives for the combined bitmap.
// Output sizes are not known in advance, but bounded: a segment's result has at most as many words as the operands have in
// that segment.  Pass 1 writes every segment's words to a temporary at the sum of the operands' offsets of the segment and
// counts them (seg_count, one total per tile of segments), one workgroup scans the tile totals, pass 2 moves the segments'
// words together and writes the result's index.  Nothing of bitmap size is read or written.
// A tile's words of all operands are staged in LDS by coalesced loads when they fit (kRunsLdsWords per workgroup); a tile
// for which they do not (a dense stretch inside a sparse bitmap) reads them from global memory, lane by lane: slower, as
// correct.
#include "wah_device.hpp"
#include "wah_internal.hpp"

namespace wah {
namespace {

#ifndef WAH_RUNS_LDS_WORDS
#define WAH_RUNS_LDS_WORDS 10240
#endif
// staged words per workgroup at most, all operands together (40 KB: three workgroups per CU; a launch asks for one and a half
// times its average tile, launch_runs_k) -- a workgroup of 256, 128 or 64 segments, by what the operands hold on average: up
// to 36, 72 or more words per segment.  Reading the words from global memory lane by lane instead costs a 128-byte line per
// word (one bit in 2^11, two operands, 63 words per segment: 0.55 ms without the image, 0.15 with).
constexpr u32 kRunsLdsWords = WAH_RUNS_LDS_WORDS;

// the combination of two group values (31 bits) by the minterms of `acc op operand` (include/wah.h: WAH_OP_AND 0, OR 1, XOR 2,
// ANDNOT 3: A and not B and not C ...): three masks, wave-uniform, made once -- no branch on the operation inside a step
struct RunsOp {
    u32 ab, a_nb, na_b;
};
__device__ __forceinline__ RunsOp runs_op(u32 op) {
    RunsOp m;
    m.ab = op <= 1u ? ~0u : 0u;
    m.a_nb = op == 0u ? 0u : ~0u;
    m.na_b = op == 1u || op == 2u ? ~0u : 0u;
    return m;
}
__device__ __forceinline__ u32 runs_combine(u32 r, u32 v, const RunsOp &m) {
    return ((r & v & m.ab) | (r & ~v & m.a_nb) | (~r & v & m.na_b)) & kOnes31;
}

// One segment of K operands merged by one lane, WITHOUT branches inside a step (the lanes of a wave are at different places
// of different segments: every branch would be taken both ways).  word(j, i): word i of operand j, i in [begin[j], end[j]).
// The result's words go to o[0 ..) -- at most as many as the operands have words in the segment (every step ends a word of
// at least one operand, and every output word begins with a step of its own), + K for operands that are not what the index
// says.  Returns their number.  bad: an operand's words do not make up exactly `nvalid` groups, or a fill of count 0.
template <int K, typename Load>
__device__ __forceinline__ u32 runs_merge(const Load &word, const u32 (&begin)[K], const u32 (&end)[K], u32 nvalid, u32 op, const __amdgpu_buffer_rsrc_t o,
                                          u32 o_at, bool &bad) {
    u32 idx[K], rem[K], val[K];
#pragma unroll
    for (int j = 0; j < K; ++j) idx[j] = begin[j], rem[j] = 0u, val[j] = 0u;
    const RunsOp mop = runs_op(op);
    u32 left = nvalid, n_out = 0u;
    // the word that is being built -- a fill (its count grows in place) or a literal -- goes out when the next one begins: one
    // store per step at most.  ptype: 0 none yet / a literal, 1 a fill of zeros, 2 of ones
    u32 ptype = 0u, pending = 0u;
    bool have = false;
    while (left != 0u) {
        u32 step = left;
        // (all operands' words asked for before the first is looked at; read whether needed or not, from an index inside the
        //  operand's words of this segment -- its first word's place when it has run out: the caller has refused empty ranges)
        u32 ww[K];
#pragma unroll
        for (int j = 0; j < K; ++j) ww[j] = word(j, idx[j] == end[j] ? begin[j] : idx[j]);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool need = rem[j] == 0u;
            const bool out_of_words = idx[j] == end[j]; // (then the segment is not what the index says)
            const u32 w = ww[j];
            const bool fill = (int)w < 0;
            u32 cnt = fill ? (w & kCountMask) : 1u;
            const u32 v = fill ? (u32)((int)(w << 1) >> 31) & kOnes31 : w; // 31 copies of bit 30
            bad |= need && (out_of_words || cnt == 0u);
            cnt = out_of_words ? left : cnt == 0u ? 1u : cnt;
            rem[j] = need ? cnt : rem[j];
            val[j] = need ? (out_of_words ? 0u : v) : val[j];
            idx[j] += need && !out_of_words ? 1u : 0u;
            step = rem[j] < step ? rem[j] : step;
        }
        u32 r = val[0];
#pragma unroll
        for (int j = 1; j < K; ++j) r = runs_combine(r, val[j], mop);
#pragma unroll
        for (int j = 0; j < K; ++j) rem[j] -= step;
        left -= step;
        const u32 type = r == 0u ? 1u : r == kOnes31 ? 2u : 0u;
        const bool same = type != 0u && type == ptype;
        if (!same && have) {
            __builtin_amdgcn_raw_buffer_store_b32(pending, o, (o_at + n_out) * 4u, 0, 0);
            ++n_out;
        }
        pending = same ? pending + step : type == 0u ? r : (type == 2u ? kFillOne : kFillZero) | step;
        ptype = type;
        have = true;
    }
    if (have) {
        __builtin_amdgcn_raw_buffer_store_b32(pending, o, (o_at + n_out) * 4u, 0, 0);
        ++n_out;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) bad |= rem[j] != 0u || idx[j] != end[j]; // (a fill that reaches past the segment; words left over)
    return n_out;
}

// Pass 1: kTileSegs segments per workgroup, one per thread: the segment's result to temp[sum of the operands' offsets of the
// segment ..) (room for as many words as the operands have there: see runs_merge), its length to seg_count, the tile's to
// tile_total.
// lds_words: the size of the LDS image this launch was given (dynamic: operands of very few words per segment get by with a
// small one, and more workgroups per CU); 0: none at all.
template <int K, int kTileSegs>
__global__ __launch_bounds__(kTileSegs) void bitop_runs_kernel(const BitopRunsArgs a, u32 lds_words) {
    extern __shared__ u32 s_words[];
    __shared__ u32 s_wave[kTileSegs / 64];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u64 seg0 = (u64)blockIdx.x * kTileSegs;
    const u64 seg1 = seg0 + kTileSegs < a.n_segments ? seg0 + kTileSegs : a.n_segments;
    const u64 seg = seg0 + tid;
    const bool active = seg < seg1;
    const u32 op = (u32)a.op;

    // ---- where the tile's words lie in every operand (wave-uniform), and whether they fit the LDS --------------------------------
    u64 r_lo[K], r_hi[K];
    u64 total = 0;
    bool region_ok = true;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        r_lo[j] = a.offs[j][seg0];
        r_hi[j] = a.offs[j][seg1];
        region_ok &= r_lo[j] <= r_hi[j] && r_hi[j] <= a.c_words[j];
        total += region_ok ? r_hi[j] - r_lo[j] : 0u;
    }
    const bool staged = region_ok && total <= lds_words && lds_words != 0u;
    // the tile's part of the temporary: [sum of r_lo, sum of r_hi) + K words (runs_merge)
    u64 tile_at = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) tile_at += region_ok ? r_lo[j] : 0u;
    const u64 tile_words = region_ok ? total + (u64)K : 0u;
    const __amdgpu_buffer_rsrc_t o_rsrc = make_rsrc(a.temp + tile_at, tile_words < 0x3FFFFFFFull ? (u32)tile_words * 4u : 0xFFFFFFFCu);
    u32 base[K]; // where operand j's words start in s_words
    if (staged) {
        u32 at = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            base[j] = at;
            const u32 len = (u32)(r_hi[j] - r_lo[j]);
            const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.comp[j] + r_lo[j], len * 4u);
#pragma nounroll
            for (u32 i = tid; i < len; i += kTileSegs) s_words[at + i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, i * 4u, 0, 0);
            at += len;
        }
        __syncthreads();
    }

    // ---- the lane's segment ------------------------------------------------------------------------------------------------------
    u32 n_out = 0;
    if (active) {
        bool bad = false;
        const u64 g0 = seg * kSegGroups;
        const u32 nvalid = a.groups - g0 < kSegGroups ? (u32)(a.groups - g0) : kSegGroups;
        u64 lo[K], hi[K], at = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            lo[j] = a.offs[j][seg];
            hi[j] = a.offs[j][seg + 1];
            // (every word of a stream of compress() holds at least one group)
            bad |= lo[j] > hi[j] || hi[j] > a.c_words[j] || hi[j] - lo[j] > nvalid || hi[j] == lo[j] || !region_ok || lo[j] < r_lo[j] || hi[j] > r_hi[j];
            at += lo[j];
        }
        if (!bad) {
            // (the tile's part of the temporary behind one descriptor: 32-bit offsets in the stores; what a corrupt operand would
            //  put behind it is dropped)
            const u32 o_at = (u32)(at - tile_at);
            u32 begin[K], end[K];
            if (staged) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    begin[j] = base[j] + (u32)(lo[j] - r_lo[j]);
                    end[j] = begin[j] + (u32)(hi[j] - lo[j]);
                }
                n_out = runs_merge<K>([&](int, u32 i) { return s_words[i]; }, begin, end, nvalid, op, o_rsrc, o_at, bad);
            } else {
                const u32 *p[K];
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    p[j] = a.comp[j] + lo[j];
                    begin[j] = 0u;
                    end[j] = (u32)(hi[j] - lo[j]);
                }
                n_out = runs_merge<K>([&](int j, u32 i) { return p[j][i]; }, begin, end, nvalid, op, o_rsrc, o_at, bad);
            }
        }
        if (bad) {
            n_out = 0u;
            atomicOr(a.ctrl + kCtlError, kErrStream);
        }
        a.seg_count[seg] = n_out;
    }
    const u32 t = wave_total32(n_out);
    if (lane == 0u) s_wave[wave] = t;
    __syncthreads();
    if (tid == 0u) {
        u64 sum = 0;
#pragma unroll
        for (int w = 0; w < kTileSegs / 64; ++w) sum += s_wave[w];
        a.tile_total[blockIdx.x] = sum;
    }
}

// tile_total[0 .. n_tiles) -> its exclusive scan, in place; the result's size, checked against the capacity
__global__ __launch_bounds__(1024) void bitop_runs_scan_kernel(const BitopRunsArgs a, u64 n_tiles) {
    __shared__ u64 s_sum[16];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u64 per = (n_tiles + 1023u) / 1024u;
    const u64 i0 = per * tid < n_tiles ? per * tid : n_tiles, i1 = i0 + per < n_tiles ? i0 + per : n_tiles;
    u64 mine = 0;
    for (u64 i = i0; i < i1; ++i) mine += a.tile_total[i];
    const u64 incl = wave_scan_incl(mine, lane);
    if (lane == 63u) s_sum[wave] = incl;
    __syncthreads();
    u64 before = incl - mine, all = 0;
    for (u32 w = 0; w < 16u; ++w) {
        const u64 s = s_sum[w];
        if (w < wave) before += s;
        all += s;
    }
    for (u64 i = i0; i < i1; ++i) {
        const u64 t = a.tile_total[i];
        a.tile_total[i] = before;
        before += t;
    }
    if (tid == 0u) {
        *a.out_words = all;
        if (a.out_offsets) a.out_offsets[a.n_segments] = all;
        if (all > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
    }
}

// Pass 2: the segments' words moved together (temp -> out), the result's index written.  A workgroup takes a tile's
// segments: their counts scanned, then output word q of the tile -- consecutive threads, consecutive words -- finds its segment
// by binary search in the scanned counts and its source behind that segment's place in temp.  Nothing is written when pass 1
// or the scan has refused.
template <int K, int kTileSegs>
__global__ __launch_bounds__(kTileSegs) void bitop_runs_place_kernel(const BitopRunsArgs a) {
    __shared__ u32 s_wave[kTileSegs / 64];
    __shared__ u32 s_refused;
    __shared__ u32 s_excl[kTileSegs + 1]; // words of the tile in front of segment i
    __shared__ u64 s_at[kTileSegs];       // where segment i's words lie in temp
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0u) s_refused = __hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_refused != 0u) return;
    const u64 seg = (u64)blockIdx.x * kTileSegs + tid;
    const bool active = seg < a.n_segments;
    const u32 c = active ? a.seg_count[seg] : 0u;
    const u32 incl = wave_scan_incl32(c);
    if (lane == 63u) s_wave[wave] = incl;
    __syncthreads();
    u32 before = incl - c;
#pragma unroll
    for (int w = 0; w < kTileSegs / 64; ++w)
        if ((u32)w < wave) before += s_wave[w];
    const u64 tile_off = a.tile_total[blockIdx.x];
    u64 at = 0;
    if (active) {
        if (a.out_offsets) a.out_offsets[seg] = tile_off + before;
#pragma unroll
        for (int j = 0; j < K; ++j) at += a.offs[j][seg];
    }
    s_excl[tid] = before;
    s_at[tid] = at;
    if (tid == kTileSegs - 1) s_excl[kTileSegs] = before + c;
    __syncthreads();
    const u32 tile_words = s_excl[kTileSegs];
    u32 *const dst = a.out + tile_off; // (the scan launch has checked the whole result against the capacity)
    for (u32 q = tid; q < tile_words; q += kTileSegs) {
        u32 lo = 0; // the last segment whose words start at or before q
#pragma unroll
        for (u32 h = kTileSegs / 2; h != 0u; h >>= 1)
            if (s_excl[lo + h] <= q) lo += h;
        dst[q] = a.temp[s_at[lo] + (q - s_excl[lo])];
    }
}

template <int K, int kTileSegs>
void launch_runs_k(const BitopRunsArgs &a, u64 n_tiles, hipStream_t s) {
    // (all operands together, per segment: more than the LDS image holds for most tiles -> the kernel without one)
    const u64 total = [&] {
        u64 t = 0;
        for (int j = 0; j < K; ++j) t += a.c_words[j];
        return t;
    }();
    // the image: one and a half times an average tile's words, in KiB steps, at most kRunsLdsWords; none when even that would
    // not hold an average tile
    u32 lds_words = 0;
    if (total * kTileSegs <= (u64)kRunsLdsWords * a.n_segments) {
        const u64 want = (total * kTileSegs * 3u / 2u / a.n_segments + 1024u) & ~(u64)255u;
        lds_words = (u32)(want < kRunsLdsWords ? want : kRunsLdsWords);
    }
    hipLaunchKernelGGL((bitop_runs_kernel<K, kTileSegs>), dim3((unsigned)n_tiles), dim3(kTileSegs), lds_words * sizeof(u32), s, a, lds_words);
    hipLaunchKernelGGL(bitop_runs_scan_kernel, dim3(1), dim3(1024), 0, s, a, n_tiles);
    hipLaunchKernelGGL((bitop_runs_place_kernel<K, kTileSegs>), dim3((unsigned)n_tiles), dim3(kTileSegs), 0, s, a);
}
template <int kTileSegs>
void launch_runs_tile(const BitopRunsArgs &a, u64 n_tiles, hipStream_t s) {
    switch (a.n) {
    case 1: launch_runs_k<1, kTileSegs>(a, n_tiles, s); break;
    case 2: launch_runs_k<2, kTileSegs>(a, n_tiles, s); break;
    case 3: launch_runs_k<3, kTileSegs>(a, n_tiles, s); break;
    case 4: launch_runs_k<4, kTileSegs>(a, n_tiles, s); break;
    case 5: launch_runs_k<5, kTileSegs>(a, n_tiles, s); break;
    case 6: launch_runs_k<6, kTileSegs>(a, n_tiles, s); break;
    case 7: launch_runs_k<7, kTileSegs>(a, n_tiles, s); break;
    default: launch_runs_k<8, kTileSegs>(a, n_tiles, s); break;
    }
}

} // namespace

hipError_t launch_bitop_runs(const BitopRunsArgs &a, hipStream_t s) {
    if (a.n_segments == 0) { // an empty bitmap: an empty stream
        hipLaunchKernelGGL(bitop_runs_scan_kernel, dim3(1), dim3(1024), 0, s, a, (u64)0);
        return hipGetLastError();
    }
    // segments per workgroup: so that an average tile's words fit the image with a tenth to spare
    u64 total = 0;
    for (int j = 0; j < a.n; ++j) total += a.c_words[j];
    const u64 fit = (u64)kRunsLdsWords * a.n_segments * 9u / 10u;
    if (total * 256u <= fit) launch_runs_tile<256>(a, (a.n_segments + 255u) / 256u, s);
    else if (total * 128u <= fit) launch_runs_tile<128>(a, (a.n_segments + 127u) / 128u, s);
    else launch_runs_tile<64>(a, (a.n_segments + 63u) / 64u, s);
    return hipGetLastError();
}

} // namespace wah
