// wah_aux.hip -- around the path: stream checker, bench generators / copy, workspace clearing, the merge pass of
// wah_merge_fills_device (the shared wavefront helpers are in wah_device.hpp)
#include "wah_device.hpp"

#include "../../include/wah_gen.h"

namespace wah {
namespace {

// The sixteen consecutive words of every thread out of the workgroup's 4096-word tile (checker, index builder, merger: each
// walks its words in stream order with a running group position).  Read as they lie -- sixteen dword loads per thread, every
// instruction touching sixty-four addresses 64 bytes apart -- the walk kernels ran at 1.9 TB/s (profiles/r04_next_rows.txt,
// first run); here the tile comes in by coalesced 16-byte loads (four per thread), is parked in LDS, and every thread takes
// its sixteen words from there with four 16-byte LDS reads.  `prev` = the word in front of the thread's first one (0 in
// front of the stream).  Streams that are only 4-byte aligned and the stream's last tile: dword loads, bounds checked;
// words behind the stream's end read as fills of count 0 (nothing).
constexpr u32 kWalkWords = (u32)kExpandWordsPerThread;
__device__ __forceinline__ void walk_load_tile(const u32 *comp, u64 c_words, u32 tile, u32 *s_tile /* kScanTileWords */, u32 (&w)[kWalkWords],
                                               u32 &prev) {
    const u64 tile_w0 = (u64)tile * kScanTileWords;
    if ((reinterpret_cast<uintptr_t>(comp) & 15u) == 0 && tile_w0 + kScanTileWords <= c_words) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(comp + tile_w0);
        u32x4 v[kWalkWords / 4];
#pragma unroll
        for (u32 k = 0; k < kWalkWords / 4; ++k) v[k] = __builtin_nontemporal_load(src + k * kExpandThreads + threadIdx.x);
#pragma unroll
        for (u32 k = 0; k < kWalkWords / 4; ++k) reinterpret_cast<u32x4 *>(s_tile)[k * kExpandThreads + threadIdx.x] = v[k];
    } else {
        for (u32 i = threadIdx.x; i < (u32)kScanTileWords; i += kExpandThreads) s_tile[i] = tile_w0 + i < c_words ? comp[tile_w0 + i] : 0x80000000u;
    }
    __syncthreads();
    const uint4 *mine = reinterpret_cast<const uint4 *>(s_tile + threadIdx.x * kWalkWords);
#pragma unroll
    for (u32 k = 0; k < kWalkWords / 4; ++k) {
        const uint4 q = mine[k];
        w[4 * k] = q.x, w[4 * k + 1] = q.y, w[4 * k + 2] = q.z, w[4 * k + 3] = q.w;
    }
    prev = threadIdx.x ? s_tile[threadIdx.x * kWalkWords - 1u] : (tile_w0 > 0 && tile_w0 - 1 < c_words ? comp[tile_w0 - 1] : 0u);
}

// ===========================================================================
// stream checker (include/wah.h: wah_validate_device).  One workgroup per 4096-word tile, after the sums pass: the
// tile bases give every word its group position, so the per-word properties that depend on position (a fill crossing
// a 1024-group boundary, two mergeable fills inside one segment) can be told from the ones that do not.
// ===========================================================================
__global__ __launch_bounds__(kExpandThreads) void validate_kernel(const u32 *comp, u64 c_words, const u64 *tile_base,
                                                                  const u64 *info, u64 *report) {
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ __attribute__((aligned(16))) u32 s_tile[kScanTileWords];
    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const u32 tile = blockIdx.x;
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread; // my 16 consecutive words
    u32 w[kExpandWordsPerThread];
    u32 prev; // the word in front of mine (a literal 0 if none)
    walk_load_tile(comp, c_words, tile, s_tile, w, prev);
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) mine += word_groups(w[k]);
    const u64 incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_wave_sum[wave] = incl;
    __syncthreads();
    u64 p = tile_base[tile] + (incl - mine); // group position of my first word
    for (u32 k = 0; k < wave; ++k) p += s_wave_sum[k];

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

// ===========================================================================
// wah_build_index_device (include/wah.h): the segment index of a stream that came without one.  Same walk as the checker:
// every word learns its group position; a word that starts on a multiple of 1024 groups is the first word of that
// segment.  A fill across such a boundary, or an empty fill, means the stream has no segment index (kErrStream).
// ===========================================================================
__global__ __launch_bounds__(kExpandThreads) void index_kernel(const u32 *comp, u64 c_words, const u64 *tile_base, const u64 *info,
                                                               u64 *offsets, u64 capacity, u32 *ctrl) {
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ __attribute__((aligned(16))) u32 s_tile[kScanTileWords];
    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const u32 tile = blockIdx.x;
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread; // my 16 consecutive words
    u32 w[kExpandWordsPerThread];
    u32 prev_unused;
    walk_load_tile(comp, c_words, tile, s_tile, w, prev_unused);
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) mine += word_groups(w[k]);
    const u64 incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_wave_sum[wave] = incl;
    __syncthreads();
    u64 p = tile_base[tile] + (incl - mine); // group position of my first word
    for (u32 k = 0; k < wave; ++k) p += s_wave_sum[k];
    const u64 n_seg = (info[1] + kSegGroups - 1) / kSegGroups;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        if (w0 + k < c_words) {
            const u32 n = word_groups(w[k]);
            const u32 in_seg = (u32)(p & (kSegGroups - 1u));
            bad |= n == 0u || in_seg + n > kSegGroups;
            if (in_seg == 0u && n != 0u && (p >> 10) < capacity) offsets[p >> 10] = w0 + k;
            p += n;
        }
    }
    if (__any(bad) && lane == 0) atomicOr(ctrl + kCtlError, kErrStream);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n_seg < capacity)
            offsets[n_seg] = c_words;
        else
            atomicOr(ctrl + kCtlError, kErrCapacity);
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

// (the bench's copy ceiling) ONE 16-byte load and store per thread, 4 KiB per workgroup, no loop: the fastest copy of the
// shapes measured on this chip (tools/copy_shapes_time.hip: 6.2 TB/s read + written; a grid-stride loop of 1024 workgroups,
// the ceiling of the earlier rounds, 5.3-5.7; four per thread 5.8)
__global__ __launch_bounds__(256) void copy_kernel(const uint4 *in, uint4 *out, u64 n16) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
    if (i < n16) out[i] = in[i];
}

} // namespace

hipError_t launch_validate(const u32 *comp, u64 c_words, const u64 *tile_base, const u64 *info, u64 *report, u64 n_tiles,
                           hipStream_t s) {
    hipLaunchKernelGGL(validate_init_kernel, dim3(1), dim3(64), 0, s, report);
    if (n_tiles) hipLaunchKernelGGL(validate_kernel, dim3((unsigned)n_tiles), dim3(kExpandThreads), 0, s, comp, c_words, tile_base, info, report);
    return hipGetLastError();
}

hipError_t launch_build_index(const u32 *comp, u64 c_words, const u64 *tile_base, const u64 *info, u64 *offsets, u64 capacity,
                              u32 *ctrl, u64 n_tiles, hipStream_t s) {
    if (n_tiles) hipLaunchKernelGGL(index_kernel, dim3((unsigned)n_tiles), dim3(kExpandThreads), 0, s, comp, c_words, tile_base, info, offsets, capacity, ctrl);
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

__device__ __forceinline__ void merge_load_tile(const MergeArgs &a, u32 tile, u64 *s_wave_sum, u32 *s_tile, u32 lane, u32 wave, MergeTile &t) {
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread;
    walk_load_tile(a.comp, a.c_words, tile, s_tile, t.w, t.prev);
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) mine += word_groups(t.w[k]);
    const u64 incl = wave_scan_incl(mine, lane);
    if (lane == 63) s_wave_sum[wave] = incl;
    __syncthreads();
    t.p = a.tile_base[tile] + (incl - mine);
    for (u32 k = 0; k < wave; ++k) t.p += s_wave_sum[k];
    t.have_prev = w0 > 0;
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

// u64 minimum over the wave (every lane gets it) and the minimum over the lanes BEHIND mine (~0 if none)
__device__ __forceinline__ u64 wave_min64(u64 v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 t = __shfl_xor(v, off);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ u64 wave_suffix_min_excl64(u64 v, u32 lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { // inclusive suffix minimum
        const u64 t = __shfl_down(v, off);
        if (lane + (u32)off < 64u && t < v) v = t;
    }
    const u64 next = __shfl_down(v, 1);
    return lane < 63u ? next : ~0ull;
}

// pass 1: per tile the number of kept words and the group position of its FIRST kept word (~0: none)
__global__ __launch_bounds__(kExpandThreads) void merge_count_kernel(const MergeArgs a) {
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ u32 s_kept[kExpandWaves];
    __shared__ u64 s_first[kExpandWaves];
    __shared__ __attribute__((aligned(16))) u32 s_tile[kScanTileWords];
    const u32 lane = lane_id(), wave = wave_id(), tile = blockIdx.x;
    MergeTile t;
    merge_load_tile(a, tile, s_wave_sum, s_tile, lane, wave, t);
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread;
    u32 kept = 0, prev = t.prev;
    bool have_prev = t.have_prev;
    u64 p = t.p, first = ~0ull;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        if (w0 + k < a.c_words && !merge_dropped(t.w[k], prev, have_prev, p)) {
            if (kept == 0) first = p;
            ++kept;
        }
        p += word_groups(t.w[k]);
        prev = t.w[k];
        have_prev = true;
    }
    const u32 wk = wave_sum32(kept);
    const u64 wf = wave_min64(first);
    if (lane == 0) {
        s_kept[wave] = wk;
        s_first[wave] = wf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.tile_kept[tile] = (u64)s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3];
        u64 f = s_first[0];
#pragma unroll
        for (u32 w = 1; w < kExpandWaves; ++w) f = s_first[w] < f ? s_first[w] : f;
        a.tile_first[tile] = f;
    }
}

// one workgroup: exclusive scan of tile_kept[0 .. n_tiles) in place, total into tile_kept[n_tiles] and *out_words; then, from
// the back, tile_first[t] := the position of the first kept word in any tile BEHIND t (the stream's group total behind the last
// one) -- what a tile's last kept fill runs up to
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
        s_carry = a.info[1]; // behind the last tile: the end of the stream
    }
    __syncthreads();
    const u64 chunks = (a.n_tiles + 1023) / 1024;
    for (u64 c = chunks; c-- > 0;) {
        const u64 i = c * 1024 + threadIdx.x;
        const u64 v = i < a.n_tiles ? a.tile_first[i] : ~0ull;
        u64 behind = wave_suffix_min_excl64(v, lane);      // ... in my wave
        const u64 wave_all = wave_min64(v);
        if (lane == 0) s_part[wave] = wave_all;
        __syncthreads();
        for (u32 k = wave + 1; k < 16; ++k) behind = s_part[k] < behind ? s_part[k] : behind; // ... in the waves behind mine
        const u64 carry = s_carry;                                                          // ... in the chunks behind this one
        if (i < a.n_tiles) a.tile_first[i] = behind < carry ? behind : carry;
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 m = carry;
            for (u32 k = 0; k < 16; ++k) m = s_part[k] < m ? s_part[k] : m;
            s_carry = m;
        }
        __syncthreads();
    }
}

// pass 2: every kept word to its place, a kept fill with its final count -- the distance to the next kept word, in the thread,
// behind it in the wave / the tile (suffix minima of the first kept positions) or in a later tile (tile_first, as left by the
// scan).  The tile's kept words are collected in LDS and leave as one coalesced run.  (Round 3: an 8-byte position per kept
// word written beside the words and read back by a fix-up launch: 1.26 ms for the sparse GiB's stream, 0.10 of the roofline.)
__global__ __launch_bounds__(kExpandThreads) void merge_scatter_kernel(const MergeArgs a) {
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ u32 s_kept[kExpandWaves];
    __shared__ u64 s_first[kExpandWaves];
    __shared__ __attribute__((aligned(16))) u32 s_tile[kScanTileWords];
    const u32 lane = lane_id(), wave = wave_id(), tile = blockIdx.x;
    MergeTile t;
    merge_load_tile(a, tile, s_wave_sum, s_tile, lane, wave, t);
    const u64 w0 = (u64)tile * kScanTileWords + (u64)threadIdx.x * kExpandWordsPerThread;
    u32 keep = 0; // bit k: word k is kept
    u64 pos[kExpandWordsPerThread];
    u32 kept = 0, prev = t.prev;
    bool have_prev = t.have_prev;
    u64 p = t.p, first = ~0ull;
#pragma unroll
    for (int k = 0; k < kExpandWordsPerThread; ++k) {
        pos[k] = p;
        if (w0 + k < a.c_words && !merge_dropped(t.w[k], prev, have_prev, p)) {
            if (kept == 0) first = p;
            keep |= 1u << k;
            ++kept;
        }
        p += word_groups(t.w[k]);
        prev = t.w[k];
        have_prev = true;
    }
    const u32 incl = wave_scan_incl32(kept);
    u64 next = wave_suffix_min_excl64(first, lane); // the first kept word behind my sixteen: in my wave ...
    const u64 wf = wave_min64(first);
    if (lane == 63) s_kept[wave] = incl;
    if (lane == 0) s_first[wave] = wf;
    __syncthreads(); // (every thread has taken its words out of s_tile: merge_load_tile ends with a barrier, and this is the next one)
    for (u32 k = wave + 1; k < kExpandWaves; ++k) next = s_first[k] < next ? s_first[k] : next; // ... in the waves behind mine ...
    const u64 tile_next = a.tile_first[tile];                                                    // ... in the tiles behind this one
    next = next < tile_next ? next : tile_next;
    u32 idx = incl - kept; // tile-local place of my first kept word
    for (u32 k = 0; k < wave; ++k) idx += s_kept[k];
    u32 total = 0;
#pragma unroll
    for (u32 k = 0; k < kExpandWaves; ++k) total += s_kept[k];
    // from the back: the word behind a kept word is known when the kept word is written
    idx += kept;
    bool bad = false;
#pragma unroll
    for (int k = kExpandWordsPerThread - 1; k >= 0; --k) {
        if (keep & (1u << k)) {
            u32 x = t.w[k];
            if ((x & kFillZero) && (x & kCountMask)) {
                const u64 cnt = next - pos[k];
                bad |= cnt > kCountMask; // cannot happen: runs do not cross 2^29-group blocks
                x = (x & ~kCountMask) | ((u32)cnt & kCountMask);
            }
            s_tile[--idx] = x;
            next = pos[k];
        }
    }
    if (__any(bad) && lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
    __syncthreads();
    const u64 out0 = a.tile_kept[tile];
    for (u32 i = threadIdx.x; i < total; i += kExpandThreads)
        if (out0 + i < a.out_capacity) a.out[out0 + i] = s_tile[i];
}

hipError_t launch_merge_fills(const MergeArgs &a, hipStream_t s) {
    if (a.n_tiles == 0) {
        hipLaunchKernelGGL(clear_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<u32 *>(a.out_words), (u64)2);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(merge_count_kernel, dim3((unsigned)a.n_tiles), dim3(kExpandThreads), 0, s, a);
    hipLaunchKernelGGL(merge_scan_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(merge_scatter_kernel, dim3((unsigned)a.n_tiles), dim3(kExpandThreads), 0, s, a);
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
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint4 *>(in),
                       reinterpret_cast<uint4 *>(out), n16);
    return hipGetLastError();
}

} // namespace wah
