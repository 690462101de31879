// wah_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the WAH path.
//
// What the reference does in five kernels, two thrust scans and four blocking
// 8-byte D2H copies (compress.cu:129-166, decompress.cu:66-115, kernels.cu),
// is done here in
//   compress   : ONE persistent kernel  (read 4N, write 4C, nothing else)
//   decompress : scan kernel + expand kernel
// built on three CDNA4 idioms:
//   * a wavefront (64 lanes) owns a whole 1024-group segment; the segment's 992
//     words are staged once in wave-private LDS with 16-byte coalesced loads and
//     re-read as 31-bit groups by a funnel shift (v_alignbit) -- the regroup of
//     kernels.cu:72-79 without 1/32 idle lanes and without the shift-by-32;
//   * zero/ones classification produces 64-bit lane masks straight from v_cmp
//     (ballot), so run detection (kernels.cu:126-149), run lengths (:156-174)
//     and the cross-warp merge (:188-229) collapse into a few SCALAR mask
//     operations per 64 groups plus one mbcnt rank per lane;
//   * output offsets come from a single-pass decoupled look-back over per-tile
//     descriptors (one 8-byte {status,value} granule per tile, written and
//     polled with agent-scope relaxed atomics: correct across the 8 non-coherent
//     XCD L2s) instead of thrust::exclusive_scan + moveData
//     (compress.cu:133-166, kernels.cu:273-280).
// Tiles are handed out by sharded arrival tickets, so forward progress never
// depends on dispatch order or on all workgroups being co-resident, and every
// spin is bounded.
#include "wah_internal.hpp"

#include "../../include/wah_gen.h"

namespace wah {
namespace {

using u32 = uint32_t;
using u64 = uint64_t;

constexpr u32 kSegLdsWords = 1008;  // 992 + 1 look-ahead word, padded to a multiple of 16 bytes
constexpr u32 kMaxSpins = 1u << 21; // bounded look-back wait

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
// wave-uniform by construction; readfirstlane tells the compiler so (keeps masks and offsets in SGPRs)
__device__ __forceinline__ u32 wave_id() { return (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ u64 desc_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// Number of set bits of a wave-uniform mask below this lane (v_mbcnt pair).
__device__ __forceinline__ u32 rank_below(u64 m) {
    return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// ---------------------------------------------------------------------------
// Tile tickets.  A workgroup first draws an arrival ticket (its virtual id),
// which fixes its shard; it then draws tile numbers j from that shard's counter
// and processes tile j*kShards + shard.  Tiles of one shard are handed out in
// increasing order to running workgroups, and the first kShards arrivals cover
// all shards, so the lowest unfinished tile is always held by (or next in line
// for) a running workgroup: the look-back below cannot deadlock, whatever the
// dispatch order or residency.
// ---------------------------------------------------------------------------
__device__ __forceinline__ u32 draw_arrival(u32 *ctrl) {
    return __hip_atomic_fetch_add(ctrl + kCtlStart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 draw_tile(u32 *ctrl, u32 shard) {
    const u32 j = __hip_atomic_fetch_add(ctrl + kCtlShard0 + 16u * shard, 1u, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    return (u64)j * kShards + shard;
}

// ---------------------------------------------------------------------------
// Decoupled look-back, executed by one full wavefront.  Lane i inspects the
// descriptor of tile (idx - i): 64 predecessors per poll.  Returns the sum of
// the aggregates of all tiles < `tile`.  Descriptors are single 8-byte
// granules, so no fence is needed: the data IS the flag.
// ---------------------------------------------------------------------------
__device__ __forceinline__ u64 lookback_exclusive(const u64 *desc, u64 tile, u32 lane, u32 *ctrl) {
    u64 acc = 0;
    long long idx = (long long)tile - 1;
    u32 spins = 0;
    for (;;) {
        const long long mine = idx - (long long)lane;
        u64 d = kStatusPrefix; // virtual "prefix 0" in front of tile 0
        if (mine >= 0) d = desc_load(desc + mine);
        const u32 st = (u32)(d >> kStatusShift);
        const u64 invalid = __ballot(st == 0u);
        const u64 prefix = __ballot(st == 2u);
        if (prefix) {
            const u32 first = (u32)__ffsll((long long)prefix) - 1u; // nearest predecessor with a full prefix
            const u64 nearer = (1ull << first) - 1ull;
            if ((invalid & nearer) == 0) {
                acc += wave_sum(lane <= first ? (d & kValueMask) : 0ull);
                return acc;
            }
        } else if (invalid == 0) {
            acc += wave_sum(d & kValueMask);
            idx -= 64;
            spins = 0;
            continue;
        }
        if (++spins > kMaxSpins) {
            if (lane == 0) atomicOr(ctrl + kCtlError, kErrTimeout);
            return acc;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Publish this tile's aggregate, resolve its exclusive prefix, publish the
// inclusive prefix.  Whole wavefront; returns the exclusive prefix.
__device__ __forceinline__ u64 tile_prefix(u64 *desc, u64 tile, u64 aggregate, u32 lane, u32 *ctrl) {
    u64 excl = 0;
    if (tile == 0) {
        if (lane == 0) desc_store(desc, kStatusPrefix | aggregate);
    } else {
        if (lane == 0) desc_store(desc + tile, kStatusAggregate | aggregate);
        excl = lookback_exclusive(desc, tile, lane, ctrl);
        if (lane == 0) desc_store(desc + tile, kStatusPrefix | (excl + aggregate));
    }
    return excl;
}

// ===========================================================================
// compress
// ===========================================================================

// Stage one segment (992 words, zero padded past the end of the input) in LDS.
__device__ __forceinline__ void stage_segment(const CompressArgs &a, u64 seg, u32 *lds, u32 lane) {
    const u64 w0 = seg * kSegWords;
    if (a.aligned16 && w0 + kSegWords <= a.n_words) {
        // 3968 B = 248 x 16 B: four coalesced dwordx4 loads per lane, issued back to back
        const uint4 *src = reinterpret_cast<const uint4 *>(a.in + w0);
        uint4 *dst = reinterpret_cast<uint4 *>(lds);
        const uint4 v0 = src[lane];
        const uint4 v1 = src[lane + 64];
        const uint4 v2 = src[lane + 128];
        uint4 v3 = make_uint4(0, 0, 0, 0);
        if (lane < 56) v3 = src[lane + 192];
        dst[lane] = v0;
        dst[lane + 64] = v1;
        dst[lane + 128] = v2;
        if (lane < 56) dst[lane + 192] = v3;
    } else {
        for (u32 i = lane; i < kSegWords; i += 64) lds[i] = (w0 + i < a.n_words) ? a.in[w0 + i] : 0u;
    }
    if (lane == 0) lds[kSegWords] = 0u; // look-ahead word of the last group (masked out by the 31-bit mask)
    // the wave re-reads other lanes' words: order the LDS traffic at wavefront scope (no barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 8) void compress_kernel(const CompressArgs a) {
    __shared__ __attribute__((aligned(16))) u32 s_seg[WAVES][kSegLdsWords];
    __shared__ u32 s_count[WAVES];
    __shared__ u64 s_tile;
    __shared__ u64 s_base;

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const u64 lt = (1ull << lane) - 1ull; // lanes below me

    // regroup constants: group g = 64*step + lane starts at stream bit 31*g;
    // 64 groups = 1984 bits = 62 words exactly, so the in-word shift is fixed per lane
    const u32 q0 = (31u * lane) >> 5;
    const u32 r = (31u * lane) & 31u;

    if (threadIdx.x == 0) s_tile = draw_arrival(a.ctrl);
    __syncthreads();
    const u32 shard = (u32)s_tile % kShards;
    __syncthreads();

    for (;;) {
        if (threadIdx.x == 0) s_tile = draw_tile(a.ctrl, shard);
        __syncthreads();
        const u64 tile = s_tile;
        if (tile >= a.n_tiles) break;

        const u64 seg = tile * WAVES + wave;
        u32 x[kSteps];
        u64 ends[kSteps];
        u32 count = 0;

        if (seg < a.n_segments) {
            u32 *lds = s_seg[wave];
            stage_segment(a, seg, lds, lane);

            const u64 g0 = seg * kSegGroups;
            const u32 nvalid = (a.n_groups - g0 < kSegGroups) ? (u32)(a.n_groups - g0) : kSegGroups;

            // classify (kernels.cu:93-112): one v_cmp per kind gives the 64-lane mask directly.
            // run ends (kernels.cu:126-141 + the merge of :188-229): a group does NOT end a run iff it
            // and its successor inside the segment are the same kind of fill.  The successor masks of
            // the last valid group are zero, so every segment closes its last run (tests.cpp:166-172).
            // Step s-1 is finished as soon as step s has been classified (needs its bit 0 only).
            const u32 *sp = lds + q0;
            u64 zprev = 0, oprev = 0, vprev = 0;
#pragma unroll
            for (int s = 0; s < (int)kSteps; ++s) {
                const u32 lo = sp[62 * s];
                const u32 hi = sp[62 * s + 1];
                const u32 xv = __builtin_amdgcn_alignbit(hi, lo, r) & kOnes31;
                x[s] = xv;
                const int rem = (int)nvalid - 64 * s;
                const u64 valid = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
                const u64 z = __ballot(xv == 0u) & valid;
                const u64 o = __ballot(xv == kOnes31) & valid;
                if (s > 0) {
                    const u64 zn = (zprev >> 1) | (z << 63);
                    const u64 on = (oprev >> 1) | (o << 63);
                    ends[s - 1] = vprev & ~((zprev & zn) | (oprev & on));
                    count += (u32)__popcll(ends[s - 1]);
                }
                zprev = z;
                oprev = o;
                vprev = valid;
            }
            ends[kSteps - 1] = vprev & ~((zprev & (zprev >> 1)) | (oprev & (oprev >> 1)));
            count += (u32)__popcll(ends[kSteps - 1]);
        } else {
#pragma unroll
            for (int s = 0; s < (int)kSteps; ++s) {
                x[s] = 0;
                ends[s] = 0;
            }
        }

        if (lane == 0) s_count[wave] = count;
        __syncthreads();

        // one wavefront resolves the tile's output offset (replaces compress.cu:133-157)
        if (wave == 0) {
            const u32 mine = lane < (u32)WAVES ? s_count[lane] : 0u;
            const u64 aggregate = wave_sum32(mine);
            const u64 excl = tile_prefix(a.desc, tile, aggregate, lane, a.ctrl);
            if (lane == 0) {
                s_base = excl;
                if (tile == a.n_tiles - 1) {
                    *a.out_words = excl + aggregate;
                    if (a.seg_offsets) a.seg_offsets[a.n_segments] = excl + aggregate;
                }
                if (excl + aggregate > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
            }
        }
        __syncthreads();

        u64 base = s_base;
        for (u32 w = 0; w < wave; ++w) base += s_count[w];
        const bool fits = base + count <= a.out_capacity;

        if (seg < a.n_segments) {
            if (lane == 0 && a.seg_offsets) a.seg_offsets[seg] = base;
            if (fits) {
                // emit (kernels.cu:233-259): rank by mbcnt, run length = distance to the previous run end
                u32 *dst = a.out + base;
                u32 done = 0;
                int last_end = -1;
#pragma unroll
                for (int s = 0; s < (int)kSteps; ++s) {
                    const u64 e = ends[s];
                    if (e) {
                        const u32 xv = x[s];
                        const bool isz = xv == 0u, iso = xv == kOnes31;
                        const u64 below = e & lt;
                        u32 val = xv;
                        if (__ballot(isz || iso) & e) {
                            const int prev = below ? (64 * s + 63 - (int)__clzll((long long)below)) : last_end;
                            const u32 len = (u32)(64 * s + (int)lane - prev);
                            val = isz ? (kFillZero | len) : (iso ? (kFillOne | len) : xv);
                        }
                        if ((e >> lane) & 1ull) dst[done + rank_below(e)] = val;
                        done += (u32)__popcll(e);
                        last_end = 64 * s + 63 - (int)__clzll((long long)e);
                    }
                }
            }
        }
        // s_tile / s_count / s_base are rewritten only after the next iteration's first barrier
    }
}

// ===========================================================================
// decompress, pass 1: per-word group counts, their exclusive scan (look-back)
// and the segment index.  Replaces getCounts + thrust::exclusive_scan
// (kernels.cu:291-309, decompress.cu:72-82) without the 8-bytes-per-word
// counts array: only one 12-byte record per OUTPUT segment is written.
// ===========================================================================
__device__ __forceinline__ u32 word_groups(u32 w) {
    return (w & kFillZero) ? (w & kCountMask) : 1u; // kernels.cu:298-304
}

__global__ __launch_bounds__(kScanThreads) void decode_scan_kernel(const ScanArgs a) {
    __shared__ u64 s_wave_sum[kScanThreads / 64];
    __shared__ u64 s_tile;
    __shared__ u64 s_base;

    const u32 lane = lane_id();
    const u32 wave = wave_id();

    if (threadIdx.x == 0) s_tile = draw_arrival(a.ctrl);
    __syncthreads();
    const u32 shard = (u32)s_tile % kShards;
    __syncthreads();

    for (;;) {
        if (threadIdx.x == 0) s_tile = draw_tile(a.ctrl, shard);
        __syncthreads();
        const u64 tile = s_tile;
        if (tile >= a.n_tiles) break;

        // thread t owns 16 consecutive words: four 16-byte loads
        const u64 w0 = tile * kScanTileWords + (u64)threadIdx.x * kScanWordsPerThread;
        u32 w[kScanWordsPerThread];
        if (a.aligned16 && w0 + kScanWordsPerThread <= a.c_words) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + w0);
#pragma unroll
            for (int k = 0; k < kScanWordsPerThread / 4; ++k) {
                const uint4 v = src[k];
                w[4 * k + 0] = v.x;
                w[4 * k + 1] = v.y;
                w[4 * k + 2] = v.z;
                w[4 * k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kScanWordsPerThread; ++k) w[k] = (w0 + k < a.c_words) ? a.comp[w0 + k] : 0u;
        }
        u64 mine = 0;
#pragma unroll
        for (int k = 0; k < kScanWordsPerThread; ++k) mine += (w0 + k < a.c_words) ? word_groups(w[k]) : 0u;

        const u64 incl = wave_scan_incl(mine, lane);
        if (lane == 63) s_wave_sum[wave] = incl;
        __syncthreads();

        if (wave == 0) {
            const u64 ws = lane < kScanThreads / 64 ? s_wave_sum[lane] : 0ull;
            const u64 aggregate = wave_sum(ws);
            const u64 excl = tile_prefix(a.desc, tile, aggregate, lane, a.ctrl);
            if (lane == 0) {
                s_base = excl;
                if (tile == a.n_tiles - 1) {
                    const u64 groups = excl + aggregate;
                    a.info[1] = groups;
                    a.info[0] = (31ull * groups + 31ull) / 32ull; // decompress.cu:84-93
                }
            }
        }
        __syncthreads();

        u64 pos = s_base + (incl - mine);
        for (u32 k = 0; k < wave; ++k) pos += s_wave_sum[k];

        // segment index: for every 1024-group boundary that falls inside a word, record the word and
        // how many of its groups lie before the boundary.  Reference streams cut fills at boundaries
        // (kernels.cu:188-229), so there it is at most one record per word with skip = 0; foreign
        // streams with long fills (decoder accepts any 30-bit count, kernels.cu:334) take the loop.
#pragma unroll
        for (int k = 0; k < kScanWordsPerThread; ++k) {
            if (w0 + k < a.c_words) {
                const u64 n = word_groups(w[k]);
                u64 b = (pos + kSegGroups - 1) / kSegGroups;
                for (; b * kSegGroups < pos + n; ++b) {
                    if (b < a.seg_capacity) {
                        a.seg_word[b] = w0 + k;
                        a.seg_skip[b] = (u32)(b * kSegGroups - pos);
                    }
                }
                pos += n;
            }
        }
    }
}

// ===========================================================================
// decompress, pass 2: output-stationary expansion.  One wavefront produces one
// output segment (1024 groups -> 992 words), whatever mix of fills and literals
// feeds it: no per-thread serial fill loop (kernels.cu:346-348), no 4-byte-per-
// group intermediate (decompress.cu:97) and no separate mergeWords pass
// (kernels.cu:369-385) -- the 31->32 repack happens in registers.
// ===========================================================================
__global__ __launch_bounds__(kExpandWaves * 64) void decode_expand_kernel(const ExpandArgs a) {
    __shared__ u32 s_val[kExpandWaves][kSegGroups];      // word that starts a run at group p
    __shared__ u32 s_mark[kExpandWaves][kSegGroups / 32]; // bit p set: a word starts at group p

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const u64 groups = a.info[1];
    const u64 out_words = a.info[0];
    const u64 n_seg = (groups + kSegGroups - 1) / kSegGroups;
    const u64 seg = (u64)blockIdx.x * kExpandWaves + wave;
    if (seg >= n_seg || seg >= a.seg_capacity) return; // wave-uniform
    if (out_words > a.out_capacity) {
        if (lane == 0 && seg == 0) atomicOr(a.ctrl + kCtlError, kErrCapacity);
        return;
    }

    u32 *val = s_val[wave];
    u32 *mark = s_mark[wave];
    if (lane < kSegGroups / 32) mark[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
    u64 wi = a.seg_word[seg];
    u32 skip = a.seg_skip[seg];

    // scatter run starts: each lane takes one compressed word per round
    u32 filled = 0;
    while (filled < nvalid && wi < a.c_words) {
        const u64 i = wi + lane;
        u32 w = 0;
        u64 n = 0;
        if (i < a.c_words) {
            w = a.comp[i];
            n = word_groups(w);
            if (lane == 0) n = n > skip ? n - skip : 0u; // groups of the first word already emitted earlier
        }
        const u64 incl = wave_scan_incl(n, lane);
        const u64 p = (u64)filled + (incl - n);
        if (n != 0u && p < nvalid) {
            val[p] = w;
            atomicOr(&mark[p >> 5], 1u << (p & 31u));
        }
        const u64 total = (u64)filled + __shfl(incl, 63);
        filled = total > kSegGroups ? kSegGroups : (u32)total;
        wi += 64;
        skip = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (filled < nvalid) {
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }

    // every lane finds the run its group belongs to: nearest mark at or below the group
    const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const u64 out0 = seg * kSegWords;
    int last_mark = 0;
#pragma unroll 4
    for (int s = 0; s < (int)kSteps; ++s) {
        const u64 m = (u64)mark[2 * s] | ((u64)mark[2 * s + 1] << 32); // same address in all lanes: broadcast
        const u64 mine = m & le;
        const int src = mine ? (64 * s + 63 - (int)__clzll((long long)mine)) : last_mark;
        if (m) last_mark = 64 * s + 63 - (int)__clzll((long long)m);
        const u32 w = val[src];
        u32 grp = (w & kFillZero) ? ((w & 0x40000000u) ? kOnes31 : 0u) : w; // kernels.cu:332-354
        if ((u32)(64 * s) + lane >= nvalid) grp = 0u;

        // 31 -> 32 repack (mergeWords, kernels.cu:375): output word 62*s + l takes stream bits
        // [32*(62 s + l), +32) = groups 64 s + l + (l >= 31) and the next one, shifted by l mod 31
        const u32 g1 = __shfl_down(grp, 1);
        const u32 g2 = __shfl_down(grp, 2);
        const bool hiHalf = lane >= 31;
        const u32 a0 = hiHalf ? g1 : grp;
        const u32 a1 = hiHalf ? g2 : g1;
        const u32 o = hiHalf ? lane - 31 : lane;
        const u32 word = (a0 >> o) | (a1 << (31u - o));
        const u64 idx = out0 + 62u * s + lane;
        if (lane < 62 && idx < out_words) a.out[idx] = word;
    }
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

int compress_grid(u64 n_tiles) {
    return persistent_grid(reinterpret_cast<const void *>(&compress_kernel<kCompressWaves>), kCompressWaves * 64, n_tiles);
}

hipError_t launch_compress(const CompressArgs &a, int grid, hipStream_t s) {
    hipLaunchKernelGGL(compress_kernel<kCompressWaves>, dim3(grid), dim3(kCompressWaves * 64), 0, s, a);
    return hipGetLastError();
}

int decode_scan_grid(u64 n_tiles) {
    return persistent_grid(reinterpret_cast<const void *>(&decode_scan_kernel), kScanThreads, n_tiles);
}

hipError_t launch_decode_scan(const ScanArgs &a, int grid, hipStream_t s) {
    hipLaunchKernelGGL(decode_scan_kernel, dim3(grid), dim3(kScanThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_decode_expand(const ExpandArgs &a, u64 max_segments, hipStream_t s) {
    const u64 blocks = (max_segments + kExpandWaves - 1) / kExpandWaves;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(decode_expand_kernel, dim3((unsigned)blocks), dim3(kExpandWaves * 64), 0, s, a);
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
    hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const uint4 *>(in),
                       reinterpret_cast<uint4 *>(out), n16);
    return hipGetLastError();
}

} // namespace wah
