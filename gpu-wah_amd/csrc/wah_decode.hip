// wah_decode.hip -- decompress: decode_sums_kernel (tile bases) and decode_expand_kernel (see wah_compress.hip for the
// conventions; the shared wavefront helpers are in wah_device.hpp)
#include "wah_device.hpp"
#include "wah_segdecode.hpp"

namespace wah {
namespace {

// ===========================================================================
// decompress
//
// The reference runs getCounts -> thrust::exclusive_scan over one u64 PER COMPRESSED WORD -> decompressWords (a
// serial fill loop per thread into a 4-byte-per-group intermediate) -> mergeWords (kernels.cu:291-385,
// decompress.cu:66-115).  Here:
//   pass 1  decode_sums_kernel   : streaming reduce.  Tiles of 4096 compressed words; per tile the number of 31-bit
//                                   groups it expands to, turned into exclusive tile bases by the same one-hop
//                                   row scan as compress.  Reads C once, writes 8 bytes per tile.
//   pass 2  decode_expand_kernel : one workgroup per tile, tile words resident in LDS.  A tile OWNS the output
//                                   segments (1024 groups -> 992 words) whose first group falls into it; each of its
//                                   wavefronts expands whole segments: group -> source word by RANK (mbcnt over a
//                                   1024-bit mask of word starts), fill / literal decode, 31 -> 32 repack in
//                                   registers with two DPP shifts, dense 248-byte stores.
// Any stream the reference decoder accepts is handled (arbitrary 30-bit counts, fills across segment boundaries).
// ===========================================================================
// ---- decode_sums_kernel ---------------------------------------------------------------------------------------------
// Short-lived workgroups, one per WORKGROUP TILE of kSumWaves expand tiles (8 x 4096 words = 128 KiB of the stream),
// workgroup tile = arrival order.  A wave sums the group counts (getCounts, kernels.cu:291-309) of one expand tile: all
// sixteen 16-byte loads of a lane are issued before the first is used.  The workgroup tile's total is published as ONE
// 8-byte granule {epoch:16, groups:48}; wave 0 then resolves the groups in front of the tile with the same one-hop ROW
// SCAN as the compress kernel (wah_compress.hip), here over 8-byte granules:
//   granule[r][0 .. i)  +  granule[r-1][0 .. 256)  +  slot[s][1 ..] of rows s0 .. r-2  +  slot[s][0]
// (rows of 256 workgroup tiles, superrows of 64 rows; a row's last tile publishes the row's slot, a superrow's last tile
// the next superrow's slot[0]) and writes the exclusive bases of its eight expand tiles.  Nothing is persistent: no
// residency census; the arrival ticket is the only shared counter; launch epochs (wah_device.hpp) instead of clearing;
// every wait is bounded.
// Totals saturate at 2^47 groups (a stream that claims more -- 500 TB of bitmap -- is reported as WAH_ERR_STREAM).
constexpr u32 kSumWaves = (u32)kSumTilesPerGroup;
constexpr u32 kSumRowTiles = 256;
constexpr u32 kSumSuperRows = 64;
constexpr u32 kSumSlotShift = 48;
constexpr u64 kSumValueMask = (1ull << kSumSlotShift) - 1ull;
constexpr u64 kSumSaturate = 1ull << 47;
static_assert(kSumScanBlockWords >= 2 * kSumSuperRows * kSumRowTiles + 2 * (kSumSuperRows + 1) && kSumScanSlotsAt == 2 * kSumSuperRows * kSumRowTiles,
              "scan block layout");

__device__ __forceinline__ u64 sat_add(u64 x, u64 y) {
    const u64 s = x + y;
    return s < kSumSaturate ? s : kSumSaturate;
}

struct SumScan {
    u32x4 a[2], b[2]; // 8-byte granules of my row (entries below me) and of the previous row: four per lane
    u64 c;            // slot of my superrow: lane 0 = groups in front of it, lane 1 + k = groups of its row k
};

// The descriptors are made of values that ARE the same in every lane (the tile's number and what follows from it); they
// are passed through readfirstlane all the same: where the compiler cannot prove it (seen in decode_tile_kernel, where the
// call sits inside `if (wave == 0)` of a large unrolled body) it wraps every load in a "waterfall" loop over the distinct
// descriptors, and a re-read issued for SOME lanes then came back with the other lanes' earlier values zeroed (ROCm 7.2;
// tools/dbg_decode_tile.py: the base of a first-generation tile of row 2 = the entries of lanes 30..61 only).
__device__ __forceinline__ void sum_scan_issue(u32 *block_, u32 row_in_super_, u32 idx_, u32 n_slots_, u32 lane, bool need_a, bool need_b,
                                               bool need_c, SumScan &p) {
    u32 *const block = reinterpret_cast<u32 *>(uniform64(reinterpret_cast<u64>(block_)));
    const u32 row_in_super = uniform32(row_in_super_), idx = uniform32(idx_), n_slots = uniform32(n_slots_);
    if (need_a) {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(block + (u64)row_in_super * kSumRowTiles * 2u, idx * 8u);
        p.a[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 32u, 0, kAuxSc1);
        p.a[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 32u + 16u, 0, kAuxSc1);
    }
    if (need_b) {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(block + (u64)(row_in_super - 1u) * kSumRowTiles * 2u, kSumRowTiles * 8u);
        p.b[0] = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 32u, 0, kAuxSc1);
        p.b[1] = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 32u + 16u, 0, kAuxSc1);
    }
    if (need_c) {
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(block + kSumScanSlotsAt, n_slots * 8u);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rc, lane * 8u, 0, kAuxSc1);
        p.c = ((u64)v.y << 32) | v.x;
    }
}

// Wave 0 of a workgroup tile of a scan kernel (decode_sums_kernel, decode_tile_kernel), once the tile's total is known:
// publishes it as ONE 8-byte granule {epoch:16, groups:48} and resolves the groups in front of the tile with the one-hop
// ROW SCAN described at decode_sums_kernel; the last tile of a row publishes the row's slot, the last tile of a superrow
// the next superrow's slot[0].  Returns the groups in front of the tile; `end` = that + total, both saturating at 2^47
// (overflow: reported by the caller as WAH_ERR_STREAM).  kPublish = false: the caller has published the granule already
// (sums_publish) and sweeps LATE -- the sweep of a tile that has other work to do first is the cheaper the later it goes out,
// as in compress_pair_kernel.
__device__ __forceinline__ void sums_publish(u32 *gen_desc, u32 wt, u32 epoch, u64 total) {
    const u32 row = wt / kSumRowTiles, idx = wt % kSumRowTiles, sup = row / kSumSuperRows, row0 = sup * kSumSuperRows;
    u64 *const my_row = reinterpret_cast<u64 *>(gen_desc + (u64)sup * kSumScanBlockWords) + (u64)(row - row0) * kSumRowTiles;
    __hip_atomic_store(my_row + idx, ((u64)epoch << kSumSlotShift) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool kPublish = true>
__device__ __forceinline__ u64 sums_resolve(u32 *ctrl, u32 *gen_desc, u32 wt, u32 epoch, u64 total, u32 lane, u64 &end) {
    const u32 row = wt / kSumRowTiles, idx = wt % kSumRowTiles, sup = row / kSumSuperRows, row0 = sup * kSumSuperRows;
    const bool has_prev = row > row0;
    const u32 n_slots = has_prev ? row - row0 : 1u;
    u32 *const block = gen_desc + (u64)sup * kSumScanBlockWords;
    u64 *const my_row = reinterpret_cast<u64 *>(block) + (u64)(row - row0) * kSumRowTiles;
    u64 *const slots = reinterpret_cast<u64 *>(block + kSumScanSlotsAt);
    if (kPublish && lane == 0) __hip_atomic_store(my_row + idx, ((u64)epoch << kSumSlotShift) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    SumScan poll = {};
    bool need_a = true, need_b = has_prev, need_c = true;
    sum_scan_issue(block, row - row0, idx, n_slots, lane, need_a, need_b, need_c, poll);
    u64 sum_a = 0, sum_b = 0, sum_c = 0;
    u32 spins = 0;
    auto granule = [](const u32x4 &q, int h) { return ((u64)(h ? q.w : q.y) << 32) | (h ? q.z : q.x); };
    for (;;) {
        u32 bad_a = 0, bad_b = 0;
        bool bad_c = false;
        u64 ba = 0, bb = 0, bc = 0;
        if (need_a) {
            u64 sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u64 gk = granule(poll.a[k >> 1], k & 1);
                if (4u * lane + k < idx && (u32)(gk >> kSumSlotShift) != epoch) bad_a |= 1u << k;
                sum += gk & kSumValueMask; // entries at and above my index lie behind the descriptor and read as zero
            }
            ba = __ballot(bad_a != 0u);
            if (ba == 0) {
                sum_a = uniform64(wave_sum(sum));
                need_a = false;
                if (idx == kSumRowTiles - 1u && lane == 0) // my row is complete with me: publish its total
                    __hip_atomic_store(slots + 1u + (row - row0), ((u64)epoch << kSumSlotShift) | sat_add(sum_a, total), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (need_b) {
            u64 sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u64 gk = granule(poll.b[k >> 1], k & 1);
                if ((u32)(gk >> kSumSlotShift) != epoch) bad_b |= 1u << k;
                sum += gk & kSumValueMask;
            }
            bb = __ballot(bad_b != 0u);
            if (bb == 0) {
                sum_b = uniform64(wave_sum(sum));
                need_b = false;
            }
        }
        if (need_c) {
            // slot 0 of superrow 0 is never written: nothing lies in front of the first tile
            const bool wanted = lane < n_slots && !(sup == 0u && lane == 0u);
            bad_c = wanted && (u32)(poll.c >> kSumSlotShift) != epoch;
            bc = __ballot(bad_c);
            if (bc == 0) {
                sum_c = uniform64(wave_sum(wanted ? poll.c & kSumValueMask : 0ull));
                need_c = false;
            }
        }
        if (!(need_a || need_b || need_c)) break;
        if (++spins > kMaxSpins) {
            if (lane == 0) atomicOr(ctrl + kCtlError, kErrTimeout);
            break;
        }
        // wait for the missing entry with the highest tile number (published last), then read the missing lanes again
        const u64 *target;
        if (need_a) {
            const u32 hl = 63u - (u32)__builtin_clzll(ba);
            const u32 km = (u32)__builtin_amdgcn_readlane((int)bad_a, (int)hl);
            target = my_row + 4u * hl + (31u - (u32)__builtin_clz(km));
        } else if (need_b) {
            const u32 hl = 63u - (u32)__builtin_clzll(bb);
            const u32 km = (u32)__builtin_amdgcn_readlane((int)bad_b, (int)hl);
            target = my_row - kSumRowTiles + 4u * hl + (31u - (u32)__builtin_clz(km));
        } else {
            target = slots + (63u - (u32)__builtin_clzll(bc));
        }
        bool timed_out = false;
        for (;;) {
            __builtin_amdgcn_s_sleep(8);
            if ((u32)(__hip_atomic_load(target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> kSumSlotShift) == epoch) break;
            if (++spins > kMaxSpins) {
                timed_out = true;
                break;
            }
        }
        if (timed_out) {
            if (lane == 0) atomicOr(ctrl + kCtlError, kErrTimeout);
            break;
        }
        // every lane reads again, not only those whose entries were missing: a re-read under a per-lane condition came back
        // with the OTHER lanes' earlier values gone in decode_tile_kernel (ROCm 7.2; tools/dbg_decode_tile.py)
        sum_scan_issue(block, row - row0, idx, n_slots, lane, need_a, need_b, need_c, poll);
    }
    const u64 base = sat_add(sat_add(sum_c, sum_b), sum_a);
    end = sat_add(base, total);
    if (lane == 0 && idx == kSumRowTiles - 1u && row - row0 == kSumSuperRows - 1u) // last tile of a superrow
        __hip_atomic_store(reinterpret_cast<u64 *>(gen_desc + (u64)(sup + 1u) * kSumScanBlockWords + kSumScanSlotsAt),
                           ((u64)epoch << kSumSlotShift) | end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return base;
}

// The counters of the list (wah_internal.hpp, kCtlDefer) at the end of a launch that may have appended -- one lane, after
// every workgroup of the launch has read which counters are the launch's: the other ones are zeroed for the next launch
// (nobody reads them any more), these are handed to whoever walks the list.
__device__ __forceinline__ void defer_counters_next(u32 *c) {
    const u32 seq = __hip_atomic_load(c + kDeferSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<u64 *>(c) + ((seq + 1u) & 1u), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(c + kDeferSeq, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the counter pair {entries:32, parts:32} a launch that may append counts in (read by every workgroup at its start)
__device__ __forceinline__ u64 *defer_counters_mine(u32 *c) {
    return reinterpret_cast<u64 *>(c) + (uniform32(__hip_atomic_load(c + kDeferSeq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u);
}

// One expand tile onto the list of tiles that are decoded by workgroups of their own (the one-pass decoder's second launch,
// the launch behind decode_expand_kernel).  An entry = {tile, parts} {sum of the parts of the entries before it}: the tile's
// output segments are shared out over `parts` WORK ITEMS of that launch, kDeferPartSegs segments each (a long fill inside
// otherwise incompressible data: millions of groups out of one word; a highly compressed stream: every tile) -- so the list
// holds ONE entry per tile whatever the size of the output.  Slot and sum come out of ONE 64-bit atomic, so the entries
// are SORTED by their sums: work item k of the launch finds its entry by search (expand_list).  c = the launch's counter
// pair (defer_counters_mine).  false: the list is full (the tile is not on it).
__device__ __forceinline__ u32 dt_defer_parts(u64 groups) {
    const u64 segs = groups / kSegGroups + 2ull;
    // (up to one and a half items' worth: ONE item -- an item's fixed costs, finding its entry and staging its words, are a
    //  quarter of a 16-segment item's life: one bit in 2^9, 35 segments per tile, 0.41 ms in two items per tile)
    if (segs <= kDeferPartSegs + kDeferPartSegs / 2) return 1u;
    return segs >= (u64)kDeferPartSegs * 65536ull ? 65536u : (u32)((segs + kDeferPartSegs - 1ull) / kDeferPartSegs);
}
// Several tiles at once (those of tile[] whose bit is set in `valid`; the arrays are indexed with constants only: they stay in
// registers): ONE atomic for all of them -- every appending workgroup's atomic goes to the same address, and a launch is served
// about 86 of those per microsecond (tools/ticket_rate.hip); a highly compressed stream appends every tile
template <u32 kMany>
__device__ __forceinline__ bool dt_defer_many(u64 *c, u64 *list, u32 capacity, u32 valid, const u64 (&tile)[kMany], const u64 (&groups)[kMany], bool buckets) {
    if (valid == 0u) return true;
    u32 parts[kMany], sum = 0;
#pragma unroll
    for (u32 i = 0; i < kMany; ++i) {
        parts[i] = (valid >> i) & 1u ? dt_defer_parts(groups[i]) : 0u;
        sum += parts[i];
    }
    const u32 n = (u32)__builtin_popcount(valid);
    const u64 old = __hip_atomic_fetch_add(c, (u64)n | ((u64)sum << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u32 slot = (u32)old, before = (u32)(old >> 32);
    if (slot >= capacity || capacity - slot < n) return false; // (also: a counter that did not start at zero must not lead outside the list)
#pragma unroll
    for (u32 i = 0; i < kMany; ++i)
        if ((valid >> i) & 1u) {
            list[2ull * slot] = tile[i] | ((u64)(parts[i] | (buckets ? kDeferBuckets : 0u)) << 32);
            list[2ull * slot + 1] = before;
            ++slot;
            before += parts[i];
        }
    return true;
}
__device__ __forceinline__ bool dt_defer(u64 *c, u64 *list, u32 capacity, u64 tile, u64 groups, bool buckets = false) {
    const u64 t[1] = {tile}, g[1] = {groups};
    return dt_defer_many<1>(c, list, capacity, 1u, t, g, buckets);
}

// kWaveTiles: expand tiles a wavefront sums one after the other (long streams: 4, so that ticket, barrier and scan are
// paid once per 128 KiB... 512 KiB of stream; short streams: 1, more workgroups)
// kNoWait: the first launch of the NO-WAIT route (WAH_NO_WAIT / WAH_FORCE_FALLBACK=1, and what decompress() takes by
// itself after a WAH_ERR_TIMEOUT): workgroup tile = blockIdx, no ticket, no epoch; every expand tile's total goes to
// tile_base[] as it is, and sums_offsets_kernel makes bases of them in a second launch -- getCounts ->
// thrust::exclusive_scan (decompress.cu:66-80) over one entry per 4096 words instead of one per word.
template <u32 kWaveTiles, bool kNoWait = false>
__global__ __launch_bounds__(kSumWaves * 64) void decode_sums_kernel(const ScanArgs a) {
    __shared__ u64 s_part[kSumWaves * kWaveTiles];
    __shared__ u32 s_empty[kSumWaves * kWaveTiles];
    __shared__ u32 s_tile;

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const u32 wt = kNoWait ? blockIdx.x : draw_tile(a.ctrl, &s_tile); // workgroup tile: arrival order (wah_device.hpp)
    const u32 n_tiles = (u32)a.n_tiles;            // expand tiles (4096 words)
    const u32 n_wg_tiles = gridDim.x;
    const u32 et0 = (wt * kSumWaves + wave) * kWaveTiles; // this wave's expand tiles: et0 .. et0 + kWaveTiles - 1

    LaunchEpoch le = {};
    if (!kNoWait) {
        le = launch_epoch_begin(a.ctrl, wt, n_wg_tiles, a.gen_desc, a.scan_words, 0);
        if (le.bad) {
            if (blockIdx.x == 0 && threadIdx.x == 0) a.info[0] = a.info[1] = 0;
            return;
        }
    } else if (wt == 0 && threadIdx.x == 0) { // a new launch: forget the previous one's status (nothing of this launch raises an error)
        __hip_atomic_store(a.ctrl + kCtlError, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const u32 epoch = le.epoch;

    // ---- this wave's expand tiles: sums of the group counts -----------------------------------------------------------
    // A fill word of count 0 expands to nothing; the reference decoder steps over it (kernels.cu:332-354).  The expand
    // kernel's rank arithmetic assumes that every word owns at least one group, so every tile is checked here and
    // expand takes its index-map route for the tiles concerned (tile_flags).
    // Rolling rounds of 1024 words (four coalesced 1 KiB loads per round, the next round in flight while this one is
    // summed): 4 KiB per wave in flight keeps the memory pipeline busy without flooding it.
    const bool fast = a.aligned16 != 0;
    auto round_whole = [&](u32 rd) { // round rd of the wave's tiles lies wholly inside the stream
        return fast && (u64)et0 * kScanTileWords + (u64)(rd + 1u) * 1024u <= a.c_words;
    };
    auto issue_round = [&](u32 rd, u32x4 (&v)[4]) {
        const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.comp + (u64)et0 * kScanTileWords + (u64)rd * 1024u, 4096u);
#pragma unroll
        // nontemporal (aux = 2), as the compress kernel's loads of the bitmap (wah_compress.hip: load16)
        for (int k = 0; k < 4; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16u + 1024u * k, 0, 2);
    };
    constexpr u32 kRounds = 4u * kWaveTiles;
#ifndef WAH_SUM_AHEAD
#define WAH_SUM_AHEAD 1
#endif
    constexpr u32 kAhead = WAH_SUM_AHEAD; // rounds in flight behind the one being summed
    u32x4 buf[kAhead + 1][4];
#pragma unroll
    for (u32 rd = 0; rd < kAhead && rd < kRounds; ++rd)
        if (round_whole(rd)) issue_round(rd, buf[rd % (kAhead + 1)]);
    u64 mine = 0;
    bool has_empty = false;
#pragma unroll
    for (u32 rd = 0; rd < kRounds; ++rd) {
        const u32 et = et0 + rd / 4u;
        if (et < n_tiles) {
            if (round_whole(rd)) {
                if (rd + kAhead < kRounds && round_whole(rd + kAhead)) issue_round(rd + kAhead, buf[(rd + kAhead) % (kAhead + 1)]);
                const u32x4(&cur)[4] = buf[rd % (kAhead + 1)];
                u32 lo = 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) { // four counts of < 2^30 each fit 32 bits
                    const u32 nx = word_groups(cur[k].x), ny = word_groups(cur[k].y), nz = word_groups(cur[k].z), nw = word_groups(cur[k].w);
                    mine += (u64)(nx + ny + nz + nw);
                    lo = min(min(lo, min(nx, ny)), min(nz, nw));
                }
                has_empty |= lo == 0u;
            } else { // the stream's last rounds, or a stream that is only 4-byte aligned: word by word, bounds checked
                const u64 w0 = (u64)et0 * kScanTileWords + (u64)rd * 1024u;
                for (u32 i = lane; i < 1024u; i += 64u) {
                    if (w0 + i < a.c_words) {
                        const u32 n = word_groups(a.comp[w0 + i]);
                        mine += n;
                        has_empty |= n == 0u;
                    }
                }
            }
        }
        if (rd % 4u == 3u) { // a tile is complete
            const u64 tile_total = uniform64(wave_sum(mine));
            const bool any_empty = __any(has_empty);
            if (lane == 0) {
                s_part[wave * kWaveTiles + rd / 4u] = tile_total;
                s_empty[wave * kWaveTiles + rd / 4u] = any_empty ? 1u : 0u;
            }
            mine = 0;
            has_empty = false;
        }
    }
    __syncthreads();
    if (wave != 0) return;

    // ---- wave 0: the workgroup tile's total goes out, then the groups in front of it ---------------------------------
    static_assert(kSumWaves * kWaveTiles <= 64, "one lane per expand tile of the workgroup tile");
    const u64 part = lane < kSumWaves * kWaveTiles ? s_part[lane] : 0ull;
    // lane w: expand tile et0 + w (wave 0: et0 = the workgroup tile's first expand tile).  A tile that expands to very many
    // segments is not left to the `parts` workgroups the expand launch gives every tile alike (a hole of 100 000 segments in
    // otherwise dense data: 26 ms in one workgroup): onto the list with it, 32 segments per entry, and bit 1 of its flags.
    if (lane < kSumWaves * kWaveTiles && et0 + lane < n_tiles) {
        bool listed = false;
        if (a.defer_list && part / kSegGroups > kListSegs) {
            listed = dt_defer(defer_counters_mine(a.ctrl + kCtlDefer), a.defer_list, a.defer_capacity, et0 + lane, part);
        }
        a.tile_flags[et0 + lane] = (uint8_t)(s_empty[lane] | (listed ? 2u : 0u));
    }
    if (kNoWait) { // totals only (wave 0: et0 = the workgroup tile's first expand tile); sums_offsets_kernel does the rest
        if (lane < kSumWaves * kWaveTiles && et0 + lane < n_tiles) a.tile_base[et0 + lane] = part < kSumSaturate ? part : kSumSaturate;
        return;
    }
    const u64 incl_part = wave_scan_incl(part, lane);
    u64 total = uniform64(__shfl(incl_part, 63));
    bool overflow = total >= kSumSaturate;
    if (overflow) total = kSumSaturate;
    u64 end;
    const u64 base = sums_resolve(a.ctrl, a.gen_desc, wt, epoch, total, lane, end);
    overflow |= end >= kSumSaturate;
    // lane w: groups in front of expand tile et0 + w (wave 0: et0 = the workgroup tile's first expand tile)
    if (lane < kSumWaves * kWaveTiles && et0 + lane < n_tiles) a.tile_base[et0 + lane] = base + (incl_part - part);
    if (lane == 0) {
        if (overflow) atomicOr(a.ctrl + kCtlError, kErrStream);
        if (wt == n_wg_tiles - 1) {
            a.tile_base[n_tiles] = end;
            a.info[1] = end;
            a.info[0] = (31ull * end + 31ull) / 32ull; // decompress.cu:84-93
            if (a.host_result) { // every scan of the launch can complete now, so the error word is final
                a.host_result[1] = (31ull * end + 31ull) / 32ull;
                a.host_result[2] = end;
                a.host_result[0] = 1ull | ((u64)__hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32);
            }
            launch_epoch_end(a.ctrl, le);
            if (a.defer_list) defer_counters_next(a.ctrl + kCtlDefer);
        }
    }
}

// The LDS image of a tile is followed by room for the words BEHIND the tile that its last segment needs (a segment
// starts inside the tile and takes at most 1024 words from there; marking reads whole batches of 128): the mark
// phase parks them there as it reads them, so that the expansion gathers every word from LDS.  Gathering them from
// global memory -- one dependent load per step, and on this hardware a load also waits for every older STORE of the
// wavefront (one counter for both) -- made the tile's last segment its slowest by far: 0.426 -> 0.395 ms on the sparse GiB.
constexpr u32 kTileBehind = kSegGroups + 128u;
constexpr u32 kTileLdsWords = (u32)kScanTileWords + kTileBehind;

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
    u32 soff = o != 31u ? (lane - (lane >> 5)) * 4u : 0xFFFFF000u; // lanes 31, 63: out of range, dropped
    // (opaque: otherwise the sixteen store offsets soff + 248 s are computed once per kernel and sit in sixteen registers
    //  for its whole life instead of being the instructions' immediate offsets)
    asm volatile("" : "+v"(soff));
    const uint4 fq = reinterpret_cast<const uint4 *>(flag)[lane];
    const u32 f[4] = {fq.x, fq.y, fq.z, fq.w};
    u32 before = uniform32(first_word) - 1u;              // first_word + flags in earlier steps - 1: scalar
    // four steps at a time: their ranks first, then the four gathers together, then decode and store
#pragma unroll
    for (int g = 0; g < (int)kSteps / 4; ++g) {
        u32 r4[4], src_word[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 fb = (f[g] >> (8 * q)) & 0xFFu;     // 1: a word starts at my group
            const u64 m = __ballot(fb != 0u);
            // rank among the flags so far = flags at or below my lane = bits of (m >> 1) below my lane (v_mbcnt) + bit 0; the
            // running count stays on the scalar unit
            const u64 m1 = m >> 1;
            const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m1, 0u));
            r4[q] = (below + (before + ((u32)m & 1u))) << 2; // byte offset of the source word
            before += (u32)__builtin_popcountll(m);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            src_word[q] = kLocal ? *reinterpret_cast<const u32 *>(reinterpret_cast<const unsigned char *>(s_words) + r4[q])
                                 : tile_word(s_words, a, tile_w0, r4[q] >> 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * g + q;
            // fill -> 31 copies of bit 30, literal -> itself (kernels.cu:332-354); bit 31 of a fill's group is not cleared: the
            // repack takes bits 0..30 only
            u32 grp = (int)src_word[q] < 0 ? (u32)__builtin_amdgcn_sbfe((int)src_word[q], 30, 1) : src_word[q];
            if (!kWhole && (u32)(64 * s) + lane >= nvalid) grp = 0u;
            const u32 hi_part = (u32)__builtin_amdgcn_mov_dpp((int)(grp << up), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
            const u32 word = __builtin_amdgcn_ubfe(grp, o, 31u - o) | hi_part;
            // (default cache policy: nontemporal stores here cost the clustered round trip 22 % -- the memory-side cache
            //  combines this kernel's 248-byte stores)
            __builtin_amdgcn_raw_buffer_store_b32(word, rsrc, soff + 248u * s, 0, 0);
        }
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
        if (local)
            expand_steps<false, true>(a, s_words, flag, tile_w0, rsrc, first_word, nvalid, lane);
        else
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

// ---- a whole segment that lies inside ONE fill word (the inside of a long run: classic WAH, mostly empty bitmaps): 992 equal
// words -- no flags, no ranks, no gathers: four 16-byte stores per lane, the store shape of a fill kernel (248 pieces of 16 bytes;
// the descriptor drops what lies behind them and behind the capacity)
__device__ __forceinline__ void store_constant_segment(u32 *out, u64 seg_w0, u32 seg_words, u32 fill_word, u32 lane) {
    const u32 v = (fill_word & 0x40000000u) ? 0xFFFFFFFFu : 0u;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(out + seg_w0, seg_words * 4u);
    const u32x4 q = {v, v, v, v};
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q, rsrc, lane * 16u, 1024 * i, 0);
}

// One word of a segment into the segment's IMAGE in LDS (992 words = its 31 744 bits, zeroed): the word's groups begin at bit
// 31 p.  A literal is OR-ed in as one or two pieces; a fill of zeros is nothing at all; a fill of ones ORs the partial words
// at its two ends and leaves [first_full, last_full) to the wave (all ones: the callers' loop below).  The 31 -> 32 repack of
// mergeWords (kernels.cu:375) is the addressing: groups abut in the image as they do in the bitmap.
__device__ __forceinline__ void image_put(lds_u32_ptr img, u32 w, u32 p, u32 n, bool active, u32 &first_full, u32 &last_full) {
    first_full = last_full = 0u;
    if (!active) return;
    const u32 bit0 = 31u * p, q = bit0 >> 5, sh = bit0 & 31u;
    if ((int)w >= 0) { // literal: 31 bits from bit0 on
        __hip_atomic_fetch_or(img + q, w << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (sh >= 2u) __hip_atomic_fetch_or(img + q + 1, w >> (32u - sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (w & 0x40000000u) { // ones: bits [bit0, bit1)
        const u32 bit1 = 31u * (p + n), q1 = bit1 >> 5, e = bit1 & 31u;
        if (q == q1) {
            __hip_atomic_fetch_or(img + q, (~0u << sh) & ~(~0u << e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (sh != 0u) __hip_atomic_fetch_or(img + q, ~0u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (e != 0u) __hip_atomic_fetch_or(img + q1, ~(~0u << e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            first_full = sh != 0u ? q + 1u : q;
            last_full = q1;
        }
    }
}
// ... the whole words inside the wave's fills of ones (first_full < last_full in some lanes): every such fill by all lanes
__device__ __forceinline__ void image_fill_ones(lds_u32_ptr img, u32 first_full, u32 last_full, u32 lane) {
    u64 m = __ballot(first_full < last_full);
    while (m != 0ull) {
        const u32 l = (u32)__builtin_ctzll(m);
        m &= m - 1ull;
        const u32 f = (u32)__builtin_amdgcn_readlane((int)first_full, l), e = (u32)__builtin_amdgcn_readlane((int)last_full, l);
        for (u32 q = f + lane; q < e; q += 64u) img[q] = 0xFFFFFFFFu;
    }
}

// ---- one output segment, fast version: the tile expands to fewer than 2^31 groups, so every position relative to the
// segment start fits a signed 32-bit integer; bookkeeping stays in vector registers (see compress_kernel) ----------

// One batch of the mark phase: the next 128 words, two per lane.  `rel` = where the batch starts, seen from the
// segment start (<= 0 at first).  Flags the group at which every contributing word starts (clipped at the segment
// start).  kFirst: returns the tile-local index of the first contributing word.
// kFirst also: `whole` = the word that covers the WHOLE segment [0, nvalid), if one does (a fill: nonzero), else 0.
// (what a lane's two words of the first batch are and which of the segment's groups they cover, clipped: for the scatter)
struct MarkBatch {
    u32 w0, w1;
    int s0, e0, s1, e1;
};
template <bool kLocal, bool kFirst>
__device__ __forceinline__ u32 mark_pairs(const ExpandArgs &a, u32 *s_words, unsigned char *flag, u64 tile_w0,
                                          u32 left_in_stream, u32 nvalid, u32 lane, int &rel, u32 &wi, u32 *whole = nullptr,
                                          MarkBatch *mb = nullptr) {
    const u32 i0 = wi + 2u * lane;
    u32 w0, w1;
    if (kLocal) {
        const uint2 q = *reinterpret_cast<const uint2 *>(s_words + i0);
        w0 = q.x;
        w1 = q.y;
    } else { // past the tile: global memory; past the stream: empty fills
        w0 = i0 < left_in_stream ? a.comp[tile_w0 + i0] : 0x80000000u;
        w1 = i0 + 1u < left_in_stream ? a.comp[tile_w0 + i0 + 1u] : 0x80000000u;
        // parked behind the tile for the expansion (words of the tile itself are written again with the same values)
        if (i0 + 1u < kTileLdsWords) *reinterpret_cast<uint2 *>(s_words + i0) = make_uint2(w0, w1);
    }
    const u32 n0 = word_groups(w0), n1 = word_groups(w1);
    // all literals (dense data): consecutive positions, no scan; otherwise one DPP scan over the pair sums
    const u32 incl = __ballot((int)(w0 | w1) < 0) == 0 ? 2u * lane + 2u : wave_scan_incl32(n0 + n1);
    const int hi1 = rel + (int)incl, lo1 = hi1 - (int)n1, lo0 = lo1 - (int)n0; // words cover [lo0, lo1) and [lo1, hi1)
    // clip to the segment [0, nvalid): a word contributes iff something is left of it
    const int s0 = lo0 > 0 ? lo0 : 0, e0 = lo1 < (int)nvalid ? lo1 : (int)nvalid;
    const int s1 = lo1 > 0 ? lo1 : 0, e1 = hi1 < (int)nvalid ? hi1 : (int)nvalid;
    const bool c0 = e0 > s0, c1 = e1 > s1;
    if (kFirst && mb) *mb = MarkBatch{w0, w1, s0, e0, s1, e1};
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
        if (whole) { // (the first contributing word is the only candidate)
            const bool w0_all = c0 && lo0 <= 0 && lo1 >= (int)nvalid, w1_all = c1 && lo1 <= 0 && hi1 >= (int)nvalid;
            const u64 ma = __ballot(w0_all || w1_all);
            *whole = ma ? (u32)__builtin_amdgcn_readlane((int)(w0_all ? w0 : w1), __builtin_ctzll(ma)) : 0u;
        }
    }
    rel += (int)(u32)__builtin_amdgcn_readlane((int)incl, 63);
    wi += 128u;
    return first;
}

#ifndef WAH_EXPAND_SCATTER
#define WAH_EXPAND_SCATTER 0 // 1: decode_expand_kernel too
#endif
#ifndef WAH_LIST_SCATTER
#define WAH_LIST_SCATTER 1 // the list's launch
#endif
// kScatter (the list's launch, whose flag areas are 2 KB): a whole segment all of whose words lie in the first marking batch --
// a highly compressed stream's: a clustered bitmap has 16 words per segment -- is expanded by SCATTER, half a segment at a time
// (512 groups = 496 words: the image of a half fits the flag area): the half's image zeroed, every word of the batch puts the
// part of itself that lies in the half where it belongs (image_put: a fill of zeros is nothing), two 16-byte stores per lane.
// The work goes with the segment's words instead of sixteen steps of rank, gather, decode and repack whatever it holds.
template <bool kScatter>
__device__ __forceinline__ void expand_segment_tame(const ExpandArgs &a, u32 *s_words, const u32 *s_coarse32,
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
    u32 whole = 0;
    MarkBatch mb;
    const u32 first_word = wi <= kLastLocal ? mark_pairs<true, true>(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi, &whole, &mb)
                                            : mark_pairs<false, true>(a, s_words, flag, tile_w0, left_in_stream, nvalid, lane, rel, wi, &whole, &mb);
    if (whole != 0u && nvalid == kSegGroups && (seg + 1) * kSegWords <= out_words) { // (wave-uniform) the inside of a long fill
        const u64 seg_w0 = seg * kSegWords;
        store_constant_segment(a.out, seg_w0, kSegWords, whole, lane);
        return;
    }
    if (kScatter && rel >= (int)nvalid && nvalid == kSegGroups && (seg + 1) * kSegWords <= out_words && ((uintptr_t)a.out & 15u) == 0u) {
        lds_u32_ptr img = (lds_u32_ptr)reinterpret_cast<u32 *>(flag);
        uint4 *const img4 = reinterpret_cast<uint4 *>(flag);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            img4[lane] = make_uint4(0, 0, 0, 0);
            img4[64 + (int)lane] = make_uint4(0, 0, 0, 0);
            const int g0 = 512 * h, g1 = g0 + 512;
            const int s0 = mb.s0 > g0 ? mb.s0 : g0, e0 = mb.e0 < g1 ? mb.e0 : g1, s1 = mb.s1 > g0 ? mb.s1 : g0, e1 = mb.e1 < g1 ? mb.e1 : g1;
            u32 f0, l0, f1, l1;
            image_put(img, mb.w0, (u32)(s0 - g0), (u32)(e0 - s0), e0 > s0, f0, l0);
            image_put(img, mb.w1, (u32)(s1 - g0), (u32)(e1 - s1), e1 > s1, f1, l1);
            image_fill_ones(img, f0, l0, lane);
            image_fill_ones(img, f1, l1, lane);
            const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + seg * kSegWords + 496u * (u32)h, 1984u);
            const u32x4 q0 = reinterpret_cast<const u32x4 *>(flag)[lane], q1 = reinterpret_cast<const u32x4 *>(flag)[64 + (int)lane];
            __builtin_amdgcn_raw_buffer_store_b128(q0, rsrc, lane * 16u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(q1, rsrc, lane * 16u, 1024, 0); // (behind word 495 of the half: dropped)
        }
        return;
    }
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
    expand_emit(a, s_words, flag, tile_w0, first_word, nvalid, out_words, seg, wi <= kTileLdsWords, lane);
}

// one expand tile (4096 words of the stream): the output segments that start inside it (every parts-th batch of them).
// (The LDS arrays are this function's own: decode_expand_kernel and decode_expand_list_kernel each inline it once.)
// kContiguous (the list's launch): part p = the p-th of `parts` equal shares of CONSECUTIVE segments of the tile, not every parts-th
// group of four -- and with `buckets` (the tile's 64 sums of group counts per 64 words, left by decode_tile_kernel) the workgroup stages
// only the words those segments need instead of the whole tile: a tile of a highly compressed stream is shared by eight or
// more work items, each of which used to read all 16 KiB of it and count them (a quarter of the item's life, 1.13 x the
// algorithmic traffic on the clustered GiB).
template <bool kContiguous>
__device__ __forceinline__ void expand_tile(const ExpandArgs &a, u32 tile, u32 part, u32 parts, const u32 *buckets) {
    __shared__ __attribute__((aligned(16))) u32 s_words[kTileLdsWords];
    __shared__ u64 s_coarse[kCoarse + 1]; // groups in front of word 64 c, relative to the tile start
    __shared__ u32 s_coarse32[kCoarse + 1]; // the same in 32 bits (valid when the tile total is below 2^31)
    __shared__ u64 s_wave_sum[kExpandWaves];
    // (the list's launch: 2 KB per wave -- the image of half a segment for expand_segment_tame's scatter)
    __shared__ __attribute__((aligned(16))) unsigned char s_flag[kExpandWaves][(WAH_EXPAND_SCATTER || (WAH_LIST_SCATTER && kContiguous)) ? 2048u : kFlagBytes]; // 1: a word starts at this group
    // (the thread's number through an opaque move: inside the list launch's loop over work items everything derived from it --
    //  lane constants, LDS addresses -- would otherwise be made once in front of the loop and kept in registers across it: 123
    //  registers against the 70 of the routine on its own)
    u32 tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const u32 lane = tid & 63u;
    const u32 wave = (u32)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
#ifdef WAH_DIAG
    const u64 dg_start = __builtin_amdgcn_s_memrealtime();
#endif
    const u64 tile_w0 = (u64)tile * kScanTileWords;
    const u64 groups = a.info[1];
    const u64 out_words = a.info[0];
    // (asked for up front, used after the prologue: this tile or the next one contains fill words of count 0)
    const u64 all_tiles = (a.c_words + kScanTileWords - 1) / kScanTileWords;
    const bool has_empties = ((a.tile_flags[tile] | ((u64)tile + 1 < all_tiles ? a.tile_flags[tile + 1] : 0)) & 1) != 0;
    if (out_words > a.out_capacity) {
        if (tid == 0) atomicOr(a.ctrl + kCtlError, kErrCapacity);
        return;
    }

    // ---- with bucket sums: the coarse prefix is their scan, and only the words of this part's segments are staged ---------
    __shared__ u32 s_partial; // 1: the coarse prefix is made, the part's words are staged
    bool partial = false;
    u64 base_ahead = 0; // (asked for in front of the bucket scan and its barrier, not behind them)
    if (kContiguous && buckets != nullptr) {
        base_ahead = a.tile_base[tile];
        if (wave == 0) {
            const u32 v = buckets[lane];
            const bool sat = __any(v == kBucketSaturated);
            const u32 incl_b = wave_scan_incl32(sat ? 0u : v);
            s_coarse[lane] = incl_b - v;
            s_coarse32[lane] = incl_b - v;
            if (lane == 63) {
                s_coarse[kCoarse] = incl_b;
                s_coarse32[kCoarse] = incl_b;
                s_partial = sat ? 0u : 1u;
            }
        }
        __syncthreads();
        partial = uniform32(s_partial) != 0u;
    }
    if (partial) {
        const u64 base_p = base_ahead;
        const u32 total_p = uniform32(s_coarse32[kCoarse]); // (< 2^31: no bucket is saturated)
        const u64 kb = (base_p + kSegGroups - 1) / kSegGroups;
        const u64 ke = (base_p + total_p + kSegGroups - 1) / kSegGroups;
        const u64 per = (ke - kb + parts - 1) / parts; // the tile's segments in `parts` EQUAL shares (see below)
        const u64 s0 = kb + per * part;
        const u64 s1 = s0 + per < ke ? s0 + per : ke;
        if (s0 >= s1) return; // (wave-uniform; dt_defer's parts are an upper bound)
        const u32 t0 = (u32)(s0 * kSegGroups - base_p);                                   // first group needed (< total)
        const u64 t1_64 = s1 * kSegGroups - base_p;
        const u32 t1 = t1_64 < total_p ? (u32)t1_64 : total_p;                             // one behind the last group needed of THIS tile
        const u32 c = lane < kCoarse ? s_coarse32[lane] : 0xFFFFFFFFu;
        const u32 b0 = (u32)__popcll(__ballot(lane < kCoarse && c <= t0)) - 1u;            // bucket of the first group
        const u32 b1 = (u32)__popcll(__ballot(lane < kCoarse && c < t1)) - 1u;             // bucket of the last one
        // (+ one marking batch of 128 words: mark_pairs reads whole batches from the segment's bucket on)
        const u32 w_lo = 64u * b0, w_hi = 64u * (b1 + 1u) + 128u < (u32)kScanTileWords ? 64u * (b1 + 1u) + 128u : (u32)kScanTileWords;
        if (a.aligned16 && tile_w0 + kScanTileWords <= a.c_words) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + tile_w0);
            uint4 *dst = reinterpret_cast<uint4 *>(s_words);
            for (u32 i = w_lo / 4u + tid; i < w_hi / 4u; i += kExpandThreads) dst[i] = src[i];
        } else {
            for (u32 i = w_lo + tid; i < w_hi; i += kExpandThreads) s_words[i] = tile_w0 + i < a.c_words ? a.comp[tile_w0 + i] : 0x80000000u;
        }
        __syncthreads();
    } else {
    // ---- stage the tile and build the coarse prefix of group counts --------------------------------------------
    constexpr int kVec = kExpandWordsPerThread / 4;
    if (a.aligned16 && tile_w0 + kScanTileWords <= a.c_words) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + tile_w0);
        uint4 *dst = reinterpret_cast<uint4 *>(s_words);
        uint4 v[kVec];
#pragma unroll
        // (default cache policy: nontemporal loads here made the round trip slower, 1469 -> 1423 GB/s on the sparse GiB)
        for (int k = 0; k < kVec; ++k) v[k] = src[k * kExpandThreads + (int)tid]; // coalesced 16-byte loads
#pragma unroll
        for (int k = 0; k < kVec; ++k) dst[k * kExpandThreads + (int)tid] = v[k];
    } else {
        for (u32 i = tid; i < (u32)kScanTileWords; i += kExpandThreads)
            s_words[i] = tile_w0 + i < a.c_words ? a.comp[tile_w0 + i] : 0x80000000u; // past the end: empty fill
    }
    __syncthreads();
    // every thread sums the counts of its own 16 consecutive words
    u64 mine = 0;
    {
        const uint4 *my = reinterpret_cast<const uint4 *>(s_words + tid * kExpandWordsPerThread);
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
    if (tid % kThreadsPer64 == 0) { // first thread of each 64 words
        s_coarse[tid / kThreadsPer64] = excl;
        s_coarse32[tid / kThreadsPer64] = (u32)excl;
    }
    if (tid == kExpandThreads - 1) {
        s_coarse[kCoarse] = excl + mine;
        s_coarse32[kCoarse] = (u32)(excl + mine);
    }
    __syncthreads();
    }

#ifdef WAH_DIAG
    const u64 dg_ready = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- segments owned by this tile: those whose first group lies in [base, base + total) ----------------------
    const u64 base = a.tile_base[tile];
    const u64 total = uniform64(s_coarse[kCoarse]);
    const u64 n_seg = (groups + kSegGroups - 1) / kSegGroups;
    const u64 k_begin = (base + kSegGroups - 1) / kSegGroups;
    u64 k_end = (base + total + kSegGroups - 1) / kSegGroups;
    // kContiguous: the segments that start in the tile in `parts` EQUAL shares of consecutive segments (dt_defer's parts = one per
    // kDeferPartSegs segments, rounded up: 35 segments are two shares of 18 and 17, not 32 and 3 -- with unequal shares, and the
    // launch's workgroups taking items w, w + G, ..., every other workgroup got the long ones: 0.59 ms where the two launches
    // take 0.28, tools/decode_density_time.py, one bit in 2^9)
    const u64 per = kContiguous ? (k_end - k_begin + parts - 1) / parts : 0;
    if (k_end > n_seg) k_end = n_seg;

    if (has_empties) {
        // this tile, or the next one (a segment reads at most 1024 + 128 words past its tile), contains fill words of
        // count 0 (found by the sums pass): index-map route, one wavefront per workgroup, the four flag areas together
        // hold its 1024-entry index map
        static_assert(sizeof(s_flag) >= kSegGroups * sizeof(u32), "index map must fit the flag areas");
        if (wave == 0) {
            const u64 e_first = kContiguous ? k_begin + per * part : k_begin + part;
            const u64 e_last = kContiguous && e_first + per < k_end ? e_first + per : k_end;
            for (u64 seg = e_first; seg < e_last; seg += kContiguous ? 1u : parts)
                expand_segment_with_empties(a, s_words, s_coarse, reinterpret_cast<u32 *>(&s_flag[0][0]), tile_w0, base, groups,
                                            out_words, seg, lane);
        }
        return;
    }
    unsigned char *flag = s_flag[wave];
    const bool tame = total < (1ull << 31); // wave-uniform: positions inside this tile fit 32 bits
    const u64 seg_first = kContiguous ? k_begin + per * part + wave : k_begin + wave + (u64)kExpandWaves * part;
    const u64 seg_last = kContiguous && k_begin + per * (part + 1u) < k_end ? k_begin + per * (part + 1u) : k_end;
    for (u64 seg = seg_first; seg < seg_last; seg += kContiguous ? (u64)kExpandWaves : (u64)kExpandWaves * parts) {
        if (tame) {
            const u32 nvalid = (groups - seg * kSegGroups < kSegGroups) ? (u32)(groups - seg * kSegGroups) : kSegGroups;
            expand_segment_tame<WAH_EXPAND_SCATTER || (WAH_LIST_SCATTER && kContiguous)>(a, s_words, s_coarse32, flag, tile_w0, (u32)(seg * kSegGroups - base), nvalid, out_words, seg, lane);
        } else {
            expand_segment_general(a, s_words, s_coarse, flag, tile_w0, base, groups, out_words, seg, lane);
        }
    }
#ifdef WAH_DIAG
    __syncthreads();
    if (wave == 0 && lane == 0 && tile % 67u == 0u) { // a sample (tools/decode_phases.py): 100 MHz stamps
        unsigned long long *d = reinterpret_cast<unsigned long long *>(a.ctrl + 224); // (192..194: kCtlDefer)
        const u64 now = __builtin_amdgcn_s_memrealtime();
        atomicAdd(d + 0, (unsigned long long)(dg_ready - dg_start)); // tile staged, coarse prefix built
        atomicAdd(d + 1, (unsigned long long)(now - dg_ready));      // the tile's segments expanded
        atomicAdd(d + 2, 1ull);
    }
#endif
}

// the tiles on the list (dt_defer): workgroup w of G takes part p of an entry iff (its sum + p) mod G == w
// regular_parts: the parts decode_expand_kernel gives every tile, 0 = none (the one-pass decoder's list): a listed tile whose
// share per workgroup is at most 256 segments there has been expanded there (short streams: `parts` is large, and a stream
// whose tiles all expand alike is spread well by `parts` alone -- the list's launch has fewer workgroups)
__device__ __forceinline__ bool listed_tile_is_regular(const ExpandArgs &a, u32 tile, u32 regular_parts) {
    if (regular_parts == 0u) return false;
    const u64 segs = (a.tile_base[tile + 1] - a.tile_base[tile]) / kSegGroups;
    return segs <= 256ull * regular_parts;
}
__device__ __forceinline__ void expand_list(const ExpandArgs &a, const u64 *list, const u32 *count, u32 capacity, u32 w, u32 G, u32 regular_parts) {
    // a workspace that the launch before refused (WAH_ERR_WORKSPACE: neither zeroed nor left by a launch) holds no list
    if (__hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kErrWorkspace) return;
    const u64 n_et = (a.c_words + kScanTileWords - 1) / kScanTileWords;
    // the counter pair of the launch that has just ended (count[kDeferSeq] = launches so far; wah_internal.hpp, kCtlDefer): read
    // only -- the NEXT launch that may append zeroes it
    // (both pairs and the launch count in ONE round of loads -- three lanes -- not one behind the other: every load in front of
    //  an item's first word is a microsecond of its life when the chip is busy)
    const u32 lane = lane_id();
    const u64 cv = lane < 3u ? __hip_atomic_load(reinterpret_cast<const u64 *>(count) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    static_assert(kDeferSeq == 4u, "the launch count is the low half of the third 64-bit word");
    const u32 seq = (u32)__builtin_amdgcn_readlane((int)(u32)cv, 2);
    const u32 which = (seq - 1u) & 1u;
    u32 n = (u32)__builtin_amdgcn_readlane((int)(u32)cv, which);
    const u32 items = (u32)__builtin_amdgcn_readlane((int)(u32)(cv >> 32), which); // work items = the sum of the entries' parts
    if (n > capacity) n = capacity; // (whatever the counter holds: every access stays inside the list and the stream)
    // Work item k (k = w, w + G, ...) belongs to the LAST entry whose sum of earlier parts is <= k; the entries are sorted by
    // that sum (dt_defer).  FIRST a guess: tiles of one stream are much alike, so the entry lies near k * entries / items (all
    // tiles in one part each -- one bit in 2^9: exactly there) -- the 64 entries around the guess in one round of 16-byte loads,
    // found when the window holds an entry <= k in front of one > k (or of the list's end).  Otherwise a 64-ary search, two or
    // three rounds of one load each.  Every wave for itself, no barrier (the values are the same in all of them).
    for (u32 k = w; k < items; k += G) {
        u32 e = 0xFFFFFFFFu;
        u64 entry = 0;
        u32 before = 0;
        {
            const u32 guess = (u32)(((u64)k * n) / items);
            u32 w_lo = guess > 32u ? guess - 32u : 0u;
            if (w_lo + 64u > n) w_lo = n > 64u ? n - 64u : 0u;
            const u32 w_len = n - w_lo < 64u ? n - w_lo : 64u;
            u64 ev = 0, bv = ~0ull;
            if (lane < w_len) {
                const ulong2 q = *reinterpret_cast<const ulong2 *>(list + 2ull * (w_lo + lane));
                ev = q.x, bv = q.y;
            }
            const u32 below = (u32)__popcll(__ballot(lane < w_len && (u32)bv <= k));
            if (below != 0u && (below < w_len || w_lo + w_len == n)) {
                e = w_lo + below - 1u;
                entry = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(ev >> 32), below - 1u) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)ev, below - 1u);
                before = (u32)__builtin_amdgcn_readlane((int)(u32)bv, below - 1u);
            }
        }
        if (e == 0xFFFFFFFFu) {
            u32 lo = 0, len = n; // the entry lies in [lo, lo + len)
            while (len > 64u) {
                const u32 stride = (len + 63u) / 64u;
                const u32 i = lo + lane * stride;
                const bool le = lane * stride < len && (u32)list[2ull * i + 1] <= k;
                const u32 below = (u32)__popcll(__ballot(le)); // (lane 0's entry is always <= k: the one found in the round before)
                const u32 at = (below ? below - 1u : 0u) * stride;
                lo += at;
                len = len - at < stride ? len - at : stride;
            }
            const bool le = lane < len && (u32)list[2ull * (lo + lane) + 1] <= k;
            const u32 below = (u32)__popcll(__ballot(le));
            if (below == 0u) continue; // (a list that is not what dt_defer wrote)
            e = uniform32(lo + below - 1u);
            entry = uniform64(list[2ull * e]);
            before = uniform32((u32)list[2ull * e + 1]);
        }
        const u32 tile = (u32)entry, parts = (u32)(entry >> 32) & ~kDeferBuckets, part = k - before;
        // (a tile in ONE part needs all its words: staged whole, at once, as decode_expand_kernel does -- the bucket sums would be
        //  two more rounds of loads in front of the first word)
        const bool buckets = ((u32)(entry >> 32) & kDeferBuckets) != 0u && a.tile_buckets != nullptr && parts > 1u;
        if (tile >= n_et || parts == 0u || parts > 65536u || part >= parts) continue;
        if (listed_tile_is_regular(a, tile, regular_parts)) continue;
        if (k != w) __syncthreads(); // the LDS image goes to the next tile
        expand_tile<true>(a, tile, part, parts, buckets ? a.tile_buckets + (u64)tile * kCoarse : nullptr);
    }
}

// a stream of few tiles (highly compressed data) expands to many segments per tile: `parts` workgroups share a tile; the
// tiles the sums pass has put on the list (bit 1 of their flags) are left to the launch of decode_expand_list_kernel behind this one
// (the tile routine inside a loop over list entries takes 125 registers instead of 68: not in this kernel)
__global__ __launch_bounds__(kExpandThreads) void decode_expand_kernel(const ExpandArgs a) {
    const u32 tile = blockIdx.x / a.parts;
    // (on the list and too much for `parts` workgroups: decode_expand_list_kernel, the launch behind this one)
    if (a.defer_list && (a.tile_flags[tile] & 2) && !listed_tile_is_regular(a, tile, a.parts)) {
        // (an output that does not fit is reported all the same: the list's launch is left out when the CAPACITY is small)
        if (threadIdx.x == 0 && a.info[0] > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
        return;
    }
    expand_tile<false>(a, tile, blockIdx.x % a.parts, a.parts, nullptr);
}

#ifndef WAH_LIST_MINW
#define WAH_LIST_MINW 5 // (the flag areas of 2 KB per wave -- the scatter's half-segment image -- leave room for five workgroups per CU: up to 96 registers)
#endif
// the expand tiles decode_tile_kernel left to this route (giant fills, fill words of count 0: foreign streams), out of its
// list; normally the list is empty and the launch ends at once.
__global__ __launch_bounds__(kExpandThreads, WAH_LIST_MINW) void decode_expand_list_kernel(const ExpandArgs a, const u64 *list, const u32 *count, u32 capacity, u32 regular_parts) {
    expand_list(a, list, count, capacity, blockIdx.x, gridDim.x, regular_parts);
}

#include "wah_decode_tile.inc"

// ---------------------------------------------------------------------------------------------------------------------
// Decoding with the segment index (wah_decompress_segments_device).  A stream of compress() never lets a fill cross a
// 1024-group segment, and wah_compress_device_indexed() keeps where every segment's words start (the
// reference computes the same array, compress.cu:146, and drops it).  With the index every output segment is an
// independent job whose input range is known: no sums pass, no second read of the stream, any sub-range of the bitmap.
// One wavefront per segment: all loads of its words (at most 1024, 4 KiB) are issued before the first one is used,
// the words are parked in LDS, and the expansion gathers from there.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef WAH_SEG_WAVES
#define WAH_SEG_WAVES 4
#endif
constexpr int kSegDecodeWaves = WAH_SEG_WAVES;

// where the 992 words of segment first_segment + k go, and the lane constants of the 31 -> 32 repack
struct SegStore {
    __amdgpu_buffer_rsrc_t rsrc;
    u32 o, up, soff;
};
__device__ __forceinline__ SegStore seg_store_setup(u32 *out, u64 out_words, u64 seg, u64 k, u32 lane) {
    SegStore st;
    const u64 seg_w0 = seg * kSegWords;
    const u32 seg_words = out_words > seg_w0 ? (u32)(out_words - seg_w0 < kSegWords ? out_words - seg_w0 : kSegWords) : 0u;
    st.rsrc = make_rsrc(out + k * kSegWords, seg_words * 4u); // stores past the end are dropped
    st.o = lane & 31u;
    st.up = 31u - ((lane - 1u) & 31u);
    st.soff = st.o != 31u ? (lane - (lane >> 5)) * 4u : 0xFFFFF000u; // lanes 31, 63 only lend their group
    return st;
}
__device__ __forceinline__ void seg_store(const SegStore &st, int s, u32 grp) {
    const u32 hi_part = (u32)__builtin_amdgcn_mov_dpp((int)(grp << st.up), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    __builtin_amdgcn_raw_buffer_store_b32((grp >> st.o) | hi_part, st.rsrc, st.soff + 248u * s, 0, 0);
}

// A WHOLE segment (1024 groups, 992 words inside the bitmap, the output 16-byte aligned) by SCATTER: the image is zeroed, every
// word of the segment puts itself where it belongs (positions by a wave scan per batch of 128 words), the image goes out as
// four 16-byte stores per lane.  Against the gather (seg_mark + sixteen steps of rank, LDS read, decode, repack, dword store)
// the work goes with the segment's WORDS, not with its groups -- and a fill of zeros, most of a sparse bitmap's groups, costs
// nothing.  false: the words do not make up the segment (nothing stored).
__device__ __forceinline__ bool seg_expand_scatter(const SegRange &rg, const u32 (&x0)[kSegBatches], const u32 (&x1)[kSegBatches], u32 *image,
                                                   u32 *out, u32 lane) {
    lds_u32_ptr img = (lds_u32_ptr)image;
    uint4 *const img4 = reinterpret_cast<uint4 *>(image);
#pragma unroll
    for (int i = 0; i < 4; ++i) img4[64 * i + (int)lane] = make_uint4(0, 0, 0, 0);
    u32 pos = 0;
    bool bad = rg.bad != 0u;
#pragma unroll
    for (int b = 0; b < kSegBatches; ++b) {
        const u32 wi = 128u * b;
        if (wi < rg.cnt) { // wave-uniform
            const u32 i0 = wi + 2u * lane;
            const bool in0 = i0 < rg.cnt, in1 = i0 + 1u < rg.cnt;
            const u32 n0 = in0 ? min(word_groups(x0[b]), 2u * kSegGroups) : 0u, n1 = in1 ? min(word_groups(x1[b]), 2u * kSegGroups) : 0u;
            bad |= (in0 && n0 == 0u) || (in1 && n1 == 0u);
            const u32 incl = (wi + 128u <= rg.cnt && __ballot((int)(x0[b] | x1[b]) < 0) == 0) ? 2u * lane + 2u : wave_scan_incl32(n0 + n1);
            const u32 p1 = pos + incl - n1, p0 = p1 - n0;
            // (nothing is put outside the image, whatever the words say; the total below then fails)
            u32 f0, e0, f1, e1;
            image_put(img, x0[b], p0, n0, in0 && p0 + n0 <= kSegGroups, f0, e0);
            image_put(img, x1[b], p1, n1, in1 && p1 + n1 <= kSegGroups, f1, e1);
            image_fill_ones(img, f0, e0, lane);
            image_fill_ones(img, f1, e1, lane);
            pos += (u32)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    if (__ballot(bad) != 0ull || pos != kSegGroups) return false;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(out, kSegWords * 4u);
    u32x4 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = reinterpret_cast<const u32x4 *>(image)[64 * i + (int)lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(q[i], rsrc, lane * 16u, 1024 * i, 0); // (behind word 991: dropped)
    return true;
}

#ifndef WAH_SEG_SCATTER_MAX
#define WAH_SEG_SCATTER_MAX 768u // words of a segment up to which it is expanded by scatter (seg_expand)
#endif
// segment first_segment + k of the bitmap, its words in x0/x1 -> a.out + 992 k
__device__ __forceinline__ void seg_expand(const SegmentsArgs &a, u64 k, const SegRange &rg, const u32 (&x0)[kSegBatches],
                                           const u32 (&x1)[kSegBatches], unsigned char *flag, u32 *words, u32 lane) {
    // a segment that is ONE fill word of all its 1024 groups (mostly empty bitmaps: most of their segments): 992 equal words
    const u32 only = (u32)__builtin_amdgcn_readfirstlane((int)x0[0]);
    if (rg.cnt == 1u && !rg.bad && rg.nvalid == kSegGroups && only >= kFillZero && (only & kCountMask) == kSegGroups &&
        (a.first_segment + k + 1) * kSegWords <= a.out_words) {
        store_constant_segment(a.out, k * kSegWords, kSegWords, only, lane);
        return;
    }
    // (up to 768 words: 1 GiB clustered -- 16 words per segment -- 0.47-0.50 -> 0.55 of the roofline, 0.67-0.70 with seg_load_words
    //  asking only for the words there are; sparse -- 476 -- 0.70 as by gather; incompressible segments stay with the gather,
    //  0.73: by scatter 0.69.  The kernel's time is two dependent round
    //  trips and the stores' drain more than its instructions: a sparse segment's zero fills cost the scatter nothing, and it
    //  is no faster for it.)
    if (rg.cnt <= WAH_SEG_SCATTER_MAX && rg.nvalid == kSegGroups && (a.first_segment + k + 1) * kSegWords <= a.out_words && ((uintptr_t)a.out & 15u) == 0u) {
        if (!seg_expand_scatter(rg, x0, x1, words, a.out + k * kSegWords, lane) && lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    // (the bitmap's last segment, or an output that is only 4-byte aligned: by gather, clipped word by word)
    if (!seg_mark(rg, x0, x1, flag, words, lane)) {
        if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
        return;
    }
    const SegStore st = seg_store_setup(a.out, a.out_words, a.first_segment + k, k, lane);
    const uint4 fq = reinterpret_cast<const uint4 *>(flag)[lane];
    const u32 f[4] = {fq.x, fq.y, fq.z, fq.w};
    u32 before = 0xFFFFFFFFu;
#pragma unroll
    for (int s = 0; s < (int)kSteps; ++s) seg_store(st, s, seg_group(s, f, before, words, rg.cnt, rg.nvalid, lane));
}

// One segment per wavefront and no loop.  Measured alternatives, both slower (1 GiB, sparse / clustered / dense:
// 0.29 / 0.29 / 0.36 ms as it is): several segments per wavefront (2, 4, 8: 0.36-0.40 / 0.34-0.36 / 0.38-0.42 ms; the
// compiler hoists the lane constants out of the loop, 100 registers, 71 when capped), and persistent wavefronts with
// the next segment's words and the following offsets in flight during the expansion (0.47 / 0.42 / 0.51 ms,
// tools/experiments/segments_stream.diff).  Short-lived wavefronts that begin with their loads overlap best.
__global__ __launch_bounds__(kSegDecodeWaves * 64) void decode_segments_kernel(const SegmentsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_flag[kSegDecodeWaves][kSegGroups]; // 1: a word starts at this group
    __shared__ __attribute__((aligned(16))) u32 s_seg[kSegDecodeWaves][kSegGroups];           // the segment's words
    const u32 wave = wave_id(), lane = lane_id();
    const u64 k = (u64)blockIdx.x * kSegDecodeWaves + wave;
    if (k >= a.n_segments) return;
    const u64 seg = a.first_segment + k;
    // (the index is read-only for this launch and the segment's number the same in all lanes: through the scalar cache -- two of
    //  a clustered segment's eight memory instructions are these)
    typedef const __attribute__((address_space(4))) u64 *const_u64_ptr;
    const const_u64_ptr offs = (const_u64_ptr)(uintptr_t)(a.seg_offsets + seg);
    const SegRange rg = seg_range(a, seg, offs[0], offs[1]);
    u32 x0[kSegBatches], x1[kSegBatches];
    seg_load_words(a, rg, x0, x1, lane);
    seg_expand(a, k, rg, x0, x1, s_flag[wave], s_seg[wave], lane);
}

// wah_bitop_many_indexed_device: up to kMaxBitopOperands indexed streams, combined left to right (A op B op C ...;
// ANDNOT: A and not B and not C ...) in one pass: the accumulated segment stays in 16 registers, every further operand
// goes through the same LDS areas, and its words are loaded while the operand before it is expanded.  (A loop over the
// operands, not unrolled: eight unrolled copies need 120 registers, 168 bytes of scratch when capped, and are slower.)
__global__ __launch_bounds__(kSegDecodeWaves * 64, 6) void bitop_many_segments_kernel(const BitopManyArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_flag[kSegDecodeWaves][kSegGroups];
    __shared__ __attribute__((aligned(16))) u32 s_seg[kSegDecodeWaves][kSegGroups];
    const u32 wave = wave_id(), lane = lane_id();
    const u64 k = (u64)blockIdx.x * kSegDecodeWaves + wave;
    if (k >= a.g.n_segments) return;
    const u64 seg = a.g.first_segment + k;
    unsigned char *flag = s_flag[wave];
    u32 *words = s_seg[wave];
    // minterm masks of `acc op operand` (include/wah.h: WAH_OP_AND 0, OR 1, XOR 2, ANDNOT 3), wave-uniform
    const u32 k_ab = a.op <= 1 ? ~0u : 0u;
    const u32 k_a_nb = a.op == 0 ? 0u : ~0u;
    const u32 k_na_b = a.op == 1 || a.op == 2 ? ~0u : 0u;

    SegmentsArgs cur = a.g; // geometry; stream and index of the operand at hand
    cur.comp = a.comp[0];
    cur.c_words = a.c_words[0];
    SegRange rg = seg_range(cur, seg, uniform64(a.offs[0][seg]), uniform64(a.offs[0][seg + 1]));
    u32 x0[kSegBatches], x1[kSegBatches];
    seg_load_words(cur, rg, x0, x1, lane);
    u32 acc[kSteps];
    bool ok = true;
#pragma nounroll
    for (int j = 0; j < a.n; ++j) {
        // the next operand's words: in flight during this operand's expansion
        SegmentsArgs nxt = a.g;
        SegRange rn = rg;
        u32 y0[kSegBatches], y1[kSegBatches];
        const bool more = j + 1 < a.n;
        if (more) {
            nxt.comp = a.comp[j + 1];
            nxt.c_words = a.c_words[j + 1];
            rn = seg_range(nxt, seg, uniform64(a.offs[j + 1][seg]), uniform64(a.offs[j + 1][seg + 1]));
            seg_load_words(nxt, rn, y0, y1, lane);
        }
        const bool good = seg_mark(rg, x0, x1, flag, words, lane);
        ok = ok && good;
        const uint4 fq = reinterpret_cast<const uint4 *>(flag)[lane];
        const u32 f[4] = {fq.x, fq.y, fq.z, fq.w};
        u32 before = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < (int)kSteps; ++s) {
            const u32 g = seg_group(s, f, before, words, good ? rg.cnt : 1u, rg.nvalid, lane);
            acc[s] = j == 0 ? g : (acc[s] & g & k_ab) | (acc[s] & ~g & k_a_nb) | (~acc[s] & g & k_na_b);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // words and flags have been read: the areas go to the next operand
        if (more) {
#pragma unroll
            for (int b = 0; b < kSegBatches; ++b) {
                x0[b] = y0[b];
                x1[b] = y1[b];
            }
            rg = rn;
        }
    }
    if (!ok) {
        if (lane == 0) atomicOr(a.g.ctrl + kCtlError, kErrStream);
        return;
    }
    const SegStore st = seg_store_setup(a.g.out, a.g.out_words, seg, k, lane);
#pragma unroll
    for (int s = 0; s < (int)kSteps; ++s) seg_store(st, s, acc[s] & kOnes31);
}

} // namespace

// second launch of the no-wait route: totals of the expand tiles -> groups in front of every tile (exclusive scan in
// place, saturating), + everything the last workgroup tile of the scan route leaves behind.  One workgroup: the table
// has one entry per 16 KiB of stream (a 1 GiB stream: 65 536 of them).
__global__ __launch_bounds__(1024) void sums_offsets_kernel(const ScanArgs a) {
    __shared__ u64 s_wave[16];
    __shared__ u64 s_carry;
    const u32 lane = lane_id(), wave = wave_id();
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 t0 = 0; t0 < a.n_tiles; t0 += 1024u) {
        const u64 t = t0 + threadIdx.x;
        const u64 mine = t < a.n_tiles ? a.tile_base[t] : 0ull; // (each < 2^47 + 1: sixteen of them cannot wrap 64 bits)
        const u64 incl = wave_scan_incl(mine, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        u64 front = s_carry;
        for (u32 w = 0; w < wave; ++w) front = sat_add(front, s_wave[w]);
        if (t < a.n_tiles) a.tile_base[t] = sat_add(front, incl - mine);
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = sat_add(front, incl);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const u64 end = s_carry;
        if (end >= kSumSaturate) atomicOr(a.ctrl + kCtlError, kErrStream);
        a.tile_base[a.n_tiles] = end;
        a.info[1] = end;
        a.info[0] = (31ull * end + 31ull) / 32ull; // decompress.cu:84-93
        if (a.host_result) {
            a.host_result[1] = (31ull * end + 31ull) / 32ull;
            a.host_result[2] = end;
            a.host_result[0] = 1ull | ((u64)__hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32);
        }
        if (a.defer_list) defer_counters_next(a.ctrl + kCtlDefer);
    }
}

// workgroups of the launch that walks the list of deferred / shared-out tiles: one per work item the output can hold (an item =
// kDeferPartSegs segments), within bounds -- an empty list costs the launch of this many workgroups that read one counter
// (a stream of compress() holds at most one segment per word: its items are bounded by its length too -- what keeps the launch
// small for a small stream decoded into a large kept buffer; a foreign stream of few words and giant fills loops)
static unsigned defer_list_grid(u64 out_capacity, u64 c_words) {
    const u64 by_capacity = out_capacity / kSegWords / kDeferPartSegs;
    const u64 by_stream = c_words / kDeferPartSegs + 2u * (c_words / kScanTileWords + 1u);
    const u64 want = (by_capacity < by_stream ? by_capacity : by_stream) + 64u;
    return (unsigned)(want < 256u ? 256u : want > 32768u ? 32768u : want);
}

hipError_t launch_decode_sums(const ScanArgs &a, hipStream_t s) {
    if (a.no_wait) {
        const u64 wg_tiles = (a.n_tiles + kSumWaves - 1) / kSumWaves;
        hipLaunchKernelGGL((decode_sums_kernel<1, true>), dim3((unsigned)wg_tiles), dim3(kSumWaves * 64), 0, s, a);
        hipLaunchKernelGGL(sums_offsets_kernel, dim3(1), dim3(1024), 0, s, a);
        return hipGetLastError();
    }
    // long streams: four expand tiles per wave (ticket, barrier and scan once per 512 KiB); short ones: more workgroups
#ifndef WAH_SUM_TILES
#define WAH_SUM_TILES 4
#endif
    if (a.n_tiles >= 16384) {
        const u64 wg_tiles = (a.n_tiles + kSumWaves * WAH_SUM_TILES - 1) / (kSumWaves * WAH_SUM_TILES);
        hipLaunchKernelGGL(decode_sums_kernel<WAH_SUM_TILES>, dim3((unsigned)wg_tiles), dim3(kSumWaves * 64), 0, s, a);
    } else {
        const u64 wg_tiles = (a.n_tiles + kSumWaves - 1) / kSumWaves;
        hipLaunchKernelGGL(decode_sums_kernel<1>, dim3((unsigned)wg_tiles), dim3(kSumWaves * 64), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_decode_expand(const ExpandArgs &a0, u64 n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    // One workgroup per tile fills the chip when the stream is long.  A short stream that expands a lot (highly
    // compressed bitmaps: thousands of output segments per tile) is shared out: `parts` workgroups per tile, each
    // taking every parts-th group of kExpandWaves segments.  The true output size is only known on the device; the
    // capacity bounds it, and a part with nothing to do costs one 16 KiB tile read.
    ExpandArgs a = a0;
    static const u64 want = [] { // workgroups: 256 CUs x 6 resident x ~2.7 (experiments: WAH_EXPAND_WANT)
        const char *e = experiment_env("WAH_EXPAND_WANT");
        const long v = e ? std::atol(e) : 0;
        return v > 0 ? (u64)v : 4096ull;
    }();
    const u64 segs_per_tile = a.out_capacity / kSegWords / n_tiles;
    u64 parts = (want + n_tiles - 1) / n_tiles;
    // ... and a workgroup should not expand much more than 24 segments (six per wave): the chip writes faster the shorter its
    // waves live (clustered GiB, 1082 tiles of 250 segments: 4 parts 0.245 ms, 8..14 parts 0.226-0.231, 16 parts 0.249)
    if (parts < segs_per_tile / 24) parts = segs_per_tile / 24;
    if (parts > segs_per_tile / (2 * kExpandWaves)) parts = segs_per_tile / (2 * kExpandWaves);
    if (parts < 1) parts = 1;
    if (parts > 1024) parts = 1024;
    a.parts = (u32)parts;
    a.n_tile_wgs = (u32)(n_tiles * parts);
    hipLaunchKernelGGL(decode_expand_kernel, dim3((unsigned)(n_tiles * parts)), dim3(kExpandThreads), 0, s, a);
    // the tiles the sums pass listed and `parts` workgroups are not enough for (normally none: the launch ends at once; no tile
    // can be one when the whole output is no more than 256 segments per part: short streams do without the launch)
    if (a.defer_list && a.out_capacity / kSegWords > 256ull * parts) {
        ExpandArgs x = a;
        x.parts = 1;
        hipLaunchKernelGGL(decode_expand_list_kernel, dim3(defer_list_grid(a.out_capacity, a.c_words)), dim3(kExpandThreads), 0, s, x, a.defer_list, a.defer_count, a.defer_capacity, a.parts);
    }
    return hipGetLastError();
}

// the general decoder in one pass (decode_tile_kernel) + the launch that takes what it deferred
hipError_t launch_decode_tiles(const ScanArgs &sa, const ExpandArgs &xa, u64 *defer, hipStream_t s) {
    static const u32 batch = [] { // tiles per workgroup: 2 (experiments: WAH_DT_BATCH=1)
        const char *e = experiment_env("WAH_DT_BATCH");
        return e && e[0] == '1' ? 1u : 2u;
    }();
    TileDecodeArgs t;
    t.comp = sa.comp;
    t.c_words = sa.c_words;
    t.n_wg_tiles = (u32)((sa.c_words + (u64)batch * kDtTileWords - 1) / ((u64)batch * kDtTileWords));
    t.out = xa.out;
    t.out_capacity = xa.out_capacity;
    t.info = sa.info;
    t.tile_base = sa.tile_base;
    t.tile_flags = sa.tile_flags;
    t.defer_count = sa.ctrl + kCtlDefer; // (wah_internal.hpp)
    t.defer_list = defer;
    t.defer_capacity = decode_defer_capacity(sa.n_tiles, sa.c_words);
    t.tile_buckets = const_cast<u32 *>(xa.tile_buckets);
    t.ctrl = sa.ctrl;
    t.gen_desc = sa.gen_desc;
    t.scan_words = sa.scan_words;
    t.host_result = sa.host_result;
    if (batch == 1)
        hipLaunchKernelGGL(decode_tile_kernel<1>, dim3(t.n_wg_tiles), dim3(kDtWaves * 64), 0, s, t);
    else
        hipLaunchKernelGGL(decode_tile_kernel<2>, dim3(t.n_wg_tiles), dim3(kDtWaves * 64), 0, s, t);
#ifdef WAH_DIAG
    if (std::getenv("WAH_DIAG_NO_LIST")) return hipGetLastError(); // (the time line in the output's head survives: tools/decode_tile_timeline.py)
#endif
    ExpandArgs x = xa;
    x.parts = 1;
    x.defer_list = nullptr; // (the list is this launch's own argument)
    static const unsigned list_dyn_lds = [] { // experiments only: dynamic LDS as an occupancy limiter (WAH_LIST_DYNLDS = bytes)
        const char *e = experiment_env("WAH_LIST_DYNLDS");
        return e ? (unsigned)std::strtoul(e, nullptr, 0) : 0u;
    }();
    hipLaunchKernelGGL(decode_expand_list_kernel, dim3(defer_list_grid(xa.out_capacity, xa.c_words)), dim3(kExpandThreads), list_dyn_lds, s, x, (const u64 *)t.defer_list, (const u32 *)t.defer_count, t.defer_capacity, 0u);
    return hipGetLastError();
}

hipError_t launch_decode_segments(const SegmentsArgs &a, hipStream_t s) {
    if (a.n_segments == 0) return hipSuccess;
    const u64 grid = (a.n_segments + kSegDecodeWaves - 1) / kSegDecodeWaves;
    hipLaunchKernelGGL(decode_segments_kernel, dim3((unsigned)grid), dim3(kSegDecodeWaves * 64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_bitop_many_segments(const BitopManyArgs &a, hipStream_t s) {
    if (a.g.n_segments == 0) return hipSuccess;
    const u64 grid = (a.g.n_segments + kSegDecodeWaves - 1) / kSegDecodeWaves;
    hipLaunchKernelGGL(bitop_many_segments_kernel, dim3((unsigned)grid), dim3(kSegDecodeWaves * 64), 0, s, a);
    return hipGetLastError();
}

} // namespace wah
