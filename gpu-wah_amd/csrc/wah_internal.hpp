// wah_internal.hpp -- declarations shared by the kernel TU and the C-ABI TU.
//
// gfx950 / CDNA4 only: 64-lane wavefronts, 8 XCDs with private L2s.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

namespace wah {

// Experiment switches (tile shapes, batch sizes, forced routes: what tools/*.sh sweep) are read from the environment only
// by builds made with -DWAH_EXPERIMENTS (`make -C gpu-wah_amd exp` -> libwah_hip_exp.so, used through WAH_LIB_PATH); the
// shipped library reads none of them.  (What it does read is documented in include/wah.h: WAH_HOST_CACHE,
// WAH_FORCE_FALLBACK, WAH_FAULT_INJECT.)
inline const char *experiment_env(const char *name) {
#ifdef WAH_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- wire format (reference const.h:3-16) ---------------------------------
constexpr uint32_t kOnes31 = 0x7FFFFFFFu;    // ONES31
constexpr uint32_t kFillZero = 0x80000000u;  // BIT31
constexpr uint32_t kFillOne = 0xC0000000u;   // BIT3130
constexpr uint32_t kCountMask = 0x3FFFFFFFu; // BIT30 - 1 (kernels.cu:300)

// ---- segment geometry (kernels.cu:68: one CUDA block = 1024 groups) ------
constexpr uint32_t kSegWords = 992;
constexpr uint32_t kSegGroups = 1024;
constexpr uint32_t kSteps = 16; // 64 groups (one wavefront) per step

// ---- inter-workgroup control block (uint32 words)
// decode / aux kernels: zeroed before each launch.  compress: zeroed ONCE (wah_workspace_init_device), then kept up by
// the kernel itself (launch epochs, see compress_tile_kernel).
constexpr uint32_t kCtlStart = 0;        // arrival ticket: order in which workgroups start running (alone on its 128-byte line:
                                         // every workgroup of a launch adds to it)
constexpr uint32_t kCtlEpoch = 64;       // tile kernels: epoch of the NEXT launch (0 = fresh workspace); read-mostly line
constexpr uint32_t kCtlMagic = 65;       // tile kernels: kWorkspaceMagic once a launch has completed (0 = fresh workspace)
constexpr uint32_t kCtlWraps = 66;       // tile kernels: how often the epoch space has been used up
constexpr uint32_t kCtlClearDone = 96;   // tile kernels: == kCtlWraps + 1 once tile 0 of a wrapping launch has cleared the scan area
constexpr uint32_t kCtlError = 160;      // sticky error bits
constexpr uint32_t kCtlResult = 162;     // host-pointer entry points: two 64-bit results of the launch live here, beside the
                                         // error word, so that ONE 32-byte copy brings status and sizes to the host
constexpr uint32_t kCtlDefer = 192;      // the decoders' list of tiles that are decoded by workgroups of their own: two 64-bit counter pairs
                                         // {entries:32, sum of their parts:32} at words [0..1] and [2..3], used in turn, and at word
                                         // [kDeferSeq] the number of launches that could append so far (launch s counts in pair s & 1; its
                                         // last tile zeroes the other pair -- nobody reads it any more -- and stores s + 1: whoever walks
                                         // the list needs no atomics to hand the counters back).  HERE, not beside the list: where the list
                                         // lies depends on the size of the workspace a call names, and a counter at a place that moves
                                         // would be found holding an earlier call's data
constexpr uint32_t kDeferSeq = 4;
constexpr uint32_t kDeferBuckets = 0x80000000u; // in a list entry's parts field: the tile's bucket sums are in tile_buckets
constexpr uint32_t kBucketSaturated = 0xFFFFFFFFu; // a bucket holding a count above 2^25: the tile is staged whole
#ifndef WAH_DEFER_PART_SEGS
#define WAH_DEFER_PART_SEGS 32 // (clustered GiB through the list, one box: 16: 0.281 ms, 24: 0.251, 32: 0.2455, 48: 0.253)
#endif
constexpr uint32_t kDeferPartSegs = WAH_DEFER_PART_SEGS;  // output segments per work item of the list's launch (eight per wave: the chip writes faster the
                                         // shorter its waves live, tools/expand_want_sweep.sh)
constexpr uint32_t kCtlWords = 256;      // 1 KiB
constexpr uint32_t kErrTimeout = 1u;     // a bounded wait expired
constexpr uint32_t kErrCapacity = 2u;    // output would exceed its capacity
constexpr uint32_t kErrStream = 4u;      // malformed compressed stream
constexpr uint32_t kErrWorkspace = 8u;   // compress workspace neither zeroed nor left by an earlier launch
constexpr uint32_t kWorkspaceMagic = 0x57414832u; // "WAH2"
constexpr uint32_t kEpochWrap = (1u << 16) - 1u;  // stored epoch >= this: the next launch clears the scan area first

// ---- compress geometry: one wavefront owns one segment, a workgroup (tile) kCompressTileWaves of them ------------
#ifndef WAH_TILE_WAVES
#define WAH_TILE_WAVES 8
#endif
constexpr int kCompressTileWaves = WAH_TILE_WAVES;
constexpr int kCompressMaxWaveSegs = 5; // segments a wavefront compresses one after the other: 1, 2 or this (by bitmap size)
uint32_t compress_wave_segs(uint64_t n_segments);
// pair-layout kernel: the tile shapes of a launch (compress_pair_kernel): body tiles of body_pairs pairs per wavefront, then
// tail tiles of tail_pairs; body_pairs == 0: kernel switched off
struct TileShape {
    uint32_t body_pairs, tail_pairs, big_tiles, n_tiles;
};
TileShape compress_tile_shape(uint64_t n_segments);
// scan area of the compress kernel (see compress_tile_kernel): one block per superrow of 64 rows x 256 tiles
constexpr uint32_t kRowSlots = 65;                 // u64 slots of a superrow: words in front of it, words of each of its rows
constexpr uint32_t kScanSlotsAt = 64 * 256;        // 32-bit words: the slots follow the superrow's granules
constexpr uint32_t kScanBlockWords = 64 * 256 + 256; // granules + slots, padded to 1 KiB
constexpr uint64_t kScanBlockTiles = 64 * 256;
// ... and of its unsegmented mode (compress_unseg_pair_kernel): 8-byte granules, two slot arrays
constexpr uint32_t kUnsegSlotsAAt = 2 * 64 * 256;          // 32-bit words
constexpr uint32_t kUnsegSlotsBAt = 2 * 64 * 256 + 256;
constexpr uint32_t kUnsegBlockWords = 2 * 64 * 256 + 512;

// ---- decode geometry -------------------------------------------------------
constexpr int kScanTileWords = 4096; // compressed words per tile (both decode passes)
constexpr int kSumTilesPerGroup = 8;  // expand tiles per workgroup tile of the sums kernel (= its worker waves)
constexpr int kExpandWaves = 4;      // wavefronts of an expand workgroup (each expands whole segments)
// scan area of the sums kernel: one block per superrow of 64 rows x 256 workgroup tiles, 8-byte granules
constexpr uint32_t kSumScanSlotsAt = 2 * 64 * 256;          // 32-bit words: the slots follow the superrow's granules
constexpr uint32_t kSumScanBlockWords = 2 * 64 * 256 + 256; // granules + 65 slots, padded to 1 KiB
constexpr uint64_t kSumScanBlockTiles = 64 * 256;

struct CompressArgs {
    const uint32_t *in;
    const uint32_t *in2;   // pair mode (wah_bitop_device): second bitmap of the same length, combined word by word
    uint32_t op;           // ... with WAH_OP_AND / OR / XOR / ANDNOT
    uint64_t n_words;
    uint32_t n_segments;          // ceil(G / 1024)
    uint32_t wave_segs;           // segments per wavefront of this launch (compress_wave_segs)
    uint32_t pair_layout;         // 1: compress_pair_kernel (a lane owns 32 consecutive groups; wave_segs / 2 pairs per wavefront)
    uint32_t n_tiles;             // ceil(n_segments / (kCompressTileWaves * wave_segs)); pair layout: body + tail tiles
    uint32_t big_tiles;           // pair layout: tiles [0, big_tiles) have wave_segs / 2 pairs per wave, the rest tail_pairs
    uint32_t tail_pairs;          // pair layout: pairs per wave of the tail tiles (== wave_segs / 2: one shape)
    uint32_t fast_segments;       // 1: input 16-byte aligned -> prefetched buffer loads; 0: scalar staging
    uint32_t last_segment_groups; // groups of the last segment (1..1024)
    uint32_t full_segments;       // n_words / 992: segments that lie wholly inside the bitmap
    uint32_t tail_bytes;          // bytes of the partial segment after them (0 if none)
    uint32_t *out;
    uint64_t out_capacity;
    uint64_t *out_words;   // device scalar: C
    uint64_t *seg_offsets; // optional, n_segments + 1 entries
    uint32_t *ctrl;        // kCtlWords
    uint32_t *gen_desc;    // scan area: blocks of kScanBlockWords (see compress_tile_kernel)
    uint32_t *unseg_desc;  // non-null: unsegmented mode, its scan area: blocks of kUnsegBlockWords (compress_unseg_pair_kernel)
    uint64_t *tile_counts; // no-wait route only: one entry per tile, its word count, then where its words start
    uint64_t scan_words;   // 32-bit words of the whole scan area
    int keep_error;        // 1: the control block was cleared by the caller and may already hold an upstream error
    uint64_t *host_result; // optional, page-locked HOST memory: [0] = 1 | error bits << 32, [1] = C, written by the last tile
                           // (the host-pointer entry points read them after one event wait, without a copy)
    uint32_t tune;         // WAH_DIAG builds only (WAH_TUNE): time line modes of tools/tile_timeline.py; 0 otherwise
};

struct ScanArgs {
    const uint32_t *comp;
    uint64_t c_words;
    uint64_t n_tiles;
    uint64_t *info;      // [0] decoded words, [1] groups
    uint64_t *tile_base; // n_tiles + 1: groups in front of each tile, last = total
    uint32_t *ctrl;
    uint32_t *gen_desc;  // scan area: blocks of kSumScanBlockWords (see decode_sums_kernel)
    uint64_t scan_words; // 32-bit words of the whole scan area
    uint8_t *tile_flags; // n_tiles: bit 0 = the tile contains a fill word of count 0, bit 1 = it is on defer_list
    int aligned16;
    uint64_t *host_result; // optional, page-locked HOST memory: [0] = 1 | error bits << 32, [1..2] = info, by the last tile
    int no_wait;           // 1: the no-wait route (per-tile totals, then one scan launch): nobody waits for anybody
    uint64_t *defer_list;  // optional: tiles that expand to more than kListSegs segments go onto this list (dt_defer) and
    uint32_t defer_capacity; // are left to the expand launch's list workgroups; their tile_flags get bit 1
};
constexpr uint32_t kListSegs = 1024; // (a tile of 4096 words that expands to more than a million groups)
// entries (16 bytes each) the list of deferred / shared-out tiles has room for: a tile goes onto it once at most, the
// one-pass decoder may add the tile of a tile's last segment once more
inline uint32_t decode_defer_capacity(uint64_t n_tiles, uint64_t) {
    const uint64_t n = 2 * n_tiles + 64;
    return n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n;
}

struct ExpandArgs {
    const uint32_t *comp;
    uint64_t c_words;
    uint32_t *out;
    uint64_t out_capacity;
    const uint64_t *info;
    const uint64_t *tile_base;
    const uint8_t *tile_flags; // from the sums pass: bit 0 = the tile has fill words of count 0, bit 1 = it is on the list
    uint32_t *ctrl;
    int aligned16;
    uint32_t parts; // workgroups that share one tile's output segments (set by the launcher)
    const uint64_t *defer_list; // optional: the list the sums pass left (ScanArgs::defer_list), shared out over the launch's workgroups
    const uint32_t *defer_count; // control block, kCtlDefer
    uint32_t defer_capacity;
    uint32_t n_tile_wgs;        // workgroups in front of them: tiles x parts (set by the launcher)
    const uint32_t *tile_buckets; // per expand tile 64 sums of group counts, one per 64 words (kBucketSaturated: no sum), written by
                                  // decode_tile_kernel for the tiles it puts on the list with kDeferBuckets: a work item of the list's
                                  // launch then stages only the words its segments need
};

// wah_decompress_segments_device: decode a range of segments through the index of wah_compress_device_indexed
struct SegmentsArgs {
    const uint32_t *comp;
    uint64_t c_words;
    const uint64_t *seg_offsets; // first compressed word of every segment of the bitmap, then C
    uint64_t first_segment, n_segments;
    uint64_t groups;    // G of the whole bitmap
    uint64_t out_words; // ceil(31 G / 32): where the bitmap's last segment is cut
    uint32_t *out;      // receives segment first_segment at word 0
    uint32_t *ctrl;
};

// wah_bitop_indexed_device (fused route): the two indexed operands of bitop_tile_kernel
constexpr uint32_t kIndexedSegsPerWave = 2; // segments a wavefront of bitop_tile_kernel handles one after the other
struct BitopOperands {
    const uint32_t *comp_a, *comp_b;
    uint64_t c_words_a, c_words_b;
    const uint64_t *offs_a, *offs_b;
    uint64_t groups; // G of the bitmap both describe
    uint32_t op;     // WAH_OP_*
};
hipError_t launch_bitop_tiles(const struct CompressArgs &a, const BitopOperands &ops, hipStream_t s);

// wah_bitop_many_indexed_device: g = geometry, output and control block (its comp / seg_offsets are not used)
constexpr int kMaxBitopOperands = 8;
struct BitopManyArgs {
    SegmentsArgs g;
    const uint32_t *comp[kMaxBitopOperands];
    uint64_t c_words[kMaxBitopOperands];
    const uint64_t *offs[kMaxBitopOperands];
    int n;  // operands
    int op; // WAH_OP_*
};

// ... on operands of few words per segment: their runs merged in the compressed domain, one lane per segment
// (wah_bitop_runs.hip): count pass, scan of the tile totals, write pass
struct BitopRunsArgs {
    const uint32_t *comp[kMaxBitopOperands];
    uint64_t c_words[kMaxBitopOperands];
    const uint64_t *offs[kMaxBitopOperands];
    int n;  // operands
    int op; // WAH_OP_*
    uint64_t groups, n_segments;
    uint32_t *seg_count;  // scratch: n_segments
    uint64_t *tile_total; // scratch: one entry per tile of segments: its words, then the words in front of it
    uint32_t *temp;       // scratch: the segments' results before they are moved together: sum of c_words + 16 words
    uint32_t *out;
    uint64_t out_capacity;
    uint64_t *out_words;
    uint64_t *out_offsets; // optional, n_segments + 1 entries
    uint32_t *ctrl;
};
hipError_t launch_bitop_runs(const BitopRunsArgs &a, hipStream_t s);

// wah_bitop_device: what the operands' decodes left behind, checked on the device before the combining pass
struct PairCheck {
    const uint64_t *info_a, *info_b; // [decoded words, groups] of the two operands
    const uint32_t *ctrl_a, *ctrl_b; // their decode control blocks (error bits)
    uint64_t groups;                 // what both must have expanded to
};

// launchers (wah_compress.hip, wah_decode.hip, wah_aux.hip)
hipError_t launch_compress(const CompressArgs &a, hipStream_t s); // pair mode when a.in2 != nullptr
hipError_t launch_compress_nowait(const CompressArgs &a, hipStream_t s); // count, scan, place: nobody waits for anybody
uint32_t compress_nowait_wave_segs();
hipError_t launch_bitop_check(const uint64_t *info_a, const uint64_t *info_b, const uint32_t *ctrl_a, const uint32_t *ctrl_b, uint64_t groups,
                              uint32_t *ctrl, hipStream_t s);
hipError_t launch_decode_sums(const ScanArgs &a, hipStream_t s);
hipError_t launch_decode_expand(const ExpandArgs &a, uint64_t n_tiles, hipStream_t s);
hipError_t launch_decode_tiles(const ScanArgs &sa, const ExpandArgs &xa, uint64_t *defer, hipStream_t s); // one pass: decode_tile_kernel
// which decoder a call ran (wah_last_decode_route)
constexpr int kRouteNone = 0, kRouteOnePass = 1, kRouteTwoLaunches = 2, kRouteNoWait = 3;
constexpr uint32_t kDecodeTileWords = 2 * kScanTileWords; // ... whose workgroup tiles are this long
hipError_t launch_clear(void *p, size_t bytes, hipStream_t s);
hipError_t launch_build_index(const uint32_t *comp, uint64_t c_words, const uint64_t *tile_base, const uint64_t *info, uint64_t *offsets,
                              uint64_t capacity, uint32_t *ctrl, uint64_t n_tiles, hipStream_t s);
hipError_t launch_decode_segments(const SegmentsArgs &a, hipStream_t s);
hipError_t launch_bitop_many_segments(const BitopManyArgs &a, hipStream_t s);

// wah_merge_fills_device (after the sums pass): kept-word counts and first kept positions per tile, their scans, scatter
struct MergeArgs {
    const uint32_t *comp;
    uint64_t c_words;
    uint64_t n_tiles;
    const uint64_t *tile_base; // groups in front of every tile (sums pass)
    const uint64_t *info;      // [decoded words, groups] (sums pass)
    uint64_t *tile_kept;       // n_tiles + 1: kept words per tile, then their exclusive scan
    uint64_t *tile_first;      // n_tiles: group position of a tile's first kept word, then of the first kept word BEHIND the tile
    uint32_t *out;
    uint64_t out_capacity;
    uint64_t *out_words;
    uint32_t *ctrl;
};
hipError_t launch_merge_fills(const MergeArgs &a, hipStream_t s);
hipError_t launch_validate(const uint32_t *comp, uint64_t c_words, const uint64_t *tile_base, const uint64_t *info, uint64_t *report,
                           uint64_t n_tiles, hipStream_t s);
hipError_t launch_gen_uniform(uint32_t *out, uint64_t n, uint64_t seed, uint64_t thr, hipStream_t s);
hipError_t launch_gen_clustered(uint32_t *out, uint64_t n, uint64_t seed, uint64_t thr, hipStream_t s);
hipError_t launch_copy(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t s);

} // namespace wah
