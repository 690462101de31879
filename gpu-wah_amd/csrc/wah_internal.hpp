// wah_internal.hpp -- declarations shared by the kernel TU and the C-ABI TU.
//
// gfx950 / CDNA4 only: 64-lane wavefronts, 8 XCDs with private L2s.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wah {

// ---- wire format (reference const.h:3-16) ---------------------------------
constexpr uint32_t kOnes31 = 0x7FFFFFFFu;    // ONES31
constexpr uint32_t kFillZero = 0x80000000u;  // BIT31
constexpr uint32_t kFillOne = 0xC0000000u;   // BIT3130
constexpr uint32_t kCountMask = 0x3FFFFFFFu; // BIT30 - 1 (kernels.cu:300)

// ---- segment geometry (kernels.cu:68: one CUDA block = 1024 groups) ------
constexpr uint32_t kSegWords = 992;
constexpr uint32_t kSegGroups = 1024;
constexpr uint32_t kSteps = 16; // 64 groups (one wavefront) per step

// ---- inter-workgroup control block (uint32 words, zeroed before each launch)
constexpr uint32_t kCtlStart = 0;        // arrival ticket: order in which workgroups start running
constexpr uint32_t kCtlError = 160;      // sticky error bits
constexpr uint32_t kCtlCensus = 161;     // census mode: workgroups resident together
constexpr uint32_t kCtlWords = 256;      // 1 KiB
constexpr uint32_t kErrTimeout = 1u;     // a bounded wait expired
constexpr uint32_t kErrCapacity = 2u;    // output would exceed its capacity
constexpr uint32_t kErrStream = 4u;      // malformed compressed stream

// ---- compress geometry: one wavefront owns one segment --------------------
// worker wavefronts = segments per tile: 7 (+1 scan wave = 512 threads, 2 workgroups per CU) or
// 15 (+1 = 1024 threads, 1 workgroup per CU); WAH_WORKERS selects at run time for experiments
constexpr int kCompressWavesDefault = 15;
int compress_workers();

// ---- decode geometry -------------------------------------------------------
constexpr int kScanTileWords = 4096; // compressed words per tile (both decode passes)
constexpr int kSumTilesPerGroup = 8;  // expand tiles per workgroup tile of the sums kernel (= its worker waves)
constexpr int kExpandWaves = 4;      // wavefronts of an expand workgroup (each expands whole segments)

struct CompressArgs {
    const uint32_t *in;
    const uint32_t *in2;   // pair mode (wah_bitop_device): second bitmap of the same length, combined word by word
    uint32_t op;           // ... with WAH_OP_AND / OR / XOR / ANDNOT
    uint64_t n_words;
    uint32_t n_segments;          // ceil(G / 1024)
    uint32_t n_tiles;             // ceil(n_segments / kCompressWaves)
    uint32_t fast_segments;       // 1: input 16-byte aligned -> prefetched buffer loads; 0: scalar staging
    uint32_t last_segment_groups; // groups of the last segment (1..1024)
    uint32_t full_segments;       // n_words / 992: segments that lie wholly inside the bitmap
    uint32_t tail_bytes;          // bytes of the partial segment after them (0 if none)
    uint32_t *out;
    uint64_t out_capacity;
    uint64_t *out_words;   // device scalar: C
    uint64_t *seg_offsets; // optional, n_segments + 1 entries
    uint32_t *ctrl;        // kCtlWords
    uint32_t *gen_desc;    // generation rows: one 4-byte granule per tile (see resolve_generation)
    int census;            // 1: residency census only (see compress_grid)
};

struct ScanArgs {
    const uint32_t *comp;
    uint64_t c_words;
    uint64_t n_tiles;
    uint64_t *info;      // [0] decoded words, [1] groups
    uint64_t *tile_base; // n_tiles + 1: groups in front of each tile, last = total
    uint32_t *ctrl;
    uint32_t *gen_desc;  // generation rows (4-byte granules)
    uint64_t *big;       // 64-bit side entries for totals that do not fit a granule
    uint8_t *tile_flags; // n_tiles: 1 = the tile contains a fill word of count 0
    int aligned16;
    int census;
};

struct ExpandArgs {
    const uint32_t *comp;
    uint64_t c_words;
    uint32_t *out;
    uint64_t out_capacity;
    const uint64_t *info;
    const uint64_t *tile_base;
    const uint8_t *tile_flags; // from the sums pass: tiles with fill words of count 0
    uint32_t *ctrl;
    int aligned16;
    uint32_t parts; // workgroups that share one tile's output segments (set by the launcher)
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

// wah_bitop_indexed_device: operand A as SegmentsArgs (its `out` receives the combined decoded words), operand B beside it
struct BitopSegArgs {
    SegmentsArgs a;
    const uint32_t *comp_b;
    uint64_t c_words_b;
    const uint64_t *seg_offsets_b;
    int op; // WAH_OP_*
};

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

// wah_bitop_device: what the operands' decodes left behind, checked on the device before the combining pass
struct PairCheck {
    const uint64_t *info_a, *info_b; // [decoded words, groups] of the two operands
    const uint32_t *ctrl_a, *ctrl_b; // their decode control blocks (error bits)
    uint64_t groups;                 // what both must have expanded to
};

// launchers (wah_compress.hip, wah_decode.hip, wah_aux.hip)
hipError_t launch_compress(int workers, const CompressArgs &a, int grid, hipStream_t s);
hipError_t launch_compress_pair(const CompressArgs &a, int grid, hipStream_t s);
hipError_t launch_bitop_check(const uint64_t *info_a, const uint64_t *info_b, const uint32_t *ctrl_a, const uint32_t *ctrl_b, uint64_t groups,
                              uint32_t *ctrl, hipStream_t s);
int compress_grid(int workers, uint32_t *d_ctrl, hipStream_t s);
hipError_t launch_decode_sums(const ScanArgs &a, int grid, hipStream_t s);
int decode_sums_grid(uint32_t *d_ctrl, hipStream_t s);
hipError_t launch_decode_expand(const ExpandArgs &a, uint64_t n_tiles, hipStream_t s);
hipError_t launch_clear(void *p, size_t bytes, hipStream_t s);
hipError_t launch_build_index(const uint32_t *comp, uint64_t c_words, const uint64_t *tile_base, const uint64_t *info, uint64_t *offsets,
                              uint64_t capacity, uint32_t *ctrl, uint64_t n_tiles, hipStream_t s);
hipError_t launch_decode_segments(const SegmentsArgs &a, hipStream_t s);
hipError_t launch_bitop_segments(const BitopSegArgs &a, hipStream_t s);
hipError_t launch_bitop_many_segments(const BitopManyArgs &a, hipStream_t s);

// wah_merge_fills_device (after the sums pass): kept-word counts per tile, their scan, scatter, count fix-up
struct MergeArgs {
    const uint32_t *comp;
    uint64_t c_words;
    uint64_t n_tiles;
    const uint64_t *tile_base; // groups in front of every tile (sums pass)
    const uint64_t *info;      // [decoded words, groups] (sums pass)
    uint64_t *tile_kept;       // n_tiles + 1: kept words per tile, then their exclusive scan
    uint64_t *positions;       // group position of every kept word (c_words entries)
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
