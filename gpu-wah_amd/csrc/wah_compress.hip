// wah_compress.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the WAH path: overview, and the compress kernel.
// (decompress: wah_decode.hip; checker, merge pass, bench support: wah_aux.hip; shared helpers: wah_device.hpp)
//
// What the reference does in five kernels, two thrust scans and four blocking 8-byte D2H copies
// (compress.cu:129-166, decompress.cu:66-115, kernels.cu), is done here in
//   compress   : ONE kernel of short-lived workgroups  (reads 4N, writes 4C, nothing else)
//   decompress : streaming sums kernel + expand        (reads 4C twice, writes 4N')
// built on these CDNA4 idioms:
//   * a wavefront (64 lanes) owns a whole 1024-group segment; its 992 words are staged once in wave-private LDS
//     with 16-byte coalesced loads and re-read as 31-bit groups by a funnel shift (v_alignbit) -- the regroup of
//     kernels.cu:72-79 without idle lanes and without the shift-by-32;
//   * zero/ones classification produces 64-lane masks straight from v_cmp, "same as the next group" is one DPP
//     compare, so run detection, run lengths and the cross-warp merge (kernels.cu:126-229) collapse into a couple
//     of mask operations per 64 groups plus one v_mbcnt rank per lane;
//   * run-end words are compacted in LDS, turned into final words in REGISTERS (where they wait for the tile's output
//     offset without holding LDS bandwidth) and leave the chip as dense 256-byte stores;
//   * output offsets come from a one-hop "row scan" over 4-byte {epoch, count} granules written and read with
//     agent-scope accesses (correct across the 8 non-coherent XCD L2s), instead of thrust::exclusive_scan + moveData
//     (compress.cu:133-166, kernels.cu:273-280); nothing is persistent and nothing is cleared between launches.
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "wah_device.hpp"
#include "wah_segdecode.hpp"

namespace wah {
namespace {

// ===========================================================================
// compress: per-segment building blocks (loads, regrouping, the two classify passes, final words); the tile kernel
// that strings them together and its offset scan are described further down, at compress_tile_kernel.
// Per wave and segment in LDS: a 4 KiB STAGE buffer (the 992 input words, later the compacted run-end words in place)
// and a 2 KiB position array (kernels.cu:126-141 run ends, :188-229 merge, :244-259 final words).
// ===========================================================================
constexpr u32 kStageWords = 1024; // staged segment (992 words + look-ahead) / compacted output words (<= 1024), aliased
constexpr u32 kOutWords = kStageWords + 4; // + one dump dword (non-end lanes), kept 16-byte aligned
constexpr u32 kPosEntries = 1032; // pos[0] = -1 sentinel, pos[k+1] = group position of run end k (u16)
static_assert(kPosEntries * 2 % 16 == 0 && kPosEntries * 2 >= kSegGroups, "the position array doubles as the segment decoder's flag area");

struct Prefetch {
    u32x4 v[4];
};

// one 16-byte load; an input that is only 4-byte aligned takes four dword loads into the same registers
template <bool kAligned>
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t rsrc, u32 off) {
    // nontemporal (aux = 2): the bitmap is read exactly once.  Default-policy loads allocate it in the memory-side cache,
    // where it evicts the dirty lines the kernel before this one left behind (in a compress / decompress round trip: 1 GiB
    // of the expand kernel's writes) right into this kernel's way: round trip 1413 -> 1469 GB/s (sparse), 2080 -> 2143
    // (clustered), 1024 -> 1071 (dense) with this and the same policy on the sums pass's reads; this kernel inside the
    // round trip 0.321 -> 0.299 / 0.296 -> 0.245 / 0.412 -> 0.389 ms (isolated: 1-2 % faster)
    if (kAligned) return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 2);
    u32x4 v;
    v.x = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
    v.y = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 4u, 0, 0);
    v.z = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 8u, 0, 0);
    v.w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + 12u, 0, 0);
    return v;
}

// issue the four coalesced 16-byte loads of one segment (3968 B = 248 x 16 B; lanes 56..63 of the fourth load and
// everything past the end of the bitmap read as zero: the descriptor's bounds do the zero padding of the tail, F5)
template <bool kAligned>
__device__ __forceinline__ void prefetch_segment(const u32 *in, const CompressArgs &a, u32 seg, u32 lane, Prefetch &p) {
    // whole segments: 3968 bytes; the (one) partial segment at the end of the bitmap: what is left of it
    const u32 bytes = seg < a.full_segments ? kSegWords * 4u : a.tail_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(in + (u64)seg * kSegWords, bytes);
    const u32 off = lane * 16u;
    p.v[0] = load16<kAligned>(rsrc, off);
    p.v[1] = load16<kAligned>(rsrc, off + 1024u);
    p.v[2] = load16<kAligned>(rsrc, off + 2048u);
    p.v[3] = load16<kAligned>(rsrc, off + 3072u);
}

// pair mode: the word-by-word combination (include/wah.h: WAH_OP_*); words behind the bitmap stay zero for every op
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

// Classify + run detect + compact one segment, in three steps that the kernel interleaves for its two segments.
//   regroup   (kernels.cu:72-79): group = funnel shift of two staged words, all 16 steps' groups into registers (the
//             staging buffer is free again afterwards).
//   pass 1    classify (kernels.cu:93-112) and run ends (kernels.cu:126-141 + the cross-warp merge of :188-229): a
//             group does NOT end a run iff it is a fill and the next group of the segment has the same value:
//               next   = lane l+1 (DPP wave_shl:1); lane 63 has no source lane and keeps the `old` operand, which a
//                        wave_rol:1 of the NEXT step's register has loaded with that step's lane 0
//               z      = (x ^ next) | ((x + 1) & 0x7FFFFFFE) is zero  <=>  x is 0 or 0x7FFFFFFF AND the next group equals it
//               ends   = v_cmp_ne z, 0 -- the 64-lane mask comes out of the compare itself, into a scalar register pair
//             The group after the last one never matches, so every segment closes its last run (tests.cpp:166-172).
//             Result: 16 masks (32 scalar registers) and their population count = the words the segment compresses to
//             -- all that the other workgroups need to know of it, so the kernel publishes it right here.
//   pass 2    compact: run-end words go to LDS at rank = running count + v_mbcnt over the saved mask, with the group
//             position beside them (fill lengths are position differences, see final_words_to_regs).
// Passes 1 and 2 are generated, hand-scheduled blocks (csrc/classify_pass*.inc, tools/gen_classify_block.py).
// The bitmap's last segment may be short (F5): the groups that do not exist are replaced by a literal -- literals
// never merge, so the last real group closes its run and each of them becomes one entry BEHIND the real ones, which
// is simply not counted.
constexpr u32 kAbsentGroup = 0x2AAAAAAAu;
constexpr u32 kSparseBelow = 192; // words per segment below which pass 2 takes its step-skipping variant

struct SegGroups {
    u32 x[kSteps]; // group 64 s + lane of the segment
};
// run ends of one segment, one word per lane: bit 15 - s says whether group 64 s + lane ends a run
using SegEnds = u32;

__device__ __forceinline__ void regroup(const u32 *sp, u32 r, u32 lane_v, u32 nvalid, SegGroups &g) {
    // all 16 LDS reads, then the funnel shifts: every staged word is in registers before the buffer is reused
#pragma unroll
    for (int s = 0; s < (int)kSteps; ++s) {
        const u32 lo = sp[62 * s];
        const u32 hi = sp[62 * s + 1];
        g.x[s] = __builtin_amdgcn_alignbit(hi, lo, r) & kOnes31;
    }
    if (nvalid != kSegGroups) {
        u32 lv = lane_v; // opaque: keeps the sixteen position constants inside this cold branch
        asm volatile("" : "+v"(lv));
#pragma unroll
        for (int s = 0; s < (int)kSteps; ++s) g.x[s] = 64u * s + lv < nvalid ? g.x[s] : kAbsentGroup;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// returns the number of run ends (absent groups of a short last segment included)
__device__ __forceinline__ u32 classify_pass1(const SegGroups &g, u32 never, SegEnds &e) {
    u32 cnt = 0, st;
    u32 na, ta, nb, tb;
    u32 f;
    asm volatile("v_mov_b32 %0, 0" : "=v"(f));
    asm volatile(
#include "classify_pass1.inc"
        : [f] "+&v"(f), [cnt] "+&s"(cnt), [st] "=&s"(st), [na] "=&v"(na), [ta] "=&v"(ta), [nb] "=&v"(nb), [tb] "=&v"(tb)
        : [x0] "v"(g.x[0]), [x1] "v"(g.x[1]), [x2] "v"(g.x[2]), [x3] "v"(g.x[3]), [x4] "v"(g.x[4]), [x5] "v"(g.x[5]), [x6] "v"(g.x[6]),
          [x7] "v"(g.x[7]), [x8] "v"(g.x[8])
        : "vcc", "scc");
    asm volatile(
#include "classify_pass1.inc"
        : [f] "+&v"(f), [cnt] "+&s"(cnt), [st] "=&s"(st), [na] "=&v"(na), [ta] "=&v"(ta), [nb] "=&v"(nb), [tb] "=&v"(tb)
        : [x0] "v"(g.x[8]), [x1] "v"(g.x[9]), [x2] "v"(g.x[10]), [x3] "v"(g.x[11]), [x4] "v"(g.x[12]), [x5] "v"(g.x[13]),
          [x6] "v"(g.x[14]), [x7] "v"(g.x[15]), [x8] "v"(never)
        : "vcc", "scc");
    e = f;
    return cnt;
}

// Some emitted word is a fill iff some group is one.  Fewer words than groups: certainly.  As many words as groups
// (incompressible data): only fills of length 1 are possible, look for an all-zero / all-one group.
__device__ __forceinline__ bool segment_has_fill(const SegGroups &g, u32 ends, u32 nvalid) {
    if (ends != kSegGroups || nvalid != kSegGroups) return true;
    u32 lo = g.x[0], hi = g.x[0];
#pragma unroll
    for (int s = 1; s < (int)kSteps; s += 2) {
        lo = s + 1 < (int)kSteps ? min(lo, min(g.x[s], g.x[s + 1])) : min(lo, g.x[s]);
        hi = s + 1 < (int)kSteps ? max(hi, max(g.x[s], g.x[s + 1])) : max(hi, g.x[s]);
    }
    return __ballot(lo == 0u || hi == kOnes31) != 0;
}

// sparse: the segment compressed to few words (pass 1 counted them): most steps lie inside long fills and are skipped
__device__ __forceinline__ void classify_pass2(const SegGroups &g, SegEnds e, u32 *lds, unsigned short *pos, u32 lane_v, bool sparse) {
    // mask : the step's run ends come back out of the flag word, top bit first (v_add_co f, f, f: the carry is the mask)
    // rank : v_mbcnt pair seeded with the running count (kept in a VECTOR register: no scalar work); count += v_bcnt pair
    // write: every lane stores; lanes that end no run store to a dump slot (cheaper than masking EXEC)
    u32 count_v;
    asm volatile("v_mov_b32 %0, 0" : "=v"(count_v));
    // LDS byte addresses: value k at vbase + 4 k, its position at pbase + 2 k; k = kStageWords is the dump slot
    const u32 vbase = (u32)(uintptr_t)(lds_u32_ptr)lds;
    const u32 pbase = (u32)(uintptr_t)(lds_u16_ptr)pos + 2u;
    u32 dump_slot;
    asm volatile("v_mov_b32 %0, 0x400" : "=v"(dump_slot)); // kStageWords, in a vector register (v_cndmask cannot take a literal)
    static_assert(kStageWords == 0x400, "dump slot literal");
    u32 na, ta, nb, tb, ps;
    const u32 lane2 = lane_v * 0x10001u; // the lane id in both halves: position words are built two at a time
    u32 f = e << 16;                     // step 0 at the top
    if (sparse) {
        asm volatile(
#include "classify_pass2_skip_a.inc"
            : [f] "+&v"(f), [na] "=&v"(na), [ta] "=&v"(ta), [nb] "=&v"(nb), [tb] "=&v"(tb), [ps] "=&v"(ps), [cn] "+&v"(count_v)
            : [x0] "v"(g.x[0]), [x1] "v"(g.x[1]), [x2] "v"(g.x[2]), [x3] "v"(g.x[3]), [x4] "v"(g.x[4]), [x5] "v"(g.x[5]), [x6] "v"(g.x[6]),
              [x7] "v"(g.x[7]), [ln2] "v"(lane_v), [vb] "s"(vbase), [pb] "s"(pbase), [dm] "v"(dump_slot)
            : "vcc", "scc", "memory");
        asm volatile(
#include "classify_pass2_skip_b.inc"
            : [f] "+&v"(f), [na] "=&v"(na), [ta] "=&v"(ta), [nb] "=&v"(nb), [tb] "=&v"(tb), [ps] "=&v"(ps), [cn] "+&v"(count_v)
            : [x0] "v"(g.x[8]), [x1] "v"(g.x[9]), [x2] "v"(g.x[10]), [x3] "v"(g.x[11]), [x4] "v"(g.x[12]), [x5] "v"(g.x[13]),
              [x6] "v"(g.x[14]), [x7] "v"(g.x[15]), [ln2] "v"(lane_v), [vb] "s"(vbase), [pb] "s"(pbase), [dm] "v"(dump_slot)
            : "vcc", "scc", "memory");
        return;
    }
    asm volatile(
#include "classify_pass2_a.inc"
        : [f] "+&v"(f), [na] "=&v"(na), [ta] "=&v"(ta), [nb] "=&v"(nb), [tb] "=&v"(tb), [ps] "=&v"(ps), [cn] "+&v"(count_v)
        : [x0] "v"(g.x[0]), [x1] "v"(g.x[1]), [x2] "v"(g.x[2]), [x3] "v"(g.x[3]), [x4] "v"(g.x[4]), [x5] "v"(g.x[5]), [x6] "v"(g.x[6]),
          [x7] "v"(g.x[7]), [ln2] "v"(lane2), [vb] "s"(vbase), [pb] "s"(pbase), [dm] "v"(dump_slot)
        : "vcc", "memory");
    asm volatile(
#include "classify_pass2_b.inc"
        : [f] "+&v"(f), [na] "=&v"(na), [ta] "=&v"(ta), [nb] "=&v"(nb), [tb] "=&v"(tb), [ps] "=&v"(ps), [cn] "+&v"(count_v)
        : [x0] "v"(g.x[8]), [x1] "v"(g.x[9]), [x2] "v"(g.x[10]), [x3] "v"(g.x[11]), [x4] "v"(g.x[12]), [x5] "v"(g.x[13]),
          [x6] "v"(g.x[14]), [x7] "v"(g.x[15]), [ln2] "v"(lane2), [vb] "s"(vbase), [pb] "s"(pbase), [dm] "v"(dump_slot)
        : "vcc", "memory");
}

// ===========================================================================
// compress_tile_kernel
//
// Workgroup = one TILE of kTileWaves x kWaveSegs consecutive segments, short-lived; tile = arrival order (draw_tile).
//   every wave : for each of its kWaveSegs segments: 4 x 16-byte loads -> LDS stage -> 16 groups per lane in registers
//                -> PASS 1 (run ends counted, one flag word per lane kept) ; counts to LDS -> barrier ->
//                PASS 2 for every segment (rank + compact the run-end words in LDS) and the final words (fill length
//                = distance between consecutive run ends) parked in registers -> barrier -> dense 256-byte stores to
//                their place in the output (kernels.cu:244-259 + moveData, kernels.cu:273-280).
//   wave 0     : additionally publishes the tile's count right after the first barrier and resolves the tile's
//                output offset with the ROW SCAN below; the sweep is in flight while all waves run pass 2.
// Nothing is persistent: no residency census, nothing to clear between launches (launch epochs, below); the only
// shared counter is the arrival ticket a workgroup draws its tile number from.  A wave that waits at the barrier
// issues no instructions; the kernel is bound by the latency of the offset hop, not by instruction issue.
//
// Row scan (replaces thrust::exclusive_scan + the two blocking 8-byte reads of compress.cu:133-157).
//   granule[t]           u32 {epoch:16, words:16} of tile t, published as soon as the tile's words are counted
//   slot[s][0]           u64 {epoch:16, words:48}: words in front of superrow s  (superrow = kSuperRows rows)
//   slot[s][1 + k]       u64 {epoch:16, words:48}: words of row k of superrow s  (row = kRowTiles tiles)
// (scan area = one block per superrow: its 64 x 256 granules, then its 65 slots -- every entry has the same address and
//  the same meaning whatever the size of the bitmap, so a workspace can serve bitmaps of different sizes in turn)
// Tile (row r, index i) adds up, in ONE round trip of three loads per lane:
//   granule[r][0 .. i)  +  granule[r-1][0 .. 256)  +  slot[s][1 ..] of rows s0 .. r-2  +  slot[s][0]
// The last tile of a row publishes the row's slot as soon as its own row is complete (no dependency on anything
// older), the last tile of a superrow publishes the next superrow's slot[.][0].  So every dependency is "published
// by a tile with a smaller number" and at most one hop old; rows r-2 and older had >= one whole row of time.
// Order: tile numbers are drawn in the order in which the workgroups start running (draw_tile, wah_device.hpp), so a
// tile only ever waits for tiles that are running; every wait is bounded all the same (WAH_ERR_TIMEOUT, never a hang).
// Epochs: the workspace is never cleared.  Every launch stamps what it publishes with the launch epoch kept in the
// control block (read by every workgroup at its start, advanced by the LAST tile once its scan is complete -- by
// then every other tile has published, hence started).  A zeroed workspace is epoch 0 = "nothing valid".  When
// the 16-bit epoch is used up, the next launch has tile 0 clear the scan area while the others wait for it.
// ===========================================================================
constexpr u32 kTileWaves = (u32)kCompressTileWaves;
constexpr u32 kRowTiles = 256;             // granules per row: one 16-byte load per lane
constexpr u32 kSuperRows = 64;             // rows per superrow: one 8-byte load per lane
constexpr u32 kGranuleCountBits = 16;             // words of a tile <= 8 * 4 * 1024 (stored minus nothing: 2^15 fits 16 bits)
constexpr u32 kGranuleCountMask = (1u << kGranuleCountBits) - 1u;
constexpr u32 kSlotShift = 48;             // u64 slots: value in the low 48 bits
constexpr u64 kSlotMask = (1ull << kSlotShift) - 1ull;
static_assert(kTileWaves * 6 * kSegGroups <= kGranuleCountMask, "tile count must fit the granule");
static_assert(kEpochWrap < (1u << (32 - kGranuleCountBits)), "epochs must fit the granule");
static_assert(kRowSlots == kSuperRows + 1, "slot layout");
static_assert(kSlotShift - 32 == kGranuleCountBits, "the high half of a slot carries its epoch where a granule does");
static_assert(kScanBlockWords >= kSuperRows * kRowTiles + 2 * kRowSlots && kScanSlotsAt == kSuperRows * kRowTiles, "scan block layout");

constexpr u32 kDirectLanes = 16; // up to this many lanes with missing entries (the nearest ~64 predecessors) are simply read again

struct TileScan {
    u32x4 a, b; // granules of my row (entries below me; the descriptor cuts the rest off) and of the previous row
    u64 c;      // slot of my superrow: lane 0 = words in front of it, lane 1 + k = words of its row k
};

struct ScanGeom {
    u32 row, idx, sup, row0; // tile = row * kRowTiles + idx; superrow of the row and its first row
    u32 n_slots;             // slots to read: [0] and the rows row0 .. row - 2
    bool has_prev;           // the previous row belongs to the same superrow (else slot[0] covers it)
};

__device__ __forceinline__ ScanGeom scan_geom(u32 tile) {
    ScanGeom g;
    g.row = tile / kRowTiles;
    g.idx = tile % kRowTiles;
    g.sup = g.row / kSuperRows;
    g.row0 = g.sup * kSuperRows;
    g.has_prev = g.row > g.row0;
    g.n_slots = g.has_prev ? g.row - g.row0 : 1u;
    return g;
}

__device__ __forceinline__ void scan_issue(const CompressArgs &a, const ScanGeom &g, u32 lane, bool need_a, bool need_b, bool need_c,
                                           TileScan &p) {
    u32 *const block = a.gen_desc + (u64)g.sup * kScanBlockWords; // my superrow's granules and slots
    if (need_a) {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(block + (g.row - g.row0) * kRowTiles, g.idx * 4u);
        p.a = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 16u, 0, kAuxSc1);
    }
    if (need_b) {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(block + (g.row - 1u - g.row0) * kRowTiles, kRowTiles * 4u);
        p.b = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 16u, 0, kAuxSc1);
    }
    if (need_c) {
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(block + kScanSlotsAt, g.n_slots * 8u);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rc, lane * 8u, 0, kAuxSc1);
        p.c = ((u64)v.y << 32) | v.x;
    }
}

// final words of one compacted segment (kernels.cu:244-249), from the wave's LDS buffer into 16 registers per lane:
// word k * 64 + lane -> out[k].  Fill length = distance between consecutive run ends: a lane reads the position of ITS
// run end, the previous one comes from the lane below by DPP (lane 0: from lane 63 of the batch before, -1 in front of
// the first).  Batches of 256 words: 8 LDS reads in flight, then the arithmetic.  Words behind `count` are garbage that
// is never stored.
__device__ __forceinline__ void final_words_to_regs(const u32 *stage, const unsigned short *pos, u32 lane, u32 count, bool any_fill,
                                                    u32 (&out)[16]) {
    const u32 *const s0 = stage + lane;
    const unsigned short *const q1 = pos + 1u + lane;
    u32 prev_e = 0xFFFFu; // position "-1"
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (256u * t < count) {
            u32 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = s0[256u * t + 64u * k];
            if (any_fill) {
                u32 e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = q1[256u * t + 64u * k];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // lane 0 <- lane 63 of the batch before (wave_ror:1), lanes 1.. <- the lane below (wave_shr:1)
                    const u32 carry = (u32)__builtin_amdgcn_mov_dpp((int)prev_e, 0x13C /* wave_ror:1 */, 0xf, 0xf, true);
                    const u32 p0 = (u32)__builtin_amdgcn_update_dpp((int)carry, (int)e[k], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                    const u32 len = (e[k] - p0) & 0xFFFFu;
                    out[4 * t + k] = v[k] - 1u >= 0x7FFFFFFEu ? (((v[k] & 0x40000000u) | kFillZero) | len) : v[k];
                    prev_e = e[k];
                }
            } else { // literals only: the compacted words are the final words
#pragma unroll
                for (int k = 0; k < 4; ++k) out[4 * t + k] = v[k];
            }
        }
    }
}

// ... and from the registers to their place in the output (kernels.cu:256 + moveData, kernels.cu:273-280): dense
// 256-byte stores through a descriptor that ends with the segment's words (and with the output's capacity: words past
// it are dropped by the hardware; the last tile raises the capacity error)
__device__ __forceinline__ void emit_regs(const CompressArgs &a, u64 base, u32 count, u32 lane, const u32 (&out)[16]) {
    if (base >= a.out_capacity || count == 0u) return;
    const u64 room = a.out_capacity - base;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(a.out + base, (room < count ? (u32)room : count) * 4u);
    const u32 off = lane * 4u;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (256u * t < count) {
#pragma unroll
            // (default cache policy: nontemporal stores gave +2 % on the sparse round trip, -3 % on the clustered one)
            for (int k = 0; k < 4; ++k) __builtin_amdgcn_raw_buffer_store_b32(out[4 * t + k], rsrc, off + 256u * (4 * t + k), 0, 0);
        }
    }
}

// Wave 0 of a tile, once the tile's count is out and its sweep issued (scan_issue): the tile's output offset, and what
// the last tile of a row / of a superrow / of the launch leaves behind ("Row scan" above).  Returns the offset.
__device__ __forceinline__ u64 tile_scan_resolve(const CompressArgs &a, const ScanGeom &g, u32 *const block, const LaunchEpoch &le, u32 tile,
                                                 u32 total, u32 lane, TileScan &poll, u64 *dg_t, u32 *dg_polls) {
    const u32 epoch = le.epoch;
    (void)dg_t;
    (void)dg_polls;
    // ---- the tile's offset ---------------------------------------------------------------------------------------
    // If a few entries of the sweep are still missing (the nearest predecessors), only their lanes read again.  If
    // many are (a tile of an XCD that runs ahead of the others), the wave does NOT sweep again and again -- hundreds
    // of waiting tiles re-reading 2.5 KB each every microsecond is traffic of the order of the bitmap's: it spins on
    // ONE word, the missing entry with the highest tile number, the one that will be published last, and sweeps
    // again when that one is there.
    bool need_a = true, need_b = g.has_prev, need_c = true;
    u32 sum_a = 0, sum_b = 0;
    u64 sum_c = 0;
    u32 spins = 0;
    for (;;) {
        u32 bad_a = 0, bad_b = 0; // per lane: which of my four entries are missing
        bool bad_c = false;
        u64 ba = 0, bb = 0, bc = 0;
        if (need_a) {
            const u32 k0 = 4u * lane;
            bad_a = ((k0 < g.idx && (poll.a.x >> kGranuleCountBits) != epoch) ? 1u : 0u) | ((k0 + 1u < g.idx && (poll.a.y >> kGranuleCountBits) != epoch) ? 2u : 0u) |
                    ((k0 + 2u < g.idx && (poll.a.z >> kGranuleCountBits) != epoch) ? 4u : 0u) | ((k0 + 3u < g.idx && (poll.a.w >> kGranuleCountBits) != epoch) ? 8u : 0u);
            ba = __ballot(bad_a != 0u);
            if (ba == 0) {
                // entries at and above my index lie behind the descriptor and read as zero
                sum_a = uniform32(wave_sum32((poll.a.x & kGranuleCountMask) + (poll.a.y & kGranuleCountMask) + (poll.a.z & kGranuleCountMask) + (poll.a.w & kGranuleCountMask)));
                need_a = false;
                // my row is complete with me: its total is all that later superrow-mates need of it
                if (g.idx == kRowTiles - 1u && lane == 0)
                    __hip_atomic_store(reinterpret_cast<u64 *>(block + kScanSlotsAt) + 1u + (g.row - g.row0),
                                       ((u64)epoch << kSlotShift) | ((u64)sum_a + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (need_b) {
            bad_b = ((poll.b.x >> kGranuleCountBits) != epoch ? 1u : 0u) | ((poll.b.y >> kGranuleCountBits) != epoch ? 2u : 0u) |
                    ((poll.b.z >> kGranuleCountBits) != epoch ? 4u : 0u) | ((poll.b.w >> kGranuleCountBits) != epoch ? 8u : 0u);
            bb = __ballot(bad_b != 0u);
            if (bb == 0) {
                sum_b = uniform32(wave_sum32((poll.b.x & kGranuleCountMask) + (poll.b.y & kGranuleCountMask) + (poll.b.z & kGranuleCountMask) + (poll.b.w & kGranuleCountMask)));
                need_b = false;
            }
        }
        if (need_c) {
            // slot 0 of superrow 0 is never written: nothing lies in front of the first tile
            const bool wanted = lane < g.n_slots && !(g.sup == 0u && lane == 0u);
            bad_c = wanted && (u32)(poll.c >> kSlotShift) != epoch;
            bc = __ballot(bad_c);
            if (bc == 0) {
                sum_c = uniform64(wave_sum(wanted ? poll.c & kSlotMask : 0ull));
                need_c = false;
            }
        }
        if (!(need_a || need_b || need_c)) break;
        if (++spins > kMaxSpins) {
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrTimeout);
            break;
        }
        const u32 n_bad = (u32)__builtin_popcountll(ba) + (u32)__builtin_popcountll(bb) + (u32)__builtin_popcountll(bc);
        if (n_bad <= kDirectLanes) {
            // a few stragglers among the nearest predecessors (the usual case): read again at once, without parking on one word
            __builtin_amdgcn_s_sleep(4);
            // (by every lane: a load under a per-lane condition into registers holding the other lanes' earlier values is a
            // pattern one ROCm 7.2 build of the decoder's scan got wrong -- wah_decode.hip, sums_resolve)
            scan_issue(a, g, lane, need_a, need_b, need_c, poll);
#ifdef WAH_DIAG
            ++*dg_polls;
#endif
            continue;
        }
        // the word to wait for: {epoch, ...} in its top bits, whichever array it belongs to
        const u32 *target;
        if (need_a) {
            const u32 hl = 63u - (u32)__builtin_clzll(ba);
            const u32 km = (u32)__builtin_amdgcn_readlane((int)bad_a, (int)hl);
            target = block + (g.row - g.row0) * kRowTiles + 4u * hl + (31u - (u32)__builtin_clz(km));
        } else if (need_b) {
            const u32 hl = 63u - (u32)__builtin_clzll(bb);
            const u32 km = (u32)__builtin_amdgcn_readlane((int)bad_b, (int)hl);
            target = block + (g.row - 1u - g.row0) * kRowTiles + 4u * hl + (31u - (u32)__builtin_clz(km));
        } else {
            const u32 hl = 63u - (u32)__builtin_clzll(bc);
            target = block + kScanSlotsAt + 2u * hl + 1u; // high half of the slot
        }
        bool timed_out = false;
        for (;;) {
            __builtin_amdgcn_s_sleep(8);
            if ((__hip_atomic_load(target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> kGranuleCountBits) == epoch) break;
            if (++spins > kMaxSpins) {
                timed_out = true;
                break;
            }
        }
        if (timed_out) {
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrTimeout);
            break;
        }
        scan_issue(a, g, lane, need_a, need_b, need_c, poll);
#ifdef WAH_DIAG
        ++*dg_polls;
#endif
    }
#ifdef WAH_DIAG
    dg_t[4] = __builtin_amdgcn_s_memrealtime();
#endif
    const u64 base = sum_c + sum_b + sum_a;
    const u64 end = base + total;
    if (lane == 0) {
        if (g.idx == kRowTiles - 1u && g.row - g.row0 == kSuperRows - 1u) // last tile of a superrow
            __hip_atomic_store(reinterpret_cast<u64 *>(a.gen_desc + (u64)(g.sup + 1u) * kScanBlockWords + kScanSlotsAt),
                               ((u64)epoch << kSlotShift) | end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile == a.n_tiles - 1) {
            *a.out_words = end;
            if (a.seg_offsets) a.seg_offsets[a.n_segments] = end;
            if (end > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
            if (a.host_result) {
                // every scan of the launch can complete now (this one needed all of them), so the error word is final
                a.host_result[1] = end;
                a.host_result[0] = 1ull | ((u64)__hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32);
            }
            launch_epoch_end(a.ctrl, le); // every other tile has published, so it has read the epoch: advance it
        }
    }
    return base;
}

// Where a wave's 31-bit groups come from -- the only thing that differs between compressing a bitmap and combining
// compressed bitmaps: a SOURCE leaves the groups of one segment in registers (SegGroups) and keeps the next segment's
// loads in flight meanwhile.
//   BitmapSource  : the bitmap itself (wah_compress_device; kPair: two decoded bitmaps combined word by word, wah_bitop_device)
//   IndexedSource : the same segment of TWO indexed compressed streams, expanded and combined group by group
//                   (wah_bitop_indexed_device) -- the result is compressed without ever existing as a bitmap
template <bool kPair, bool kAligned>
struct BitmapSource {
    Prefetch pre, pre2; // pre2: pair mode only, the second bitmap's words
    __device__ __forceinline__ void begin(const CompressArgs &a, u32 seg0, u32, u32 lane) {
        if (seg0 < a.n_segments) {
            prefetch_segment<kAligned>(a.in, a, seg0, lane, pre);
            if (kPair) prefetch_segment<kAligned>(a.in2, a, seg0, lane, pre2);
        }
    }
    // segment `seg` -> g; `more`: the wave has another segment after this one
    __device__ __forceinline__ void produce(const CompressArgs &a, u32 seg, bool more, u32 nvalid, u32 *stage, unsigned short *, u32 lane,
                                            SegGroups &g) {
        if (kPair) combine_pair(pre, pre2, a.op);
        stage_prefetched(pre, stage, lane);
        // the wave re-reads other lanes' words: order the LDS traffic at wavefront scope (no barrier needed)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the next segment's loads are in flight while this one is classified
        if (more && seg + 1 < a.n_segments) {
            prefetch_segment<kAligned>(a.in, a, seg + 1, lane, pre);
            if (kPair) prefetch_segment<kAligned>(a.in2, a, seg + 1, lane, pre2);
        }
        regroup(stage + ((31u * lane) >> 5), (31u * lane) & 31u, lane, nvalid, g);
    }
};

// The wave's LDS buffer doubles as the segment decoder's areas (wah_segdecode.hpp): 4 KiB of words, 1 KiB of flags.
struct IndexedSource {
    SegmentsArgs sa, sb;         // geometry + stream of operand A / B
    const u64 *offs_a, *offs_b;  // their segment indexes
    u32 op;
    u64 oa, ob;                  // lane j: first word of segment seg0 + j in A / B (one load per operand for the whole wave)
    u32 x0[kSegBatches], x1[kSegBatches], y0[kSegBatches], y1[kSegBatches];
    SegRange ra, rb;
    bool bad = false;

    __device__ __forceinline__ void load(const CompressArgs &a, u32 j, u32 lane) {
        ra = seg_range(sa, 0, uniform64(__shfl(oa, (int)j)), uniform64(__shfl(oa, (int)j + 1)));
        rb = seg_range(sb, 0, uniform64(__shfl(ob, (int)j)), uniform64(__shfl(ob, (int)j + 1)));
        (void)a;
        seg_load_words(sa, ra, x0, x1, lane);
        seg_load_words(sb, rb, y0, y1, lane);
    }
    __device__ __forceinline__ void begin(const CompressArgs &a, u32 seg0, u32 n_segs, u32 lane) {
        // the index entries of all the wave's segments (and the one behind them) in one round trip
        const u64 i = (u64)seg0 + lane;
        const bool in = lane <= n_segs && i <= a.n_segments;
        oa = in ? offs_a[i] : 0;
        ob = in ? offs_b[i] : 0;
        if (seg0 < a.n_segments) load(a, 0, lane);
    }
    __device__ __forceinline__ void produce(const CompressArgs &a, u32 seg, bool more, u32 nvalid, u32 *stage, unsigned short *pos, u32 lane,
                                            SegGroups &g) {
        unsigned char *flag = reinterpret_cast<unsigned char *>(pos);
        ra.nvalid = rb.nvalid = nvalid;
        bool ok = seg_mark(ra, x0, x1, flag, stage, lane);
        u32 ga[kSteps];
        {
            const uint4 fq = reinterpret_cast<const uint4 *>(flag)[lane];
            const u32 f[4] = {fq.x, fq.y, fq.z, fq.w};
            u32 before = 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s < (int)kSteps; ++s) ga[s] = seg_group(s, f, before, stage, ok ? ra.cnt : 1u, nvalid, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // A's words and flags have been read: the areas go to B
        ok = seg_mark(rb, y0, y1, flag, stage, lane) && ok;
        const u32 cnt_b = ok ? rb.cnt : 1u;
        // the next segment's words are in flight while this one is expanded and classified
        const u32 j = seg - (seg / kIndexedSegsPerWave) * kIndexedSegsPerWave; // position inside the wave's run of segments
        if (more && seg + 1 < a.n_segments) load(a, j + 1, lane);
        bad |= !ok;
        // any of the four operations (include/wah.h: WAH_OP_AND 0, OR 1, XOR 2, ANDNOT 3) as a sum of minterms; the
        // masks are wave-uniform
        const u32 k_ab = op <= 1 ? ~0u : 0u;
        const u32 k_a_nb = op == 0 ? 0u : ~0u;
        const u32 k_na_b = op == 1 || op == 2 ? ~0u : 0u;
        const uint4 fq = reinterpret_cast<const uint4 *>(flag)[lane];
        const u32 f[4] = {fq.x, fq.y, fq.z, fq.w};
        u32 before = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < (int)kSteps; ++s) {
            const u32 gb = seg_group(s, f, before, stage, cnt_b, nvalid, lane);
            g.x[s] = ((ga[s] & gb & k_ab) | (ga[s] & ~gb & k_a_nb) | (~ga[s] & gb & k_na_b)) & kOnes31;
        }
        if (nvalid != kSegGroups) { // the bitmap's last, short segment: absent groups become throw-away literals (regroup())
            u32 lv = lane;
            asm volatile("" : "+v"(lv));
#pragma unroll
            for (int s = 0; s < (int)kSteps; ++s) g.x[s] = 64u * s + lv < nvalid ? g.x[s] : kAbsentGroup;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the areas are written again by pass 2 / the next segment
    }
};

// kWaveSegs: segments a wavefront compresses one after the other (the tile = kTileWaves x kWaveSegs segments).  Large
// bitmaps take 5: a wave's four idle stretches (first load, barrier, offset, store drain) are paid once per five
// segments, and 16 waves per CU keep 80 segments in flight.  Small bitmaps take 1 or 2: more, shorter tiles.
//
// kMode: kTileScan is the kernel described above.  kTileCount and kTilePlace are the two halves of the NO-WAIT route
// (wah_compress_device_ex, WAH_NO_WAIT; include/wah.h): three launches in which no workgroup ever waits for another --
// count (pass 1 only: the tile's word count to a table), tile_offsets_kernel (exclusive scan of the table), place
// (the whole tile again, its offset out of the table).  The bitmap is read twice; tile = blockIdx, no ticket, no epoch.
enum : int { kTileScan = 0, kTileCount = 1, kTilePlace = 2 };
template <class Source, u32 kWaveSegs, int kMode = kTileScan>
__device__ __forceinline__ void compress_tile_body(const CompressArgs &a, Source &src) {
    __shared__ __attribute__((aligned(16))) u32 s_out[kTileWaves][kOutWords];
    __shared__ __attribute__((aligned(16))) unsigned short s_pos[kTileWaves][kPosEntries];
    __shared__ u32 s_count[kTileWaves];
    __shared__ u32 s_prefix[kTileWaves];
    __shared__ u64 s_base;
    __shared__ u32 s_tile;

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    const u32 tile = kMode == kTileScan ? draw_tile(a.ctrl, &s_tile) : blockIdx.x;
    const u32 seg0 = (tile * kTileWaves + wave) * kWaveSegs; // this wave's segments: seg0 .. seg0 + kWaveSegs - 1
#ifdef WAH_DIAG
    u64 dg_t[8];
    u32 dg_polls = 0;
    dg_t[0] = __builtin_amdgcn_s_memrealtime();
#define DG(i) dg_t[i] = __builtin_amdgcn_s_memrealtime()
#else
#define DG(i)
#endif

    // ---- launch epoch (wah_device.hpp): the same value for every workgroup of the launch --------------------------------
    LaunchEpoch le = {};
    if (kMode == kTileScan) {
        le = launch_epoch_begin(a.ctrl, tile, a.n_tiles, a.gen_desc, a.scan_words, a.keep_error);
        if (le.bad) {
            if (blockIdx.x == 0 && threadIdx.x == 0) *a.out_words = 0;
            return;
        }
    } else if (kMode == kTileCount && tile == 0 && threadIdx.x == 0 && !a.keep_error) {
        // a new launch: forget the previous one's status (nothing of this route raises an error before tile_offsets_kernel)
        __hip_atomic_store(a.ctrl + kCtlError, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const u32 epoch = le.epoch;

    // ---- this wave's segments, one after the other through the wave's LDS buffer; the final words of each are parked
    //      in 16 registers per lane, so nothing of a tile that waits for its offset occupies LDS bandwidth or needs a
    //      second LDS pass, and twice as many segments are in flight per CU as there are LDS buffers -------------------
    u32 *const stage = s_out[wave];
    unsigned short *const pos = s_pos[wave];
    u32 out[kWaveSegs][16];
    u32 cnt[kWaveSegs];
    SegGroups grp[kWaveSegs];
    SegEnds ends[kWaveSegs];
    u32 nval[kWaveSegs];
    u32 never;
    asm volatile("v_mov_b32 %0, -1" : "=v"(never)); // "group after the last one": a value no 31-bit group can equal
    src.begin(a, seg0, kWaveSegs, lane);
    // ---- pass 1 of all the wave's segments: nothing but their word counts, which is all the other workgroups wait for.
    //      What follows the publication (pass 2, final words) has the round trip of the sweep to hide in ----------------
#pragma unroll
    for (u32 j = 0; j < kWaveSegs; ++j) {
        const u32 seg = seg0 + j;
        cnt[j] = 0;
        nval[j] = kSegGroups;
        if (seg < a.n_segments) {
            nval[j] = (seg == a.n_segments - 1) ? a.last_segment_groups : kSegGroups;
            src.produce(a, seg, j + 1 < kWaveSegs, nval[j], stage, pos, lane, grp[j]);
            if (j == 0) DG(1);
            cnt[j] = classify_pass1(grp[j], never, ends[j]) - (kSegGroups - nval[j]);
        }
    }
    u32 count = 0;
#pragma unroll
    for (u32 j = 0; j < kWaveSegs; ++j) count += cnt[j];
    if (lane == 0) s_count[wave] = count;
    DG(2);
    __syncthreads();
    DG(3);

    // ---- wave 0: the tile's count goes out, the sweep of the others' counts is issued --------------------------------
    const ScanGeom g = scan_geom(tile);
    u32 *const block = a.gen_desc + (u64)g.sup * kScanBlockWords;
    TileScan poll = {};
    u32 total = 0;
    if (wave == 0) {
        const u32 mine = lane < kTileWaves ? s_count[lane] : 0u;
        const u32 incl = wave_scan_incl32(mine);
        if (lane < kTileWaves) s_prefix[lane] = incl - mine;
        total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (kMode == kTileCount) {
            if (lane == 0) a.tile_counts[tile] = total;
        } else if (kMode == kTilePlace) {
            if (lane == 0) s_base = a.tile_counts[tile]; // (an offset by now: tile_offsets_kernel)
        } else {
            if (lane == 0)
                __hip_atomic_store(block + (g.row - g.row0) * kRowTiles + g.idx, (epoch << kGranuleCountBits) | total, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef WAH_DIAG
        if (a.tune == 77u) { // time line mode: how long does the sweep itself take?
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            DG(5);
        }
#endif
    }

    if (kMode == kTileCount) return;

    // ---- pass 2 of all the wave's segments: compaction in LDS, final words into registers -------------------------------------
#pragma unroll
    for (u32 j = 0; j < kWaveSegs; ++j) {
        if (seg0 + j < a.n_segments) {
            if (lane == 0) pos[0] = 0xFFFFu; // position "-1": the run before the first one ends there
            classify_pass2(grp[j], ends[j], stage, pos, lane, cnt[j] < kSparseBelow);
            const bool any_fill = segment_has_fill(grp[j], cnt[j] + (kSegGroups - nval[j]), nval[j]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            final_words_to_regs(stage, pos, lane, cnt[j], any_fill, out[j]);
            // the buffer is written again next: order these reads before those writes
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }

    DG(6);
    if (kMode == kTileScan && wave == 0) {
        // the sweep of the other tiles' counts: one round trip of three loads per lane, issued only now (see
        // compress_pair_body: issued right behind the publication it mostly finds the nearest predecessors missing)
        scan_issue(a, g, lane, true, g.has_prev, true, poll);
#ifdef WAH_DIAG
        const u64 base = tile_scan_resolve(a, g, block, le, tile, total, lane, poll, dg_t, &dg_polls);
#else
        const u64 base = tile_scan_resolve(a, g, block, le, tile, total, lane, poll, nullptr, nullptr);
#endif
        if (lane == 0) s_base = base;
    }
    __syncthreads();
#ifdef WAH_DIAG
    if (wave == 0 && lane == 0 && a.seg_offsets && (a.tune == 77u || a.tune == 78u)) { // per-tile time line (tools/tile_timeline.py)
        a.seg_offsets[(u64)tile * 8 + 0] = dg_t[0]; // start
        a.seg_offsets[(u64)tile * 8 + 1] = dg_t[3]; // counts known (barrier 1 passed) = publish
        a.seg_offsets[(u64)tile * 8 + 2] = dg_t[5]; // first sweep returned
        a.seg_offsets[(u64)tile * 8 + 3] = dg_t[4]; // offset known
        a.seg_offsets[(u64)tile * 8 + 4] = dg_polls;
        a.seg_offsets[(u64)tile * 8 + 5] = dg_t[6]; // pass 2 + final words done
        a.seg_offsets[(u64)tile * 8 + 6] = __builtin_amdgcn_s_memrealtime(); // barrier 2 passed
    }
    if (wave == 0 && lane == 0 && tile % 67u == 0u) { // a sample: the atomics must not become the bottleneck
        unsigned long long *d = reinterpret_cast<unsigned long long *>(a.ctrl + 192);
        atomicAdd(d + 0, (unsigned long long)(dg_t[1] - dg_t[0])); // loads -> staged
        atomicAdd(d + 1, (unsigned long long)(dg_t[2] - dg_t[1])); // classify
        atomicAdd(d + 2, (unsigned long long)(dg_t[3] - dg_t[2])); // barrier 1 (slowest wave of the tile)
        atomicAdd(d + 3, (unsigned long long)(dg_t[4] - dg_t[3])); // scan
        atomicAdd(d + 4, (unsigned long long)dg_polls);
        atomicAdd(d + 5, 1ull);
        atomicAdd(d + 6, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - dg_t[0]));
        u32 xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        atomicAdd(d + 8 + xcc, (unsigned long long)(dg_t[4] - dg_t[3]));  // scan (incl. pass 2) by XCD
        atomicAdd(d + 16 + xcc, (unsigned long long)dg_polls);            // re-polls by XCD
        atomicAdd(d + 24 + xcc, 1ull);                                     // sampled tiles by XCD
        if (xcc != (tile & 7u)) atomicAdd(d + 7, 1ull);                    // tiles NOT on XCD blockIdx % 8
    }
#endif

#ifdef WAH_DIAG
    if (a.seg_offsets && (a.tune == 77u || a.tune == 78u)) return; // time line mode: the index buffer holds the stamps
#endif
    // ---- the parked words to their place ---------------------------------------------------------------------------
    u64 base = uniform64(s_base) + uniform32(s_prefix[wave]);
#pragma unroll
    for (u32 j = 0; j < kWaveSegs; ++j) {
        const u32 seg = seg0 + j;
        if (seg < a.n_segments) {
            if (lane == 0 && a.seg_offsets) a.seg_offsets[seg] = base;
            emit_regs(a, base, cnt[j], lane, out[j]);
            base += cnt[j];
        }
    }
}

constexpr u32 kNoWaitWaveSegs = 4; // segments per wave of the no-wait routes on the pair body (two pairs)

#include "wah_compress_pair.inc"

// ===========================================================================
// Unsegmented ("classic") WAH straight out of the encoder (wah_compress_device_ex, WAH_UNSEGMENTED):
// the stream that wah_merge_fills_device makes of compress()'s -- a fill may cross the 1024-group cut of
// kernels.cu:68,188-229 (but not a multiple of 2^29 groups, so that every count fits 30 bits).
//
// The unit is the PAIR of segments a wavefront classifies at once (compress_unseg_pair_body, wah_compress_unseg_pair.inc).
// Where the pairs' cut falls inside a run, pair s + 1 begins with the same fill group pair s ends with: a LOCAL property
// of two groups.  The run's word is written by the pair in which the run ENDS; every earlier piece is dropped:
//   drop_s   = the trailing fill of pair s continues into s + 1           -> pair s emits one word less (its last)
//   merge_s  = the leading fill of pair s continues one from s - 1        -> its first word's count grows by carry_s
// so the word counts stay ADDITIVE (count_s - drop_s) and the offsets come from the same row scan.  What crosses tiles
// is the length of the run that is open at a tile's end -- and it only passes THROUGH tiles that are one single run
// ("transparent", T): a tile publishes (T, L) beside its count, L = length of its trailing run (if T: its group count),
//   carry_in(t) = L_j + sum of L over the transparent tiles between j and t,  j = nearest non-transparent tile before t
// found in the same sweep (8-byte granules {epoch:16, count:16, T:1, L:17}; rows and superrows carry (T, L) in a second
// slot array; the superrow prefix is an absolute length, which ends every walk).
// ===========================================================================
constexpr u32 kCutSegs = 1u << 19;                  // segments per block of 2^29 groups
constexpr u64 kUnsegT = 1ull << 31;                 // granule: the tile is one single run that continues the one before it
constexpr u64 kUnsegLMask = (1ull << 17) - 1ull;    // granule: length of the tile's trailing run
constexpr u64 kSlotT = 1ull << 47;                  // slot B: the row is transparent
constexpr u64 kSlotLMask = (1ull << 47) - 1ull;
static_assert(kTileWaves * 3u * 2u * kSegGroups <= kUnsegLMask, "a tile's groups (three pairs per wave) must fit the granule's L");
static_assert(kUnsegBlockWords >= kUnsegSlotsBAt + 2 * kRowSlots && kUnsegSlotsAAt == 2 * kSuperRows * kRowTiles, "unsegmented scan block layout");

__device__ __forceinline__ bool is_fill_group(u32 v) { return v == 0u || v == kOnes31; }

struct UnsegSweep {
    u32x4 a[2], b[2]; // granules of my row (entries below me) and of the previous row: four per lane
    u64 ca, cb;       // slots A (words) and B (T, L) of my superrow: lane 0 = the prefix, lane 1 + k = row k
};

__device__ __forceinline__ void unseg_issue(u32 *block, u32 row_in_super, u32 idx, u32 n_slots, u32 lane, bool need_a, bool need_b,
                                            bool need_c, UnsegSweep &p) {
    if (need_a) {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(block + (u64)row_in_super * kRowTiles * 2u, idx * 8u);
        p.a[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 32u, 0, kAuxSc1);
        p.a[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, lane * 32u + 16u, 0, kAuxSc1);
    }
    if (need_b) {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(block + (u64)(row_in_super - 1u) * kRowTiles * 2u, kRowTiles * 8u);
        p.b[0] = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 32u, 0, kAuxSc1);
        p.b[1] = __builtin_amdgcn_raw_buffer_load_b128(rb, lane * 32u + 16u, 0, kAuxSc1);
    }
    if (need_c) {
        const __amdgpu_buffer_rsrc_t rca = make_rsrc(block + kUnsegSlotsAAt, n_slots * 8u);
        const __amdgpu_buffer_rsrc_t rcb = make_rsrc(block + kUnsegSlotsBAt, n_slots * 8u);
        const u32x2 va = __builtin_amdgcn_raw_buffer_load_b64(rca, lane * 8u, 0, kAuxSc1);
        const u32x2 vb = __builtin_amdgcn_raw_buffer_load_b64(rcb, lane * 8u, 0, kAuxSc1);
        p.ca = ((u64)va.y << 32) | va.x;
        p.cb = ((u64)vb.y << 32) | vb.x;
    }
}

// Fold of up to four (T, L) entries per lane, entry k of lane l having sequence number 4 l + k, walked from the RIGHT:
// returns whether all of them are transparent, and the sum of L from the nearest non-transparent entry (included) on.
__device__ __forceinline__ bool fold_right(const bool (&valid)[4], const bool (&t)[4], const u64 (&len)[4], u32 lane, u64 &sum) {
    u32 stop = 0; // 1 + index of my highest non-transparent entry
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (valid[k] && !t[k]) stop = (u32)k + 1u;
    const u64 m = __ballot(stop != 0u);
    u64 part = 0;
    if (m == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) part += valid[k] ? len[k] : 0ull;
        sum = uniform64(wave_sum(part));
        return true;
    }
    const u32 hl = 63u - (u32)__builtin_clzll(m);
    const u32 kk = (u32)__builtin_amdgcn_readlane((int)stop, (int)hl) - 1u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (valid[k] && (lane > hl || (lane == hl && (u32)k >= kk))) part += len[k];
    sum = uniform64(wave_sum(part));
    return false;
}

// (T, L) of the tile's waves folded by the lanes of wave 0 (lane w = wave w; lanes behind the last wave: transparent,
// length 0) instead of one wave after the other -- sixteen dependent LDS reads on the way to the tile's publication, and
// sixteen more in front of its second barrier.
//   unseg_fold_tile:    the tile's own (T, L)
//   unseg_wave_carries: lane w <- length of the run that is open where wave w begins, given the one open where the tile
//                       begins (c_in); returns the one open where the tile ends
__device__ __forceinline__ void unseg_fold_tile(const u32 *s_t, const u32 *s_l, u32 lane, u32 &tile_t, u32 &tile_l) {
    const u32 t = lane < kTileWaves ? s_t[lane] : 1u, l = lane < kTileWaves ? s_l[lane] : 0u;
    const u64 non_t = __ballot(t == 0u);
    tile_t = non_t == 0 ? 1u : 0u;
    const u32 from = non_t == 0 ? 0u : 63u - (u32)__builtin_clzll(non_t); // the last wave that is not transparent
    tile_l = uniform32(wave_sum32(lane >= from ? l : 0u));
}
__device__ __forceinline__ u64 unseg_wave_carries(const u32 *s_t, const u32 *s_l, u32 *s_carry, u64 c_in, u32 lane) {
    const u32 t = lane < kTileWaves ? s_t[lane] : 1u, l = lane < kTileWaves ? s_l[lane] : 0u;
    const u64 non_t = __ballot(t == 0u);
    const u32 excl = wave_scan_incl32(l) - l;             // lengths of the waves below me
    const u64 below = non_t & ((1ull << lane) - 1ull);    // waves below me that are not transparent
    const u32 v = below ? 63u - (u32)__builtin_clzll(below) : 0u;
    const u32 excl_v = (u32)__shfl((int)excl, (int)v);
    const u64 c = below ? (u64)(excl - excl_v) : c_in + excl;
    if (lane < kTileWaves) s_carry[lane] = (u32)c;
    return uniform64(__shfl(c, (int)kTileWaves));         // ("wave 8": behind the tile's last wave)
}

// Wave 0 of a tile of the unsegmented mode (compress_unseg_pair_body), after the tile's granule
// {words, (T, L)} has gone out: the tile's offset and the length of the run that is open where it begins (the sweep of the
// other tiles' granules is issued only here, late: compress_pair_body), the carries of the tile's waves (s_carry), what a
// row's or superrow's last tile publishes, and what the launch's last tile leaves behind.
__device__ __forceinline__ void unseg_tile_resolve(const CompressArgs &a, const ScanGeom &g, u32 *block, u64 *my_row, const LaunchEpoch &le, u32 tile,
                                                   u32 total, u32 tile_t, u32 tile_l, u32 lane, const u32 *s_t, const u32 *s_l, u32 *s_carry,
                                                   u64 *s_base) {
    const u32 epoch = le.epoch;
    UnsegSweep poll = {};
    unseg_issue(block, g.row - g.row0, g.idx, g.n_slots, lane, true, g.has_prev, true, poll);
    bool need_a = true, need_b = g.has_prev, need_c = true;
    u64 words_a = 0, words_b = 0, words_c = 0, len_a = 0, len_b = 0, len_c = 0;
    bool all_a = true, all_b = true;
    u32 spins = 0;
    auto granule = [](const u32x4 &q, int h) { return ((u64)(h ? q.w : q.y) << 32) | (h ? q.z : q.x); };
    u64 *const slots_a = reinterpret_cast<u64 *>(block + kUnsegSlotsAAt);
    u64 *const slots_b = reinterpret_cast<u64 *>(block + kUnsegSlotsBAt);
    for (;;) {
        u64 ba = 0, bb = 0, bc = 0;
        u32 bad_a = 0, bad_b = 0;
        bool bad_ca = false, bad_cb = false;
        if (need_a) {
            bool valid[4], t[4];
            u64 len[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u64 gk = granule(poll.a[k >> 1], k & 1);
                valid[k] = 4u * lane + k < g.idx;
                if (valid[k] && (u32)(gk >> 48) != epoch) bad_a |= 1u << k;
                t[k] = (gk & kUnsegT) != 0;
                len[k] = gk & kUnsegLMask;
                sum += (gk >> 32) & 0xFFFFull; // entries at and above my index lie behind the descriptor and read as zero
            }
            ba = __ballot(bad_a != 0u);
            if (ba == 0) {
                words_a = uniform64(wave_sum(sum));
                all_a = fold_right(valid, t, len, lane, len_a);
                need_a = false;
                if (g.idx == kRowTiles - 1u && lane == 0) { // my row is complete with me: its words and its (T, L)
                    __hip_atomic_store(slots_a + 1u + (g.row - g.row0), ((u64)epoch << 48) | (words_a + total), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    const u64 row_tl = tile_t ? ((all_a ? kSlotT : 0ull) | (len_a + tile_l)) : (u64)tile_l;
                    __hip_atomic_store(slots_b + 1u + (g.row - g.row0), ((u64)epoch << 48) | row_tl, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (need_b) {
            bool valid[4], t[4];
            u64 len[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u64 gk = granule(poll.b[k >> 1], k & 1);
                valid[k] = true;
                if ((u32)(gk >> 48) != epoch) bad_b |= 1u << k;
                t[k] = (gk & kUnsegT) != 0;
                len[k] = gk & kUnsegLMask;
                sum += (gk >> 32) & 0xFFFFull;
            }
            bb = __ballot(bad_b != 0u);
            if (bb == 0) {
                words_b = uniform64(wave_sum(sum));
                all_b = fold_right(valid, t, len, lane, len_b);
                need_b = false;
            }
        }
        if (need_c) {
            // slots 0 of superrow 0 are never written: nothing lies in front of the first tile
            const bool wanted = lane < g.n_slots && !(g.sup == 0u && lane == 0u);
            bad_ca = wanted && (u32)(poll.ca >> 48) != epoch;
            bad_cb = wanted && (u32)(poll.cb >> 48) != epoch;
            bc = __ballot(bad_ca || bad_cb);
            if (bc == 0) {
                words_c = uniform64(wave_sum(wanted ? poll.ca & ((1ull << 48) - 1ull) : 0ull));
                // walk from the right: lane 0 (the superrow prefix, an absolute length) ends it at the latest
                const bool stop = lane < g.n_slots && (lane == 0u || !(poll.cb & kSlotT));
                const u32 hl = 63u - (u32)__builtin_clzll(__ballot(stop));
                len_c = uniform64(wave_sum(wanted && lane >= hl ? poll.cb & kSlotLMask : 0ull));
                need_c = false;
            }
        }
        if (!(need_a || need_b || need_c)) break;
        if (++spins > kMaxSpins) {
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrTimeout);
            break;
        }
        if ((u32)__builtin_popcountll(ba) + (u32)__builtin_popcountll(bb) + (u32)__builtin_popcountll(bc) <= kDirectLanes) {
            // a few stragglers among the nearest predecessors (the usual case): read again at once, without parking on one word
            __builtin_amdgcn_s_sleep(4);
            unseg_issue(block, g.row - g.row0, g.idx, g.n_slots, lane, need_a, need_b, need_c, poll);
            continue;
        }
        // wait for the missing entry with the highest tile number, then read what is missing again
        const u64 *target;
        if (need_a) {
            const u32 hl = 63u - (u32)__builtin_clzll(ba);
            const u32 km = (u32)__builtin_amdgcn_readlane((int)bad_a, (int)hl);
            target = my_row + 4u * hl + (31u - (u32)__builtin_clz(km));
        } else if (need_b) {
            const u32 hl = 63u - (u32)__builtin_clzll(bb);
            const u32 km = (u32)__builtin_amdgcn_readlane((int)bad_b, (int)hl);
            target = my_row - kRowTiles + 4u * hl + (31u - (u32)__builtin_clz(km));
        } else {
            const u32 hl = 63u - (u32)__builtin_clzll(bc);
            target = (uniform32(__shfl((u32)bad_ca, (int)hl)) ? slots_a : slots_b) + hl;
        }
        bool timed_out = false;
        for (;;) {
            __builtin_amdgcn_s_sleep(8);
            if ((u32)(__hip_atomic_load(target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 48) == epoch) break;
            if (++spins > kMaxSpins) {
                timed_out = true;
                break;
            }
        }
        if (timed_out) {
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrTimeout);
            break;
        }
        unseg_issue(block, g.row - g.row0, g.idx, g.n_slots, lane, need_a, need_b, need_c, poll);
    }
    const u64 base = words_c + words_b + words_a;
    const u64 end = base + total;
    // the run that is open in front of this tile: through my row, the previous one, the older rows, the prefix
    u64 carry = len_a;
    if (all_a) carry += g.has_prev ? (all_b ? len_b + len_c : len_b) : len_c;
    const u64 c = unseg_wave_carries(s_t, s_l, s_carry, carry, lane);
    if (lane == 0) {
        *s_base = base;
        if (g.idx == kRowTiles - 1u && g.row - g.row0 == kSuperRows - 1u) { // last tile of a superrow: the next one's prefix
            u32 *const nb = a.unseg_desc + (u64)(g.sup + 1u) * kUnsegBlockWords;
            __hip_atomic_store(reinterpret_cast<u64 *>(nb + kUnsegSlotsAAt), ((u64)epoch << 48) | end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<u64 *>(nb + kUnsegSlotsBAt), ((u64)epoch << 48) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tile == a.n_tiles - 1) {
            *a.out_words = end;
            if (end > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
            if (a.host_result) {
                a.host_result[1] = end;
                a.host_result[0] = 1ull | ((u64)__hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32);
            }
            launch_epoch_end(a.ctrl, le);
        }
    }
}

// {words, (T, L)} of every tile -> {first word, length of the run that is open where the tile begins}, in place; + what
// the last tile of the scan route leaves behind.  ONE wavefront: the (T, L) of consecutive tiles fold like
//   (T1, L1) . (T2, L2) = (T1 & T2, T2 ? L1 + L2 : L2)
// which is associative, so 64 tiles at a time go through a wave scan and the carry of the batches before them is applied
// on top (a 1 GiB bitmap has 17 000 tiles of 16 segments).
__global__ __launch_bounds__(64) void unseg_offsets_kernel(const CompressArgs a) {
    const u32 lane = lane_id();
    u64 words_front = 0, run_front = 0; // in front of the batch: words, length of the open run
    for (u64 t0 = 0; t0 < a.n_tiles; t0 += 64u) {
        const u64 t = t0 + lane;
        const bool in = t < a.n_tiles;
        const u64 cnt = in ? a.tile_counts[2 * t] : 0ull;
        const u64 tl = in ? a.tile_counts[2 * t + 1] : kSlotT; // (behind the end: transparent, length 0)
        u64 wsum = cnt, len = tl & kSlotLMask;
        bool tr = (tl & kSlotT) != 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { // inclusive scan of both
            const u64 w2 = __shfl_up(wsum, off), l2 = __shfl_up(len, off);
            const bool t2 = __shfl_up((int)tr, off) != 0;
            if (lane >= (u32)off) {
                wsum += w2;
                if (tr) len += l2;
                tr = tr && t2;
            }
        }
        // exclusive values = the inclusive ones of the lane below, on top of what lies in front of the batch
        const u64 w_ex = __shfl_up(wsum, 1), l_ex = __shfl_up(len, 1);
        const bool t_ex = __shfl_up((int)tr, 1) != 0;
        const u64 my_words = words_front + (lane ? w_ex : 0ull);
        const u64 my_run = lane ? (t_ex ? run_front + l_ex : l_ex) : run_front;
        if (in) {
            a.tile_counts[2 * t] = my_words;
            a.tile_counts[2 * t + 1] = my_run;
        }
        const u64 w_all = __shfl(wsum, 63), l_all = __shfl(len, 63);
        const bool t_all = __shfl((int)tr, 63) != 0;
        words_front += w_all;
        run_front = t_all ? run_front + l_all : l_all;
    }
    if (lane == 0) {
        const u64 end = words_front;
        *a.out_words = end;
        if (end > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
        if (a.host_result) {
            a.host_result[1] = end;
            a.host_result[0] = 1ull | ((u64)__hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32);
        }
    }
}

#include "wah_compress_unseg_pair.inc"

template <bool kPair, bool kAligned, u32 kWaveSegs>
__global__ __launch_bounds__(kTileWaves * 64, kWaveSegs <= 2 ? 6 : 4) void compress_tile_kernel(const CompressArgs a) {
    BitmapSource<kPair, kAligned> src;
    compress_tile_body<BitmapSource<kPair, kAligned>, kWaveSegs>(a, src);
}

// ---- the no-wait route of the pair mode (wah_bitop_device's combining compress): two segments per wave ----------------
template <int kMode>
__global__ __launch_bounds__(kTileWaves * 64, 4) void compress_tile_pair_nowait_kernel(const CompressArgs a) {
    BitmapSource<true, true> src;
    compress_tile_body<BitmapSource<true, true>, 2, kMode>(a, src);
}

// ---- the no-wait route (kTileCount / kTilePlace of compress_pair_body): two pairs per wave, whatever the size of the bitmap ------
template <bool kAligned, int kMode>
__global__ __launch_bounds__(kTileWaves * 64, 4) void compress_nowait_kernel(const CompressArgs a) {
    __shared__ __attribute__((aligned(1024))) u32 s_stage[kTileWaves][kPairStageWords]; // (the swizzle is made of address bits 7-9)
    __shared__ u32 s_count[kTileWaves];
    __shared__ u32 s_prefix[kTileWaves];
    __shared__ u64 s_base;
    const PairShared sm = {s_stage, s_count, s_prefix, &s_base};
    const LaunchEpoch le = {};
    compress_pair_body<kAligned, kNoWaitWaveSegs / 2, kMode>(a, sm, blockIdx.x, blockIdx.x * (kTileWaves * (kNoWaitWaveSegs / 2)), le);
}

// counts of the tiles -> where every tile's words start (exclusive scan, in place), + everything the last tile of the
// scan route leaves behind: C, the index's last entry, the capacity check, the host's copy of the result.  One workgroup:
// the table has one entry per 16 segments (64 KB of bitmap), a 1 GiB bitmap has 17 000 of them.
__global__ __launch_bounds__(1024) void tile_offsets_kernel(const CompressArgs a) {
    __shared__ u64 s_wave[16];
    __shared__ u64 s_carry;
    const u32 lane = lane_id(), wave = wave_id();
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u32 t0 = 0; t0 < a.n_tiles; t0 += 1024u) {
        const u32 t = t0 + threadIdx.x;
        const u64 mine = t < a.n_tiles ? a.tile_counts[t] : 0ull;
        const u64 incl = wave_scan_incl(mine, lane);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        u64 front = s_carry;
        for (u32 w = 0; w < wave; ++w) front += s_wave[w];
        if (t < a.n_tiles) a.tile_counts[t] = front + incl - mine;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = front + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const u64 end = s_carry;
        *a.out_words = end;
        if (a.seg_offsets) a.seg_offsets[a.n_segments] = end;
        if (end > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
        if (a.host_result) {
            a.host_result[1] = end;
            a.host_result[0] = 1ull | ((u64)__hip_atomic_load(a.ctrl + kCtlError, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32);
        }
    }
}

// wah_bitop_indexed_device: compress(A op B) straight from the two indexed streams.  Same tile kernel; the groups come
// from the segment decoder instead of the bitmap.  Nothing of bitmap size is written or read: traffic = 4 C_A + 4 C_B
// + 4 C_out (+ the indexes).  A range that is not exactly its segment (not a stream of compress() for this bitmap) is
// reported as WAH_ERR_STREAM; the output is then undefined but every access stays inside the buffers.
template <int kMode>
__global__ __launch_bounds__(kTileWaves * 64, 4) void bitop_tile_kernel(const CompressArgs a, const BitopOperands ops) {
    IndexedSource src;
    src.sa = SegmentsArgs{};
    src.sa.comp = ops.comp_a;
    src.sa.c_words = ops.c_words_a;
    src.sa.groups = ops.groups;
    src.sb = src.sa;
    src.sb.comp = ops.comp_b;
    src.sb.c_words = ops.c_words_b;
    src.offs_a = ops.offs_a;
    src.offs_b = ops.offs_b;
    src.op = ops.op;
    compress_tile_body<IndexedSource, kIndexedSegsPerWave, kMode>(a, src);
    if (kMode != kTileCount && __any(src.bad) && lane_id() == 0) atomicOr(a.ctrl + kCtlError, kErrStream);
}

} // namespace

template <bool kPair, bool kAligned>
static void launch_tiles(const CompressArgs &a, hipStream_t s) {
    const dim3 grid(a.n_tiles), block(kTileWaves * 64);
    switch (a.wave_segs) {
    case 1: hipLaunchKernelGGL((compress_tile_kernel<kPair, kAligned, 1>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((compress_tile_kernel<kPair, kAligned, 2>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((compress_tile_kernel<kPair, kAligned, (u32)kCompressMaxWaveSegs>), grid, block, 0, s, a); break;
    }
}

template <bool kAligned>
static void launch_unseg(const CompressArgs &a, hipStream_t s) { // compress_unseg_pair_kernel: tile shapes as compress_pair_kernel's
    const dim3 grid(a.n_tiles), block(kTileWaves * 64);
    const u32 body = a.wave_segs / 2, tail = a.tail_pairs;
    if (body == 3 && tail == 1)
        hipLaunchKernelGGL((compress_unseg_pair_kernel<kAligned, 3, 1>), grid, block, 0, s, a);
    else if (body == 3 && tail == 2)
        hipLaunchKernelGGL((compress_unseg_pair_kernel<kAligned, 3, 2>), grid, block, 0, s, a);
    else if (body == 3)
        hipLaunchKernelGGL((compress_unseg_pair_kernel<kAligned, 3, 3>), grid, block, 0, s, a);
    else if (body == 2)
        hipLaunchKernelGGL((compress_unseg_pair_kernel<kAligned, 2, 2>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((compress_unseg_pair_kernel<kAligned, 1, 1>), grid, block, 0, s, a);
}

uint32_t compress_nowait_wave_segs() { return kNoWaitWaveSegs; }

hipError_t launch_compress_nowait(const CompressArgs &a, hipStream_t s) {
    const dim3 grid(a.n_tiles), block(kTileWaves * 64);
    if (a.in2) { // pair mode
        hipLaunchKernelGGL(compress_tile_pair_nowait_kernel<kTileCount>, grid, block, 0, s, a);
        hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(compress_tile_pair_nowait_kernel<kTilePlace>, grid, block, 0, s, a);
        return hipGetLastError();
    }
    if (a.unseg_desc) { // unsegmented mode
        if (a.fast_segments) {
            hipLaunchKernelGGL((compress_unseg_pair_nowait_kernel<true, kTileCount>), grid, block, 0, s, a);
            hipLaunchKernelGGL(unseg_offsets_kernel, dim3(1), dim3(64), 0, s, a);
            hipLaunchKernelGGL((compress_unseg_pair_nowait_kernel<true, kTilePlace>), grid, block, 0, s, a);
        } else {
            hipLaunchKernelGGL((compress_unseg_pair_nowait_kernel<false, kTileCount>), grid, block, 0, s, a);
            hipLaunchKernelGGL(unseg_offsets_kernel, dim3(1), dim3(64), 0, s, a);
            hipLaunchKernelGGL((compress_unseg_pair_nowait_kernel<false, kTilePlace>), grid, block, 0, s, a);
        }
        return hipGetLastError();
    }
    if (a.fast_segments) {
        hipLaunchKernelGGL((compress_nowait_kernel<true, kTileCount>), grid, block, 0, s, a);
        hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, a);
        hipLaunchKernelGGL((compress_nowait_kernel<true, kTilePlace>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((compress_nowait_kernel<false, kTileCount>), grid, block, 0, s, a);
        hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, a);
        hipLaunchKernelGGL((compress_nowait_kernel<false, kTilePlace>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

template <bool kAligned>
static void launch_pairs(const CompressArgs &a, hipStream_t s) {
    const dim3 grid(a.n_tiles), block(kTileWaves * 64);
    const u32 body = a.wave_segs / 2, tail = a.tail_pairs;
    if (body == 3 && tail == 1)
        hipLaunchKernelGGL((compress_pair_kernel<kAligned, 3, 1>), grid, block, 0, s, a);
    else if (body == 3 && tail == 2)
        hipLaunchKernelGGL((compress_pair_kernel<kAligned, 3, 2>), grid, block, 0, s, a);
    else if (body == 3)
        hipLaunchKernelGGL((compress_pair_kernel<kAligned, 3, 3>), grid, block, 0, s, a);
    else if (body == 2)
        hipLaunchKernelGGL((compress_pair_kernel<kAligned, 2, 2>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((compress_pair_kernel<kAligned, 1, 1>), grid, block, 0, s, a);
}

hipError_t launch_compress(const CompressArgs &a, hipStream_t s) {
    if (a.pair_layout && !a.unseg_desc) {
        if (a.fast_segments)
            launch_pairs<true>(a, s);
        else
            launch_pairs<false>(a, s);
        return hipGetLastError();
    }
    if (a.unseg_desc) {
        if (a.fast_segments)
            launch_unseg<true>(a, s);
        else
            launch_unseg<false>(a, s);
        return hipGetLastError();
    }
    if (a.in2)
        launch_tiles<true, true>(a, s);
    else if (a.fast_segments)
        launch_tiles<false, true>(a, s);
    else // input only 4-byte aligned: dword loads
        launch_tiles<false, false>(a, s);
    return hipGetLastError();
}

hipError_t launch_bitop_tiles(const CompressArgs &a, const BitopOperands &ops, hipStream_t s) {
    const dim3 grid(a.n_tiles), block(kTileWaves * 64);
    if (a.tile_counts) { // the no-wait route: count, scan, place (nobody waits for anybody; the operands are decoded twice)
        hipLaunchKernelGGL(bitop_tile_kernel<kTileCount>, grid, block, 0, s, a, ops);
        hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(bitop_tile_kernel<kTilePlace>, grid, block, 0, s, a, ops);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bitop_tile_kernel<kTileScan>, grid, block, 0, s, a, ops);
    return hipGetLastError();
}

// segments per wavefront for a bitmap of n_segments: enough tiles to occupy the chip first, long tiles after that
uint32_t compress_wave_segs(uint64_t n_segments) {
    static const int forced = [] { // experiments only
        const char *e = experiment_env("WAH_WAVE_SEGS");
        return e ? std::atoi(e) : 0;
    }();
    if (forced == 1 || forced == 2 || forced == 4 || forced == kCompressMaxWaveSegs) return (uint32_t)forced;
    // measured on 4 MiB .. 512 MiB bitmaps (tools/scratch/size_s_sweep.py): 4 MiB 8.0 / 8.6 / 12.9 us with 1 / 2 / 5
    // segments per wave, 16 MiB 16.2 / 13.1 / 15.0, 32 MiB 27.5 / 21.9 / 17.8, 128 MiB 76.6 / 57.2 / 50.6
    if (n_segments <= 2400) return 1;
    if (n_segments <= 6000) return 2;
    return (uint32_t)kCompressMaxWaveSegs;
}

// pair-layout kernel (compress_pair_kernel): the tile shapes of a launch.  A tile is one workgroup's work; the chip runs
// `slots` of them at a time (two workgroups per CU: LDS).  Whole rounds of the slots are filled with BODY tiles of three
// pairs per wave (48 segments); what is left over gets the smallest shape -- one, two or three pairs per wave -- that puts
// it into ONE more round, so that a launch does not end with a few full-size tiles running alone.  Bitmaps of less than a
// round: one shape, the smallest that fits them into one round.
TileShape compress_tile_shape(uint64_t n_segments) {
    static const int forced = [] { // experiments only: 0 switches the kernel off, 1..3: one shape
        const char *e = experiment_env("WAH_WAVE_PAIRS");
        return e ? std::atoi(e) : -1;
    }();
    // (per device: a process may drive several, and not all of them need be the same part)
    static std::atomic<uint32_t> slots_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    uint32_t known = slots_of[dev].load(std::memory_order_relaxed);
    if (known == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        known = (uint32_t)cus * 2u;
        slots_of[dev].store(known, std::memory_order_relaxed);
    }
    const uint64_t slots = known;
    const uint64_t pairs = (n_segments + 1) / 2;
    const uint64_t w = (uint64_t)kTileWaves;
    TileShape t = {3, 3, 0, 0};
    if (forced == 0) return TileShape{0, 0, 0, 0};
    static const char *shape_env = experiment_env("WAH_SHAPE"); // experiments only: "big_tiles,tail_pairs" (body tiles of 3 pairs)
    if (shape_env) {
        unsigned long big = 0, tail = 0;
        if (std::sscanf(shape_env, "%lu,%lu", &big, &tail) == 2 && tail >= 1 && tail <= 2 && big * w * 3 <= pairs) {
            t.tail_pairs = (uint32_t)tail;
            t.big_tiles = (uint32_t)big;
            t.n_tiles = t.big_tiles + (uint32_t)((pairs - big * w * 3 + w * tail - 1) / (w * tail));
            return t;
        }
    }
    if (forced >= 1 && forced <= 3) {
        t.body_pairs = t.tail_pairs = (uint32_t)forced;
        t.big_tiles = t.n_tiles = (uint32_t)((pairs + w * forced - 1) / (w * forced));
        return t;
    }
    auto fits = [&](uint64_t rest) -> uint32_t { // smallest shape that puts `rest` pairs into one round (3: whatever it takes)
        return rest <= slots * w ? 1u : rest <= slots * w * 2 ? 2u : 3u;
    };
    const uint64_t round = slots * w * 3; // pairs of a full round of body tiles
    const uint64_t rounds = pairs / round;
    const uint64_t rest = pairs - rounds * round;
    if (rounds == 0 || rest == 0) { // one shape.  Less than a round: measured (sparse, tools/pair_sizes.py; 1 / 2 / 3 pairs
        // per wave): 8 MiB 8.4 / 10.0 / 12.1 us, 16 MiB 12.3 / 11.5 / 13.3, 32 MiB 22.4 / 18.4 / 16.1, 64 MiB 36.1 / 32.4 / 26.6 --
        // few long tiles beat many short ones as soon as the bitmap is worth more than the launch's latency chain
        const uint32_t p = rounds == 0 ? (pairs <= 1400 ? 1u : pairs <= 3000 ? 2u : 3u) : 3u;
        t.body_pairs = t.tail_pairs = p;
        t.big_tiles = t.n_tiles = (uint32_t)((pairs + w * p - 1) / (w * p));
        return t;
    }
    t.tail_pairs = fits(rest);
    t.big_tiles = (uint32_t)(rounds * slots);
    t.n_tiles = t.big_tiles + (uint32_t)((rest + w * t.tail_pairs - 1) / (w * t.tail_pairs));
    if (t.tail_pairs == 3) t.big_tiles = t.n_tiles;
    return t;
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

} // namespace wah
