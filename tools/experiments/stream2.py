p='gpu-wah_amd/csrc/wah_kernels.hip'
s=open(p).read()
anchor="// ===========================================================================\n// bench support"
kernel=r'''// ===========================================================================
// single-pass decode (long streams): decode_stream_kernel
//
// The sums + expand pair reads the compressed stream twice.  Here a persistent workgroup walks over tiles (handed out
// by an arrival ticket), keeps TWO tiles in LDS and overlaps the only thing that needs other workgroups -- where the
// tile starts in the output -- with the expansion of the tile before:
//     publish the group count of tile t+1  ->  expand tile t  ->  look up the base of tile t+1 (answered by then)
// Bases come from a two-level decoupled look-back over 8-byte granules {bit 63 = valid, value}: T[t] = groups of tile t;
// per group of 64 tiles GA[g] = its groups, GP[g] = groups of everything up to and including it.  A ticket holder
// only ever waits for tiles with smaller tickets, which have started; every wait is bounded.
// Base of tile t = 64 g + m:  [nearest earlier GP + the GAs after it] + T[64 g .. t).  The last tile of a group
// publishes GA as soon as its lower tiles are in (before it expands anything), and GP after its own look-back.
// ===========================================================================
constexpr u64 kLbValid = 1ull << 63;
constexpr u32 kLbGroup = 64;

__device__ __forceinline__ u64 lb_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_store(u64 *p, u64 v) { __hip_atomic_store(p, v | kLbValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// groups of the tiles below `tile` in its group of 64 (whole wavefront); false on a timeout
__device__ __forceinline__ bool lb_lower_tiles(const ExpandArgs &a, u32 tile, u32 lane, u64 &low) {
    const u32 g = tile / kLbGroup, m = tile % kLbGroup;
    low = 0;
    for (u32 spins = 0; m != 0u;) {
        const u64 v = lane < m ? lb_load(a.tile_desc + (u64)g * kLbGroup + lane) : kLbValid;
        if (!__any(!(v & kLbValid))) {
            low = uniform64(wave_sum(lane < m ? v & ~kLbValid : 0ull));
            break;
        }
        if (++spins > kMaxSpins) {
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrTimeout);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}

// called when the tile's words are in LDS: makes its group count known (whole wavefront)
__device__ __forceinline__ bool lb_publish(const ExpandArgs &a, u32 tile, u64 total, u32 lane) {
    if (lane == 0) lb_store(a.tile_desc + tile, total);
    if (tile % kLbGroup == kLbGroup - 1u || tile == a.n_tiles - 1u) { // last of its group: the group's total
        u64 low;
        if (!lb_lower_tiles(a, tile, lane, low)) return false;
        if (lane == 0) lb_store(a.group_desc + tile / kLbGroup, low + total);
    }
    return true;
}

// groups in front of the tile (whole wavefront); false on a timeout
__device__ __forceinline__ bool lb_base(const ExpandArgs &a, u32 tile, u64 total, u32 lane, u64 &base_out) {
    u64 *const GA = a.group_desc;                                            // group aggregates
    u64 *const GP = a.group_desc + ((a.n_tiles + kLbGroup - 1u) / kLbGroup); // group inclusive prefixes
    const u32 g = tile / kLbGroup;
    u64 low;
    if (!lb_lower_tiles(a, tile, lane, low)) return false;
    u64 before = 0; // groups of all earlier groups of tiles
    u32 back = 0;   // groups already walked over behind g (all had aggregates, none a prefix)
    for (u32 spins = 0; back < g;) {
        // lane l looks at group g - 1 - back - l
        const bool in = back + lane < g;
        const u32 gi = in ? g - 1u - back - lane : 0u;
        const u64 vp = in ? lb_load(GP + gi) : 0ull;
        const u64 va = in ? lb_load(GA + gi) : kLbValid; // (in front of the stream: aggregate 0)
        const u64 has_p = __ballot(in && (vp & kLbValid));
        const u64 has_a = __ballot(!in || (va & kLbValid));
        // the nearest group with a prefix counts if every group in front of it has its aggregate
        const u32 first_p = has_p ? (u32)__ffsll((long long)has_p) - 1u : 64u;
        const u64 front = first_p >= 64u ? ~0ull : ((1ull << first_p) - 1ull);
        if ((has_a & front) == front) {
            const u64 contrib = lane < first_p ? (in ? va & ~kLbValid : 0ull) : (lane == first_p ? vp & ~kLbValid : 0ull);
            before += uniform64(wave_sum(contrib));
            if (first_p < 64u) break;
            back += 64u; // a whole window of aggregates without a prefix: keep walking back
            spins = 0;
            continue;
        }
        if (++spins > kMaxSpins) {
            if (lane == 0) atomicOr(a.ctrl + kCtlError, kErrTimeout);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if ((tile % kLbGroup == kLbGroup - 1u || tile == a.n_tiles - 1u) && lane == 0) lb_store(GP + g, before + low + total);
    base_out = before + low;
    return true;
}

__global__ __launch_bounds__(kExpandThreads) void decode_stream_kernel(const ExpandArgs a) {
    __shared__ __attribute__((aligned(16))) u32 s_words[2][kScanTileWords];
    __shared__ u64 s_coarse[2][kCoarse + 1];   // groups in front of word 64 c, relative to the tile start
    __shared__ u32 s_coarse32[2][kCoarse + 1]; // the same in 32 bits (valid when the tile total is below 2^31)
    __shared__ u64 s_wave_sum[kExpandWaves];
    __shared__ u32 s_wave_empty[kExpandWaves];
    __shared__ __attribute__((aligned(16))) unsigned char s_flag[kExpandWaves][kFlagBytes];
    __shared__ u64 s_base;
    __shared__ u64 s_deferred; // segment that has to be redone by the routine for empty fills (~0: none)
    __shared__ u32 s_ticket;
    __shared__ u32 s_ok;

    const u32 lane = lane_id();
    const u32 wave = wave_id();
    constexpr int kVec = kExpandWordsPerThread / 4;
    constexpr u32 kThreadsPer64 = 64 / kExpandWordsPerThread;

    // tiles that can be prefetched into registers: whole, 16-byte aligned
    auto prefetchable = [&](u32 tile) { return a.aligned16 && (u64)(tile + 1u) * kScanTileWords <= a.c_words; };
    auto prefetch = [&](u32 tile, uint4 (&v)[kVec]) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + (u64)tile * kScanTileWords);
#pragma unroll
        for (int k = 0; k < kVec; ++k) v[k] = src[k * kExpandThreads + (int)threadIdx.x]; // coalesced 16-byte loads
    };
    // words into LDS buffer b, coarse prefix, group count published; returns the count, sets `empty`
    auto stage = [&](u32 b, u32 tile, const uint4 (&v)[kVec], bool have, bool &empty) -> u64 {
        const u64 tile_w0 = (u64)tile * kScanTileWords;
        if (have) {
            uint4 *dst = reinterpret_cast<uint4 *>(s_words[b]);
#pragma unroll
            for (int k = 0; k < kVec; ++k) dst[k * kExpandThreads + (int)threadIdx.x] = v[k];
        } else {
            for (u32 i = threadIdx.x; i < (u32)kScanTileWords; i += kExpandThreads)
                s_words[b][i] = tile_w0 + i < a.c_words ? a.comp[tile_w0 + i] : 0x80000000u; // past the end: empty fill
        }
        __syncthreads();
        // every thread sums the counts of its own 16 consecutive words (and looks for empty fills among the real ones)
        u64 mine = 0;
        u32 n_min = 1;
        {
            const uint4 *my = reinterpret_cast<const uint4 *>(s_words[b] + threadIdx.x * kExpandWordsPerThread);
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                const uint4 q = my[k];
                const u32 nx = word_groups(q.x), ny = word_groups(q.y), nz = word_groups(q.z), nw = word_groups(q.w);
                mine += (u64)(nx + ny + nz + nw);
                const u64 w = tile_w0 + threadIdx.x * kExpandWordsPerThread + 4u * k;
                n_min = min(n_min, w + 0 < a.c_words ? nx : 1u);
                n_min = min(n_min, w + 1 < a.c_words ? ny : 1u);
                n_min = min(n_min, w + 2 < a.c_words ? nz : 1u);
                n_min = min(n_min, w + 3 < a.c_words ? nw : 1u);
            }
        }
        const u64 incl = wave_scan_incl(mine, lane);
        if (lane == 63) s_wave_sum[wave] = incl;
        const bool wave_empty = __any(n_min == 0u);
        if (lane == 0) s_wave_empty[wave] = wave_empty;
        __syncthreads();
        u64 excl = incl - mine;
        for (u32 k = 0; k < wave; ++k) excl += s_wave_sum[k];
        if (threadIdx.x % kThreadsPer64 == 0) { // first thread of each 64 words
            s_coarse[b][threadIdx.x / kThreadsPer64] = excl;
            s_coarse32[b][threadIdx.x / kThreadsPer64] = (u32)excl;
        }
        if (threadIdx.x == kExpandThreads - 1) {
            s_coarse[b][kCoarse] = excl + mine;
            s_coarse32[b][kCoarse] = (u32)(excl + mine);
        }
        empty = (s_wave_empty[0] | s_wave_empty[1] | s_wave_empty[2] | s_wave_empty[3]) != 0u;
        __syncthreads();
        const u64 total = uniform64(s_coarse[b][kCoarse]);
        if (wave == 0 && !lb_publish(a, tile, total, lane) && lane == 0) s_ok = 0;
        return total;
    };

    if (threadIdx.x == 0) {
        s_ticket = draw_arrival(a.ctrl);
        s_ok = 1;
    }
    __syncthreads();
    u32 tile = uniform32(s_ticket);
    if (tile >= a.n_tiles) return;
    uint4 regs[kVec];
    u32 b = 0;
    bool empty = false;
    {
        const bool have = prefetchable(tile);
        if (have) prefetch(tile, regs);
        // (the first tile has nothing to overlap with)
    }
    u64 total = stage(b, tile, regs, prefetchable(tile), empty);

    for (;;) {
        // ---- next tile: ticket, loads in flight
        __syncthreads(); // (s_ticket, s_base, s_deferred of the previous round are no longer read)
        if (threadIdx.x == 0) {
            s_ticket = draw_arrival(a.ctrl);
            s_deferred = ~0ull;
        }
        __syncthreads();
        const u32 next = uniform32(s_ticket);
        const bool has_next = next < a.n_tiles;
        const bool next_in_regs = has_next && prefetchable(next);
        if (next_in_regs) prefetch(next, regs);

        // ---- where the current tile starts in the output
        if (wave == 0) {
            u64 base_w = 0;
            const bool got = lb_base(a, tile, total, lane, base_w);
            if (lane == 0) {
                s_base = base_w;
                if (!got) s_ok = 0;
                if (got && tile == a.n_tiles - 1u) { // the length of the stream is known here, and only here
                    const u64 g_all = base_w + total;
                    const u64 w_all = (31u * g_all + 31u) / 32u;
                    a.info_out[0] = w_all;
                    a.info_out[1] = g_all;
                    if (w_all > a.out_capacity) atomicOr(a.ctrl + kCtlError, kErrCapacity);
                }
            }
        }
        __syncthreads();
        if (!uniform32(s_ok)) return; // a bounded wait expired (error raised)
        const u64 base = uniform64(s_base);

        // ---- the next tile's words and count first: its base resolves while this tile is expanded
        u64 next_total = 0;
        bool next_empty = false;
        if (has_next) next_total = stage(b ^ 1u, next, regs, next_in_regs, next_empty);

        // ---- expand the current tile: the segments whose first group lies in [base, base + total)
        {
            const u64 tile_w0 = (u64)tile * kScanTileWords;
            const u64 k_begin = (base + kSegGroups - 1) / kSegGroups;
            const u64 k_end = (base + total + kSegGroups - 1) / kSegGroups;
            const u64 groups = ~0ull; // not known: the last segment finds its own end (dynamic_tail)
            if (empty) {
                // the tile contains fill words of count 0: index-map route, one wavefront, all four flag areas
                if (wave == 0)
                    for (u64 seg = k_begin; seg < k_end; ++seg)
                        expand_segment_with_empties(a, s_words[b], s_coarse[b], reinterpret_cast<u32 *>(&s_flag[0][0]), tile_w0, base,
                                                    groups, a.out_capacity, seg, lane);
            } else {
                unsigned char *flag = s_flag[wave];
                const bool tame = total < (1ull << 31); // wave-uniform: positions inside this tile fit 32 bits
                for (u64 seg = k_begin + wave; seg < k_end; seg += kExpandWaves) {
                    const bool done = tame ? expand_segment_tame(a, s_words[b], s_coarse32[b], flag, tile_w0, (u32)(seg * kSegGroups - base),
                                                                 kSegGroups, a.out_capacity, seg, lane)
                                           : expand_segment_general(a, s_words[b], s_coarse[b], flag, tile_w0, base, groups, a.out_capacity,
                                                                    seg, lane);
                    if (!done && lane == 0) s_deferred = seg; // (only the last segment reaches past the tile)
                }
                __syncthreads();
                const u64 deferred = uniform64(s_deferred);
                if (deferred != ~0ull && wave == 0)
                    expand_segment_with_empties(a, s_words[b], s_coarse[b], reinterpret_cast<u32 *>(&s_flag[0][0]), tile_w0, base, groups,
                                                a.out_capacity, deferred, lane);
            }
        }
        if (!has_next) break;
        tile = next;
        total = next_total;
        empty = next_empty;
        b ^= 1u;
    }
}

'''
assert anchor in s
s=s.replace(anchor,kernel+anchor,1)
# launcher
anchor2="hipError_t launch_gen_uniform(u32 *out, u64 n, u64 seed, u64 thr, hipStream_t s) {"
launcher='''// single pass (ctrl and the granule tables zeroed by the caller).  Persistent: as many workgroups as are resident
// together (a correctness-neutral choice here: tiles go by ticket, nobody waits for a workgroup that has not started).
hipError_t launch_decode_stream(const ExpandArgs &a0, u64 n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    ExpandArgs a = a0;
    a.parts = 1;
    a.dynamic_tail = 1;
    a.n_tiles = (u32)n_tiles;
    static int per_cu = 0, cus = 0;
    if (per_cu == 0) {
        int dev = 0, n = 0, blocks = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess)
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void *>(&decode_stream_kernel), kExpandThreads, 0);
        if (e != hipSuccess) return e;
        cus = n > 0 ? n : 1;
        per_cu = blocks > 0 ? blocks : 1;
    }
    u64 grid = (u64)per_cu * cus;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(decode_stream_kernel, dim3((unsigned)grid), dim3(kExpandThreads), 0, s, a);
    return hipGetLastError();
}

'''
assert anchor2 in s
s=s.replace(anchor2,launcher+anchor2,1)
open(p,'w').write(s)

p='gpu-wah_amd/csrc/wah_api.hip'
s=open(p).read()
old='''    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(d_workspace);
    hipError_t e = hipSuccess;
    if (do_scan) {
        const int resident = c_words ? wah::decode_sums_grid'''
new='''    hipStream_t s = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(d_workspace);
    hipError_t e = hipSuccess;
    if (do_scan && do_expand && l.n_tiles >= stream_min_tiles()) {
        // single pass: persistent expand kernel that finds its own tile bases (look-back); C is read once
        const uint64_t n_groups = ceil_div(l.n_tiles, (uint64_t)64);
        const size_t used = round256(l.desc_off + (l.n_tiles + 2 * n_groups) * sizeof(uint64_t));
        e = hipMemsetAsync(ws, 0, used < l.zero_bytes ? used : l.zero_bytes, s);
        if (e != hipSuccess) {
            set_err("hipMemsetAsync", e);
            return WAH_ERR_HIP;
        }
        wah::ExpandArgs x = {};
        x.comp = d_comp;
        x.c_words = c_words;
        x.out = d_out;
        x.out_capacity = out_capacity_words;
        x.info = d_out_info;
        x.info_out = d_out_info;
        x.tile_base = nullptr;
        x.ctrl = reinterpret_cast<uint32_t *>(ws + l.ctrl_off);
        x.aligned16 = aligned16(d_comp) ? 1 : 0;
        x.tile_desc = reinterpret_cast<uint64_t *>(ws + l.desc_off);
        x.group_desc = x.tile_desc + l.n_tiles;
        e = wah::launch_decode_stream(x, l.n_tiles, s);
        if (e != hipSuccess) {
            set_err("decode kernel launch", e);
            return WAH_ERR_HIP;
        }
        return WAH_OK;
    }
    if (do_scan) {
        const int resident = c_words ? wah::decode_sums_grid'''
assert old in s
s=s.replace(old,new,1)
old2="int read_status(void *d_workspace, void *stream) {"
new2='''// Streams of at least this many 4096-word tiles are decoded in a single pass (decode_stream_kernel); shorter ones by
// the sums + expand pair, which can give several workgroups to one tile.  WAH_STREAM_MIN_TILES overrides the
// threshold (the tests run both routes on small inputs).
uint64_t stream_min_tiles() {
    if (const char *e = std::getenv("WAH_STREAM_MIN_TILES")) return std::strtoull(e, nullptr, 10);
    return 4096;
}

int read_status(void *d_workspace, void *stream) {'''
s=s.replace(old2,new2,1)
open(p,'w').write(s)
