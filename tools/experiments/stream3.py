p='gpu-wah_amd/csrc/wah_kernels.hip'
s=open(p).read()
a=s.index("__global__ __launch_bounds__(kExpandThreads) void decode_stream_kernel(const ExpandArgs a) {")
b=s.index("// ===========================================================================\n// bench support")
kernel=r'''__global__ __launch_bounds__(kExpandThreads) void decode_stream_kernel(const ExpandArgs a) {
    __shared__ __attribute__((aligned(16))) u32 s_words[kScanTileWords];
    __shared__ u64 s_coarse[kCoarse + 1];   // groups in front of word 64 c, relative to the tile start
    __shared__ u32 s_coarse32[kCoarse + 1]; // the same in 32 bits (valid when the tile total is below 2^31)
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

    // The words of a tile are read twice by the same workgroup, some ten microseconds apart: once only to count its
    // groups (so that the count is public before the tile in front is expanded), once to stage it.  The second read
    // is served by the memory-side cache, and nothing of the tile has to be held in registers or LDS in between.
    auto count_tile = [&](u32 tile) -> u64 {
        const u64 tile_w0 = (u64)tile * kScanTileWords;
        u64 mine = 0;
        if (a.aligned16 && tile_w0 + kScanTileWords <= a.c_words) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + tile_w0);
            uint4 v[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] = src[k * kExpandThreads + (int)threadIdx.x];
#pragma unroll
            for (int k = 0; k < kVec; ++k)
                mine += (u64)(word_groups(v[k].x) + word_groups(v[k].y) + word_groups(v[k].z) + word_groups(v[k].w));
        } else {
            for (u32 i = threadIdx.x; i < (u32)kScanTileWords; i += kExpandThreads)
                if (tile_w0 + i < a.c_words) mine += word_groups(a.comp[tile_w0 + i]);
        }
        const u64 w = wave_sum(mine);
        __syncthreads(); // (s_wave_sum free)
        if (lane == 0) s_wave_sum[wave] = w;
        __syncthreads();
        return uniform64(s_wave_sum[0] + s_wave_sum[1] + s_wave_sum[2] + s_wave_sum[3]);
    };
    // words into LDS, coarse prefix; sets `empty`: the tile contains fill words of count 0
    auto stage = [&](u32 tile, bool &empty) {
        const u64 tile_w0 = (u64)tile * kScanTileWords;
        if (a.aligned16 && tile_w0 + kScanTileWords <= a.c_words) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.comp + tile_w0);
            uint4 *dst = reinterpret_cast<uint4 *>(s_words);
            uint4 v[kVec];
#pragma unroll
            for (int k = 0; k < kVec; ++k) v[k] = src[k * kExpandThreads + (int)threadIdx.x]; // coalesced 16-byte loads
#pragma unroll
            for (int k = 0; k < kVec; ++k) dst[k * kExpandThreads + (int)threadIdx.x] = v[k];
        } else {
            for (u32 i = threadIdx.x; i < (u32)kScanTileWords; i += kExpandThreads)
                s_words[i] = tile_w0 + i < a.c_words ? a.comp[tile_w0 + i] : 0x80000000u; // past the end: empty fill
        }
        __syncthreads();
        // every thread sums the counts of its own 16 consecutive words (and looks for empty fills among the real ones)
        u64 mine = 0;
        u32 n_min = 1;
        {
            const uint4 *my = reinterpret_cast<const uint4 *>(s_words + threadIdx.x * kExpandWordsPerThread);
            const u64 w0 = tile_w0 + threadIdx.x * kExpandWordsPerThread;
            const bool all_real = w0 + kExpandWordsPerThread <= a.c_words;
#pragma unroll
            for (int k = 0; k < kVec; ++k) {
                const uint4 q = my[k];
                const u32 nx = word_groups(q.x), ny = word_groups(q.y), nz = word_groups(q.z), nw = word_groups(q.w);
                mine += (u64)(nx + ny + nz + nw);
                if (all_real) {
                    n_min = min(min(n_min, min(nx, ny)), min(nz, nw));
                } else {
                    const u64 w = w0 + 4u * k;
                    n_min = min(n_min, w + 0 < a.c_words ? nx : 1u);
                    n_min = min(n_min, w + 1 < a.c_words ? ny : 1u);
                    n_min = min(n_min, w + 2 < a.c_words ? nz : 1u);
                    n_min = min(n_min, w + 3 < a.c_words ? nw : 1u);
                }
            }
        }
        const u64 incl = wave_scan_incl(mine, lane);
        const bool wave_empty = __any(n_min == 0u);
        if (lane == 63) s_wave_sum[wave] = incl;
        if (lane == 0) s_wave_empty[wave] = wave_empty;
        __syncthreads();
        u64 excl = incl - mine;
        for (u32 k = 0; k < wave; ++k) excl += s_wave_sum[k];
        if (threadIdx.x % kThreadsPer64 == 0) { // first thread of each 64 words
            s_coarse[threadIdx.x / kThreadsPer64] = excl;
            s_coarse32[threadIdx.x / kThreadsPer64] = (u32)excl;
        }
        if (threadIdx.x == kExpandThreads - 1) {
            s_coarse[kCoarse] = excl + mine;
            s_coarse32[kCoarse] = (u32)(excl + mine);
        }
        empty = (s_wave_empty[0] | s_wave_empty[1] | s_wave_empty[2] | s_wave_empty[3]) != 0u;
        __syncthreads();
    };

    if (threadIdx.x == 0) {
        s_ticket = draw_arrival(a.ctrl);
        s_ok = 1;
    }
    __syncthreads();
    u32 tile = uniform32(s_ticket);
    if (tile >= a.n_tiles) return;
    bool empty = false;
    stage(tile, empty);
    u64 total = uniform64(s_coarse[kCoarse]);
    if (wave == 0 && !lb_publish(a, tile, total, lane) && lane == 0) s_ok = 0;

    for (;;) {
        // ---- where the current tile starts in the output (its count has been public for a whole expansion)
        __syncthreads(); // (s_ticket, s_base, s_deferred of the previous round are no longer read)
        if (threadIdx.x == 0) {
            s_ticket = draw_arrival(a.ctrl);
            s_deferred = ~0ull;
        }
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
        const u32 next = uniform32(s_ticket);
        const bool has_next = next < a.n_tiles;

        // ---- the next tile's count goes public now: its base resolves while this tile is expanded
        u64 next_total = 0;
        if (has_next) {
            next_total = count_tile(next);
            if (wave == 0 && !lb_publish(a, next, next_total, lane) && lane == 0) s_ok = 0;
        }

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
                        expand_segment_with_empties(a, s_words, s_coarse, reinterpret_cast<u32 *>(&s_flag[0][0]), tile_w0, base, groups,
                                                    a.out_capacity, seg, lane);
            } else {
                unsigned char *flag = s_flag[wave];
                const bool tame = total < (1ull << 31); // wave-uniform: positions inside this tile fit 32 bits
                for (u64 seg = k_begin + wave; seg < k_end; seg += kExpandWaves) {
                    const bool done = tame ? expand_segment_tame(a, s_words, s_coarse32, flag, tile_w0, (u32)(seg * kSegGroups - base),
                                                                 kSegGroups, a.out_capacity, seg, lane)
                                           : expand_segment_general(a, s_words, s_coarse, flag, tile_w0, base, groups, a.out_capacity, seg,
                                                                    lane);
                    if (!done && lane == 0) s_deferred = seg; // (only the last segment reaches past the tile)
                }
                __syncthreads();
                const u64 deferred = uniform64(s_deferred);
                if (deferred != ~0ull && wave == 0)
                    expand_segment_with_empties(a, s_words, s_coarse, reinterpret_cast<u32 *>(&s_flag[0][0]), tile_w0, base, groups,
                                                a.out_capacity, deferred, lane);
            }
        }
        if (!has_next) break;
        __syncthreads(); // everybody is done with the tile in LDS
        tile = next;
        total = next_total;
        stage(tile, empty);
    }
}

'''
s=s[:a]+kernel+s[b:]
open(p,'w').write(s)
