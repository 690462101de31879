/*
 * wah_refsim.c -- lane-by-lane CPU emulation of the shipped CUDA pipeline
 *                 compressData -> exclusive_scan -> moveData.
 * TEST INFRASTRUCTURE ONLY (see wah_oracle.h).
 *
 * Why it exists: the reference cannot be built or run here (CUDA 8, sm_60), so
 * "bit-exact vs the reference CUDA path" can only be argued from the kernel
 * text.  wah_oracle.c states the *canonical* result (maximal runs inside a
 * 1024-group block); this file instead plays the kernel the way a (32,32)
 * thread block executes it -- 32 warps of 32 lanes, the five __shared__
 * arrays, the warp-0 merge phase -- so that tests can check
 *        refsim(x) == oracle(x)
 * on structured random blocks and on the reference's own vectors.  It is an
 * emulation of behaviour, written from the kernel's description; it shares no
 * code with the product.
 *
 * Execution-model assumptions (sm_60, pre-Volta lock-step warps):
 *  - lanes of a warp run in lock step; a divergent loop reconverges after the
 *    loop, so every read of endLengths[] inside the warp-0 while-loop
 *    (kernels.cu:200-218) sees the values written before the merge phase, and
 *    the stores at kernels.cu:219/222 happen afterwards;
 *  - PTX shr by 32 yields 0, __shfl_up from lane 0 returns the
 *    caller's own value (irrelevant after the >>32).
 * Only whole blocks are defined (kernels.cu:70).
 */
#include "wah_oracle.h"

#include <stdlib.h>
#include <string.h>

enum { T_ZEROS = 0, T_ONES = 1, T_LITERAL = 2 }; /* const.h:14-16 */
enum { NW = 32, NL = 32 };                        /* blockDim (32,32), compress.cu:84 */

static unsigned popc32(uint32_t v) { return (unsigned)__builtin_popcount(v); }

typedef struct {
    /* per-lane registers after the warp-local phase */
    uint32_t word[NW][NL];
    int type[NW][NL];
    int idle[NW][NL];
    int index[NW][NL];
    int blockSize[NW][NL];
    /* __shared__ arrays, kernels.cu:53-61 */
    int counts[NW], endLengths[NW], endings[NW], beginnings[NW], merged[NW];
} block_state;

/* kernels.cu:66-182, one warp. */
static void warp_local_phase(block_state *b, int w, const uint32_t *blk) {
    uint32_t raw[NL];
    for (int id = 0; id < NL; ++id) raw[id] = (id < NL - 1) ? blk[w * 31 + id] : 0u; /* :72-74 */

    uint32_t zeros = 0, ones = 0;
    for (int id = 0; id < NL; ++id) {
        const uint32_t up = (id == 0) ? raw[0] : raw[id - 1];         /* __shfl_up(word,1) */
        const uint32_t hi = (id == 0) ? 0u : (up >> (32 - id));       /* PTX shr clamps */
        const uint32_t g = WAH_O_ONES31 & (hi | (raw[id] << id));     /* :79 */
        b->word[w][id] = g;
        if (g == 0u) {
            zeros |= 1u << id;
            b->type[w][id] = T_ZEROS;
        } else if (g == WAH_O_ONES31) {
            ones |= 1u << id;
            b->type[w][id] = T_ONES;
        } else {
            b->type[w][id] = T_LITERAL;
        }
    }
    b->endings[w] = b->type[w][NL - 1]; /* markEndWordTypes, :30-34 */
    const uint32_t literals = ~(zeros | ones); /* :117 (after the two OR all-reduces) */

    uint32_t flags = 0x80000000u; /* :127 */
    for (int id = 0; id < NL; ++id) {
        int idle = 1;
        if (id < 31) {
            /* :126-138.  n = 0x3 << id is an int shift; at id == 30 it overflows into the
             * sign bit, the masks still compare as the kernel's 32-bit registers do. */
            const uint32_t n = 0x3u << id, res = 1u << id;
            if ((n & zeros) == res || (n & ones) == res || (literals & res)) {
                flags |= res;
                idle = 0;
            }
        } else {
            idle = 0; /* :139-141 */
        }
        b->idle[w][id] = idle;
    }
    for (int id = 0; id < NL; ++id) {
        b->index[w][id] = (int)popc32(((1u << id) - 1u) & flags); /* :149 */
        if (b->index[w][id] == 0) b->beginnings[w] = b->type[w][id]; /* :151-153 */
        int bs = 1;
        if (!b->idle[w][id]) {
            for (int i = id - 1; i >= 0; --i) { /* :157-162 */
                if (flags & (1u << i)) break;
                bs++;
            }
            if (id == NL - 1) /* writeEndingSize, :36-40,163-173 */
                b->endLengths[w] = (b->type[w][id] == T_LITERAL) ? 0 : bs;
        }
        b->blockSize[w][id] = bs;
    }
    b->counts[w] = (int)popc32(flags); /* :177-179 */
}

/* kernels.cu:188-229, executed by warp 0; lane `id` owns warp slot `id`. */
static void merge_phase(block_state *b, int keep_clause_195) {
    int mergeShift[NL], count[NL], newEnd[NL];
    int endLengthsBefore[NW];
    memcpy(endLengthsBefore, b->endLengths, sizeof endLengthsBefore);
    for (int id = 0; id < NL; ++id) b->merged[id] = 0; /* :190 */
    for (int id = 0; id < NL; ++id) {
        mergeShift[id] = 0;
        count[id] = b->counts[id];
        int not_absorbed = (id == NL - 1) || (b->endings[id] != b->beginnings[id + 1]) ||
                           (b->endings[id] == T_LITERAL);
        if (keep_clause_195) not_absorbed = not_absorbed || (b->counts[id] > 1); /* :195 */
        if (not_absorbed) {
            int i = 1, bonus = 0;
            for (;;) { /* :200-218 */
                const int same = (i <= id) && b->beginnings[id] == b->endings[id - i] &&
                                 b->beginnings[id] != T_LITERAL;
                if (i < id && b->counts[id - i] == 1 && same) { /* whole single-run warp absorbed */
                    mergeShift[id]++;
                    b->merged[id - i] = 1;
                    bonus += endLengthsBefore[id - i];
                    i++;
                } else if (same) { /* tail run of an earlier warp absorbed, then stop */
                    mergeShift[id]++;
                    b->merged[id - i] = 1;
                    bonus += endLengthsBefore[id - i];
                    i++;
                    break;
                } else {
                    break;
                }
            }
            newEnd[id] = bonus; /* :219 */
        } else {
            newEnd[id] = 0; /* :221-223 */
        }
    }
    int shiftScan = 0, offScan = 0;
    for (int id = 0; id < NL; ++id) { /* two localScan()s, :225-228 */
        shiftScan += mergeShift[id];
        offScan += count[id];
        b->endLengths[id] = newEnd[id];
        b->counts[id] = offScan - count[id] - shiftScan;
    }
}

/* kernels.cu:233-259 for one block; returns the block's word count. */
static uint64_t emit_phase(const block_state *b, uint32_t *dst) {
    uint64_t blockCount = 0;
    for (int w = 0; w < NW; ++w) {
        for (int id = 0; id < NL; ++id) {
            int idle = b->idle[w][id];
            if (id == NL - 1) idle = b->merged[w]; /* :233-235 */
            if (idle) continue;
            const int bonus = (b->index[w][id] == 0) ? b->endLengths[w] : 0; /* :242 */
            const int slot = b->index[w][id] + b->counts[w];                 /* :243 */
            uint32_t word = b->word[w][id];
            if (word == WAH_O_ONES31)
                word = 0xC0000000u | (uint32_t)(b->blockSize[w][id] + bonus); /* :244-246 */
            else if (word == 0u)
                word = 0x80000000u | (uint32_t)(b->blockSize[w][id] + bonus); /* :247-249 */
            if (id == NL - 1 && w == NW - 1) blockCount = (uint64_t)slot + 1;  /* :252-253 (lane 31's value) */
            dst[slot] = word;                                                  /* :256, slot inside the block's 1024-word gap */
        }
    }
    return blockCount;
}

static uint64_t refsim(const uint32_t *in, uint64_t n_words, uint32_t *out, int keep_clause_195) {
    if (n_words % WAH_O_SEG_WORDS) return ~(uint64_t)0; /* undefined in the reference */
    const uint64_t blocks = n_words / WAH_O_SEG_WORDS;
    block_state *b = (block_state *)malloc(sizeof *b);
    uint32_t gapped[NW * NL];
    uint64_t c = 0;
    for (uint64_t k = 0; k < blocks; ++k) {
        memset(gapped, 0, sizeof gapped);
        for (int w = 0; w < NW; ++w) warp_local_phase(b, w, in + k * WAH_O_SEG_WORDS);
        merge_phase(b, keep_clause_195);
        const uint64_t cnt = emit_phase(b, gapped);
        /* exclusive_scan of block counts (compress.cu:146) + moveData's copy of the
         * non-zero words to offset[k] + slot (kernels.cu:273-280) */
        for (uint64_t i = 0; i < cnt; ++i) out[c + i] = gapped[i];
        c += cnt;
    }
    free(b);
    return c;
}

uint64_t wah_refsim_compress(const uint32_t *in, uint64_t n_words, uint32_t *out) {
    return refsim(in, n_words, out, 1);
}

uint64_t wah_refsim_compress_pre195(const uint32_t *in, uint64_t n_words, uint32_t *out) {
    return refsim(in, n_words, out, 0);
}
