/*
 * wah_oracle.c -- serial CPU restatement of the reference WAH path.
 * TEST INFRASTRUCTURE ONLY (see wah_oracle.h for the rules and parity status).
 *
 * Every function cites the reference lines it restates.  Nothing here is
 * derived from the product's HIP code; the two only meet in tests/.
 */
#define _POSIX_C_SOURCE 200809L /* pthread_barrier_t under -std=c11 */
#include "wah_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* compress.cu:74-81: maxExpectedSize = ceil(8*sizeof(int)*dataSize / 31). */
uint64_t wah_oracle_max_words(uint64_t n_words) {
    return (32u * n_words + 30u) / 31u;
}

/* Stream bit k = in[k/32] >> (k%32) & 1 (kernels.cu:79 places input bit i of
 * word j at stream position 32j+i); bits past the input are zero (F5). */
static inline uint64_t stream_word(const uint32_t *in, uint64_t n_words, uint64_t w) {
    return w < n_words ? in[w] : 0u;
}

/* kernels.cu:72-79: lane `id` of a warp builds
 *   ONES31 & ((word[id-1] >> (32-id)) | (word[id] << id)),
 * i.e. the 31 stream bits starting at bit 31*g.  Done here with one 64-bit
 * window, which also avoids the shift-by-32 of lane 0. */
uint32_t wah_oracle_group(const uint32_t *in, uint64_t n_words, uint64_t g) {
    const uint64_t bit = 31u * g;
    const uint64_t q = bit >> 5;
    const unsigned r = (unsigned)(bit & 31u);
    const uint64_t win = stream_word(in, n_words, q) | (stream_word(in, n_words, q + 1) << 32);
    return (uint32_t)(win >> r) & WAH_O_ONES31;
}

/* One open fill run inside the current segment. */
typedef struct {
    uint32_t kind; /* 0 = none, WAH_O_FILL = zero fill, WAH_O_FILL|WAH_O_FILL_ONE = one fill */
    uint32_t len;
} run_t;

static inline uint64_t flush_run(run_t *r, uint32_t *out, uint64_t c) {
    if (r->kind) {
        out[c++] = r->kind | r->len; /* kernels.cu:244-249 */
        r->kind = 0;
        r->len = 0;
    }
    return c;
}

/* Classify + run detect + emit for one group (kernels.cu:93-112 classify,
 * :126-141 run ends, :188-229 merge == maximal run inside the block). */
static inline uint64_t push_group(uint32_t x, run_t *r, uint32_t *out, uint64_t c) {
    if (x == 0u || x == WAH_O_ONES31) {
        const uint32_t kind = x ? (WAH_O_FILL | WAH_O_FILL_ONE) : WAH_O_FILL;
        if (r->kind == kind) {
            r->len++;
        } else {
            c = flush_run(r, out, c);
            r->kind = kind;
            r->len = 1;
        }
    } else {
        c = flush_run(r, out, c);
        out[c++] = x; /* literal: bit31 clear, kernels.cu:256 */
    }
    return c;
}

/* Compress segments [seg_lo, seg_hi) of the bitmap into out; returns words. */
static uint64_t compress_segments(const uint32_t *in, uint64_t n_words, uint64_t seg_lo, uint64_t seg_hi,
                                  uint32_t *out) {
    const uint64_t G = wah_oracle_max_words(n_words);
    uint64_t c = 0;
    for (uint64_t s = seg_lo; s < seg_hi; ++s) {
        run_t run = {0, 0};
        const uint64_t w0 = s * WAH_O_SEG_WORDS;
        if (w0 + WAH_O_SEG_WORDS <= n_words) {
            /* whole block: 32 rows of 31 words -> 32 groups each (one CUDA warp, kernels.cu:68) */
            for (unsigned row = 0; row < 32; ++row) {
                const uint32_t *w = in + w0 + 31u * row;
                c = push_group(w[0] & WAH_O_ONES31, &run, out, c);
                for (unsigned k = 1; k < 31; ++k)
                    c = push_group(((w[k] << k) | (w[k - 1] >> (32 - k))) & WAH_O_ONES31, &run, out, c);
                c = push_group(w[30] >> 1, &run, out, c);
            }
        } else {
            /* tail block (F5): zero padded, may be short */
            const uint64_t g0 = s * WAH_O_SEG_GROUPS;
            const uint64_t g1 = (g0 + WAH_O_SEG_GROUPS < G) ? g0 + WAH_O_SEG_GROUPS : G;
            for (uint64_t g = g0; g < g1; ++g) c = push_group(wah_oracle_group(in, n_words, g), &run, out, c);
        }
        c = flush_run(&run, out, c); /* F4: never carry a run across a block (tests.cpp:166-172) */
    }
    return c;
}

static uint64_t segment_count(uint64_t n_words) {
    const uint64_t G = wah_oracle_max_words(n_words);
    return (G + WAH_O_SEG_GROUPS - 1) / WAH_O_SEG_GROUPS; /* == ceil(n/992), compress.cu:62-67 */
}

uint64_t wah_oracle_compress(const uint32_t *in, uint64_t n_words, uint32_t *out) {
    return compress_segments(in, n_words, 0, segment_count(n_words), out);
}

/* ---- multi-threaded variant (CPU baseline only) ----
 * Segments are independent (F4): every thread COUNTS the words of its contiguous range of segments (compressing each
 * segment into a 4 KiB stack buffer), thread 0 turns the per-thread counts into offsets (= the block-offset scan,
 * compress.cu:146), and every thread compresses its range again, straight to its place.  (Per-thread buffers of up to
 * a bitmap's worth of words + a serial join made the all-cores figure slower than one core on incompressible data:
 * page faults on the second copy of the output, not arithmetic, set the pace.) */
typedef struct {
    const uint32_t *in;
    uint64_t n_words, seg_lo, seg_hi, count, offset;
    uint32_t *out;
} mt_job;

/* A persistent pool: the workers are created once and parked on a barrier between calls.  (Creating 256 threads per
 * call cost as much as the compression itself and made the all-cores figure swing 11 .. 38 GB/s between two boxes of the
 * same CPU.)  Three barriers per call: start (jobs are set), counted (every job's word count is known -- thread 0 turns
 * them into offsets), done.  The caller takes part in start and done. */
static struct {
    pthread_t *tid;
    mt_job *jobs;
    int threads;          /* workers in the pool (0: none) */
    int wanted;           /* ... and how many were asked for */
    pthread_barrier_t start, counted, done;
    volatile int quit;
} g_pool;

static pthread_mutex_t g_gate_m = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_gate_c = PTHREAD_COND_INITIALIZER;
static int g_gate_open; /* set once the barriers are sized for the workers that actually started */

static void *mt_worker(void *p) {
    const int index = (int)(intptr_t)p;
    uint32_t one[WAH_O_SEG_GROUPS + 1]; /* a segment emits at most 1024 words */
    pthread_mutex_lock(&g_gate_m);
    while (!g_gate_open) pthread_cond_wait(&g_gate_c, &g_gate_m);
    pthread_mutex_unlock(&g_gate_m);
    for (;;) {
        pthread_barrier_wait(&g_pool.start);
        if (g_pool.quit) return NULL;
        mt_job *j = &g_pool.jobs[index];
        j->count = 0;
        for (uint64_t s = j->seg_lo; s < j->seg_hi; ++s) j->count += compress_segments(j->in, j->n_words, s, s + 1, one);
        pthread_barrier_wait(&g_pool.counted);
        if (index == 0) {
            uint64_t c = 0;
            for (int t = 0; t < g_pool.threads; ++t) {
                g_pool.jobs[t].offset = c;
                c += g_pool.jobs[t].count;
            }
        }
        pthread_barrier_wait(&g_pool.counted);
        (void)compress_segments(j->in, j->n_words, j->seg_lo, j->seg_hi, j->out + j->offset);
        pthread_barrier_wait(&g_pool.done);
    }
}

static void pool_stop(void) {
    if (!g_pool.threads) return;
    g_pool.quit = 1;
    pthread_barrier_wait(&g_pool.start);
    for (int t = 0; t < g_pool.threads; ++t) pthread_join(g_pool.tid[t], NULL);
    pthread_barrier_destroy(&g_pool.start);
    pthread_barrier_destroy(&g_pool.counted);
    pthread_barrier_destroy(&g_pool.done);
    free(g_pool.tid);
    free(g_pool.jobs);
    g_pool.tid = NULL;
    g_pool.jobs = NULL;
    g_pool.threads = 0;
    g_pool.quit = 0;
    g_gate_open = 0;
}

/* A pool of up to `threads` workers; returns how many there are (a thread that fails to start is simply not counted:
 * the barriers are sized for the ones that did, 0 = none). */
static int pool_start(int threads) {
    if (g_pool.threads && g_pool.wanted == threads) return g_pool.threads;
    pool_stop();
    g_pool.tid = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    g_pool.jobs = (mt_job *)calloc((size_t)threads, sizeof(mt_job));
    if (!g_pool.tid || !g_pool.jobs) {
        free(g_pool.tid);
        free(g_pool.jobs);
        g_pool.tid = NULL;
        g_pool.jobs = NULL;
        return 0;
    }
    int started = 0;
    for (; started < threads; ++started)
        if (pthread_create(&g_pool.tid[started], NULL, mt_worker, (void *)(intptr_t)started) != 0) break;
    if (started == 0) {
        free(g_pool.tid);
        free(g_pool.jobs);
        g_pool.tid = NULL;
        g_pool.jobs = NULL;
        return 0;
    }
    pthread_barrier_init(&g_pool.start, NULL, (unsigned)started + 1u);
    pthread_barrier_init(&g_pool.counted, NULL, (unsigned)started);
    pthread_barrier_init(&g_pool.done, NULL, (unsigned)started + 1u);
    g_pool.threads = started;
    g_pool.wanted = threads;
    pthread_mutex_lock(&g_gate_m);
    g_gate_open = 1;
    pthread_cond_broadcast(&g_gate_c);
    pthread_mutex_unlock(&g_gate_m);
    return started;
}

void wah_oracle_pool_release(void) { pool_stop(); }

uint64_t wah_oracle_compress_mt(const uint32_t *in, uint64_t n_words, uint32_t *out, int threads) {
    const uint64_t nseg = segment_count(n_words);
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > nseg) threads = nseg ? (int)nseg : 1;
    if (threads > 1) threads = pool_start(threads);
    if (threads <= 1) return wah_oracle_compress(in, n_words, out);
    const uint64_t per = (nseg + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        uint64_t lo = per * t, hi = lo + per;
        if (lo > nseg) lo = nseg;
        if (hi > nseg) hi = nseg;
        g_pool.jobs[t].in = in;
        g_pool.jobs[t].n_words = n_words;
        g_pool.jobs[t].seg_lo = lo;
        g_pool.jobs[t].seg_hi = hi;
        g_pool.jobs[t].out = out;
    }
    pthread_barrier_wait(&g_pool.start);
    pthread_barrier_wait(&g_pool.done);
    uint64_t c = 0;
    for (int t = 0; t < threads; ++t) c += g_pool.jobs[t].count;
    return c;
}

/* getCounts (kernels.cu:291-309): fill -> low 30 bits, literal -> 1; summed
 * as the exclusive scan + last element of decompress.cu:72-82 does. */
uint64_t wah_oracle_decoded_groups(const uint32_t *comp, uint64_t c_words) {
    uint64_t g = 0;
    for (uint64_t i = 0; i < c_words; ++i) g += (comp[i] & WAH_O_FILL) ? (comp[i] & WAH_O_COUNT_MASK) : 1u;
    return g;
}

/* decompress.cu:84-93. */
uint64_t wah_oracle_decoded_words(uint64_t groups) {
    return (31u * groups + 31u) / 32u;
}

/* decompressWords (kernels.cu:321-359) expands every word to 31-bit groups;
 * mergeWords (kernels.cu:369-385) re-packs 32 groups into 31 words, i.e. the
 * groups are appended LSB-first to the output bit stream. */
uint64_t wah_oracle_decompress(const uint32_t *comp, uint64_t c_words, uint32_t *out) {
    uint64_t acc = 0; /* pending stream bits, LSB first */
    unsigned have = 0;
    uint64_t o = 0;
    for (uint64_t i = 0; i < c_words; ++i) {
        const uint32_t w = comp[i];
        uint32_t grp, reps;
        if (w & WAH_O_FILL) {
            grp = (w & WAH_O_FILL_ONE) ? WAH_O_ONES31 : 0u; /* kernels.cu:337-344 */
            reps = w & WAH_O_COUNT_MASK;                    /* kernels.cu:334 */
        } else {
            grp = w; /* kernels.cu:353 */
            reps = 1;
        }
        for (uint32_t k = 0; k < reps; ++k) {
            acc |= (uint64_t)grp << have;
            have += 31;
            if (have >= 32) {
                out[o++] = (uint32_t)acc;
                acc >>= 32;
                have -= 32;
            }
        }
    }
    if (have) out[o++] = (uint32_t)acc; /* zero padded last word */
    return o;
}
