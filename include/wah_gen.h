/*
 * wah_gen.h -- synthetic bitmap generators for the benchmark and the tests.
 *
 * Header-only, plain C, usable from host code and from HIP device code, so
 * the GPU bench input and the CPU-side check input are the same bits.
 *
 * These replace the reference's generateRandomData() (tests.cpp:42-64), which
 * draws one glibc rand() per bit (platform specific, serial) and whose
 * `size*32` wraps in 32-bit arithmetic at 2^27 words.  Ours is
 * counter based: every word depends only on (seed, word index), 64-bit safe.
 */
#ifndef WAH_GEN_H_
#define WAH_GEN_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define WAH_GEN_FN static __host__ __device__ __forceinline__
#else
#define WAH_GEN_FN static inline
#endif

/* splitmix64 finaliser over a (seed, counter) pair. */
WAH_GEN_FN uint64_t wah_gen_hash(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed + (ctr + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Word `w` of a Bernoulli(p) bitmap: bit b is set iff a 32-bit draw is below
 * `threshold` = floor(p * 2^32) (pass 1ull<<32 for p = 1). */
WAH_GEN_FN uint32_t wah_gen_uniform_word(uint64_t seed, uint64_t w, uint64_t threshold) {
    uint32_t out = 0;
    for (unsigned j = 0; j < 16; ++j) {
        const uint64_t h = wah_gen_hash(seed, w * 16u + j);
        out |= (uint32_t)((h & 0xFFFFFFFFull) < threshold) << (2 * j);
        out |= (uint32_t)((h >> 32) < threshold) << (2 * j + 1);
    }
    return out;
}

/* Clustered bitmap: alternating 0-runs / 1-runs with geometric lengths.  Every
 * bit flips the running value with probability threshold / 2^32 (mean run
 * 4096 bits <=> threshold = 2^20).  The state restarts at every chunk of
 * WAH_GEN_CHUNK_WORDS words from a hashed start value, so chunks are
 * independent (one GPU thread per chunk).  Writes `count` words of chunk
 * `chunk` (count <= WAH_GEN_CHUNK_WORDS) to dst. */
#define WAH_GEN_CHUNK_WORDS 4096u

WAH_GEN_FN void wah_gen_clustered_chunk(uint64_t seed, uint64_t chunk, uint64_t threshold, uint32_t *dst,
                                        uint32_t count) {
    uint32_t cur = (uint32_t)(wah_gen_hash(seed ^ 0xC1A57E12EDull, chunk) & 1u) ? 0xFFFFFFFFu : 0u;
    for (uint32_t i = 0; i < count; ++i) {
        const uint32_t flips = wah_gen_uniform_word(seed, chunk * WAH_GEN_CHUNK_WORDS + i, threshold);
        /* prefix-xor of the flip bits: bit b = parity of flips[0..b] */
        uint32_t px = flips;
        px ^= px << 1;
        px ^= px << 2;
        px ^= px << 4;
        px ^= px << 8;
        px ^= px << 16;
        dst[i] = px ^ cur;
        cur = (dst[i] >> 31) ? 0xFFFFFFFFu : 0u;
    }
}

#endif /* WAH_GEN_H_ */
