/*
 * wah_gen_host.c -- host-side fill loops over include/wah_gen.h.
 * TEST / BENCH INFRASTRUCTURE (part of the oracle library): lets Python build
 * the same synthetic bitmaps on the CPU that the HIP generator kernels build
 * on the GPU, so device inputs can be checked against the oracle.
 */
#include "../include/wah_gen.h"

void wah_gen_host_uniform(uint32_t *dst, uint64_t n_words, uint64_t seed, uint64_t threshold) {
    for (uint64_t w = 0; w < n_words; ++w) dst[w] = wah_gen_uniform_word(seed, w, threshold);
}

void wah_gen_host_clustered(uint32_t *dst, uint64_t n_words, uint64_t seed, uint64_t threshold) {
    const uint64_t chunks = (n_words + WAH_GEN_CHUNK_WORDS - 1) / WAH_GEN_CHUNK_WORDS;
    for (uint64_t c = 0; c < chunks; ++c) {
        const uint64_t w0 = c * WAH_GEN_CHUNK_WORDS;
        const uint64_t left = n_words - w0;
        wah_gen_clustered_chunk(seed, c, threshold, dst + w0,
                                (uint32_t)(left < WAH_GEN_CHUNK_WORDS ? left : WAH_GEN_CHUNK_WORDS));
    }
}
