/*
 * wah_oracle.h -- CPU ORACLE for the WAH compress()/decompress() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gpu-wah_amd/ (the product) may
 * include, link or call this.  Allowed callers: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  The restatement is checked against every known-answer
 * vector the reference's own tests hold for this path (tests.cpp:83-239, see
 * tests/golden/ and tests/test_oracle_kats.py), and against a lane-by-lane
 * emulation of the shipped CUDA kernels (oracle/wah_refsim.c).  The reference
 * itself is CUDA 8 / sm_60 and cannot be built or run in this pipeline
 * (no nvcc, no NVIDIA device), so no reference binary exists under oracle/_ref/
 * for the kernels; oracle/_ref/ only holds the reference's host-side tests.cpp
 * compiled against our boundary (oracle/Makefile, target ref_tests).
 *
 * Wire format restated (reference const.h:3-16, kernels.cu:79,244-249,298-354):
 *   F1  bitmap = LSB-first bit stream over uint32[]; cut into 31-bit groups.
 *   F2  literal word = group value, bit31 = 0; never 0 nor 0x7FFFFFFF.
 *   F3  fill word = 0x80000000 | fillbit<<30 | count (count in groups, >= 1).
 *   F4  fills are maximal inside a 1024-group segment (= 992 input words =
 *       one CUDA block, kernels.cu:68) and never cross a segment boundary.
 *   F5  (reference UB, defined here) if 32*n is not a multiple of 31 the stream
 *       is zero-padded to G = ceil(32n/31) groups (compress.cu:74-81); the
 *       last segment may be short.
 */
#ifndef WAH_ORACLE_H_
#define WAH_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WAH_O_SEG_GROUPS 1024u /* kernels.cu:68  (32 warps x 32 lanes)      */
#define WAH_O_SEG_WORDS 992u   /* compress.cu:62 (31*32 input words/block)  */
#define WAH_O_ONES31 0x7FFFFFFFu
#define WAH_O_FILL 0x80000000u
#define WAH_O_FILL_ONE 0x40000000u
#define WAH_O_COUNT_MASK 0x3FFFFFFFu /* kernels.cu:300 (BIT30 - 1) */

/* G = ceil(32 n / 31): upper bound of the compressed size, compress.cu:74-81 */
uint64_t wah_oracle_max_words(uint64_t n_words);

/* 31-bit group g of the zero-padded stream (kernels.cu:72-79). */
uint32_t wah_oracle_group(const uint32_t *in, uint64_t n_words, uint64_t g);

/* Serial canonical compressor (kernels.cu:85-259 + compress.cu:133-166).
 * out must hold wah_oracle_max_words(n) words.  Returns C. */
uint64_t wah_oracle_compress(const uint32_t *in, uint64_t n_words, uint32_t *out);

/* Same, split over `threads` pthreads by runs of whole segments (segments are
 * independent by F4).  Bit-identical output.  Used as the multi-core CPU
 * baseline only. */
uint64_t wah_oracle_compress_mt(const uint32_t *in, uint64_t n_words, uint32_t *out, int threads);
void wah_oracle_pool_release(void); /* joins the worker threads wah_oracle_compress_mt keeps between calls */

/* Number of 31-bit groups a compressed stream expands to (getCounts,
 * kernels.cu:291-309 + the scan at decompress.cu:72-82). */
uint64_t wah_oracle_decoded_groups(const uint32_t *comp, uint64_t c_words);

/* ceil(31 G / 32), decompress.cu:84-93. */
uint64_t wah_oracle_decoded_words(uint64_t groups);

/* Serial decoder (kernels.cu:321-385).  Writes ceil(31G/32) words to out
 * (caller sizes it with the two functions above).  Returns that count. */
uint64_t wah_oracle_decompress(const uint32_t *comp, uint64_t c_words, uint32_t *out);

/* Lane-level emulation of the shipped CUDA pipeline for whole blocks
 * (n_words % 992 == 0 -- the only domain where the reference is defined,
 * kernels.cu:70).  See wah_refsim.c.  Returns C. */
uint64_t wah_refsim_compress(const uint32_t *in, uint64_t n_words, uint32_t *out);

/* The reference kernel with the `|| counts[id] > 1` clause of kernels.cu:195
 * removed: reproduces the stale 93/186-word vectors of tests.cpp:66-77. */
uint64_t wah_refsim_compress_pre195(const uint32_t *in, uint64_t n_words, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* WAH_ORACLE_H_ */
