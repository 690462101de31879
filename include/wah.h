/*
 * wah.h -- C ABI of the MI355X-native WAH bitmap compressor / decompressor.
 *
 * This is the drop-in boundary for the reference's hot path.  Every entry point
 * is `extern "C"`, takes plain pointers and sizes, and names the reference
 * interface it replaces.  The reference's own two symbols have C++ linkage
 * (compress.h:12-18, decompress.h:11-17 carry no extern "C"); the library also
 * exports those, declared in include/compress.h and include/decompress.h, so
 * the reference's source.cpp / tests.cpp link against libwah_hip.so unchanged.
 *
 * Units: all sizes are in 32-bit words unless a name says bytes
 * (compress.cu:35-36, decompress.cu:12-13).  Sizes, word indices and offsets are
 * 64-bit throughout; a bitmap or a stream of 2^40 words (4 TiB) or more is
 * refused with WAH_ERR_ARG (the reference: int indices, dataSize < 2^31,
 * kernels.cu:51).  Tested on the GPU up to one launch of 4 362 072 000 words
 * (130 columns of 128 MiB: tests/test_gpu_parity.py).
 *
 * Wire format (reference const.h:3-16, kernels.cu:79,244-249,298-354):
 * 31-bit groups of the LSB-first bit stream; literal = group (bit31 = 0);
 * fill = 0x80000000 | bit<<30 | count; fills are maximal inside a segment of
 * 1024 groups (= 992 input words) and never cross a segment boundary.
 */
#ifndef WAH_H_
#define WAH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares (and the reference's two C++ symbols of
 * compress.h / decompress.h) is ALL it exports (tests/test_abi.py checks the dynamic symbol table) */
#pragma GCC visibility push(default)

#define WAH_SEGMENT_WORDS 992u   /* compress.cu:62  blockCount = dataSize / (31*32) */
#define WAH_SEGMENT_GROUPS 1024u /* kernels.cu:68   one (32,32) CUDA block          */

/* status codes of the device-pointer API (0 = success) */
#define WAH_OK 0
#define WAH_ERR_ARG (-1)      /* null / misaligned pointer, size out of range        */
#define WAH_ERR_WORKSPACE (-2) /* workspace too small, or never initialised           */
#define WAH_ERR_HIP (-3)      /* a HIP runtime call failed (see wah_last_error())    */
#define WAH_ERR_CAPACITY (-4) /* output buffer too small (reported by wah_*_status)  */
#define WAH_ERR_TIMEOUT (-5)  /* an in-kernel bounded wait expired                   */
#define WAH_ERR_STREAM (-6)   /* malformed compressed stream                         */

/* ------------------------------------------------------------------------- *
 * Host-pointer entry points: the reference's API.
 * ------------------------------------------------------------------------- */

/* Replaces compress() -- compress.h:12-18, compress.cu:41-209.
 * data_host: caller-owned host memory, read only, n_words words.
 * Returns a malloc()ed host buffer of *out_words compressed words (release
 * with free() or wah_free()), or NULL on error (message on stderr;
 * compress.cu:89-114 prints and returns NULL).  out_words and the three
 * timing pointers may be NULL (timeMeasuring.h:27-28).  Timings are
 * milliseconds from device events: (alloc + H2D), (device work), (D2H + free)
 * -- compress.cu:117-120,169-172,199-202. */
uint32_t *wah_compress(const uint32_t *data_host, uint64_t n_words, uint64_t *out_words, float *t_to_device_ms,
                       float *t_device_ms, float *t_from_device_ms);

/* Replaces decompress() -- decompress.h:11-17, decompress.cu:18-141.
 * comp_host: c_words compressed words.  Returns a malloc()ed buffer holding G
 * words (G = number of 31-bit groups, decompress.cu:127) of which the first
 * *out_words = ceil(31*G/32) are the bitmap (decompress.cu:84-93); the rest
 * are zero.  NULL on error. */
uint32_t *wah_decompress(const uint32_t *comp_host, uint64_t c_words, uint64_t *out_words, float *t_to_device_ms,
                         float *t_device_ms, float *t_from_device_ms);

/* free() for buffers returned above (source.cpp:108-109, tests.h:22 use free()). */
void wah_free(void *p);

/* The two entry points above keep their device buffers between calls (grow-only, one set per device; calls
 * from several threads take turns), where the reference allocates and frees inside every call
 * (compress.cu:57-114,177-202; decompress.cu:34-54,124-131).  This returns them to the device now; setting
 * WAH_HOST_CACHE=0 in the environment restores allocate-and-free per call.
 * decompress() does not know the decoded size before it has scanned the stream (nor does the reference,
 * decompress.cu:72-97).  With a kept buffer that is large enough -- the one a compress() or decompress() of this
 * process left behind: the reference's callers run them in turn on one size, source.cpp:70-103 -- it decodes into it in
 * ONE pass, one host round trip; without one (a process's first call, a larger bitmap than any before) it sizes one
 * from a sample of the stream in host memory (the group counts of 65 536 words spread evenly over it, plus a
 * sixteenth) and does the same.  Only when that prediction was too small (a foreign stream whose long fills the sample
 * missed) does it go by the reference's order: scan the stream, read the size back, allocate, expand -- two passes
 * over the stream, two round trips.
 *
 * Environment read by the library (nothing else is; the experiment switches of tools/ exist only in builds made with
 * -DWAH_EXPERIMENTS): WAH_HOST_CACHE=0 (above); WAH_FORCE_FALLBACK=1 (every launch by its no-wait route, below);
 * WAH_FAULT_INJECT=timeout -- fault injection: compress() / decompress() treat their first launch as if a bounded
 * in-kernel wait had expired and take the no-wait route by themselves, as they would on a GPU shared in a way that
 * starves the waits (the reference has no failure handling to compare with, compress.cu:89-114). */
void wah_host_cache_release(void);

/* ------------------------------------------------------------------------- *
 * Sizes.
 * ------------------------------------------------------------------------- */

/* ceil(32*n/31): number of 31-bit groups == worst-case compressed size
 * (compress.cu:74-81 maxExpectedSize). */
uint64_t wah_max_compressed_words(uint64_t n_words);

/* ceil(31*G/32): words a stream of G groups decodes to (decompress.cu:84-93). */
uint64_t wah_decoded_words(uint64_t n_groups);

/* Scratch the device-pointer calls need (bytes; 256-byte aligned pointer). */
size_t wah_compress_workspace_bytes(uint64_t n_words);
size_t wah_decompress_workspace_bytes(uint64_t c_words, uint64_t out_capacity_words);

/* ------------------------------------------------------------------------- *
 * Device-pointer entry points (inputs and outputs resident in HBM, caller's
 * stream, no allocation, no host synchronisation -- graph capturable).  These
 * replace the kernel + scan sections of the reference hosts:
 *   compress.cu:129-166  compressData -> exclusive_scan -> moveData
 *   decompress.cu:66-115 getCounts -> exclusive_scan -> decompressWords -> mergeWords
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 * ------------------------------------------------------------------------- */

/* A workspace (compress, decompress, merge_fills) is initialised ONCE (all zero bytes: this call, or any memset) and
 * then keeps itself up: every launch stamps what it leaves there with a launch epoch, so nothing is cleared between
 * launches, and one workspace may serve inputs of different sizes (up to the one it was sized for) in turn -- but only
 * one launch at a time.  A workspace that is neither zeroed nor left by an earlier launch is reported as
 * WAH_ERR_WORKSPACE by the status calls (checked: the magic word, the launch epoch, and every tile number drawn from
 * the ticket counter, before anything is indexed with it).  Where a launch keeps its entries depends on the workspace's
 * size only, so pass the SAME workspace_bytes with a workspace every time.  (The `scratch` of the wah_bitop_* calls
 * needs no initialisation.)  Asynchronous on `stream`. */
int wah_workspace_init_device(void *d_workspace, size_t workspace_bytes, void *stream);

/* d_in: n_words words, 16-byte aligned.  d_out: room for out_capacity_words
 * (wah_max_compressed_words(n) always suffices).  d_out_words: one device
 * uint64 that receives C.  d_workspace: wah_compress_workspace_bytes(n_words) bytes or more, initialised as above.
 * Result status is left in the workspace; read it with wah_compress_status() after the stream has been synchronised.
 * One kernel launch, nothing else: no clearing pass, no residency requirement (the kernel's workgroups are short-lived
 * and only ever wait for workgroups dispatched before them), so it shares the GPU with other work like any kernel. */
int wah_compress_device(const uint32_t *d_in, uint64_t n_words, uint32_t *d_out, uint64_t out_capacity_words,
                        uint64_t *d_out_words, void *d_workspace, size_t workspace_bytes, void *stream);

/* The same with flags.  WAH_UNSEGMENTED: classic WAH straight out of the encoder -- a fill may cross the 1024-group
 * segment cut of the reference (kernels.cu:68,188-229; never a multiple of 2^29 groups, so that every count fits 30
 * bits): exactly the stream wah_merge_fills_device makes of wah_compress_device's, in the one pass.  It
 * decodes to the same bitmap with wah_decompress*; it is not what the reference's encoder emits and has no segment
 * index. */
#define WAH_UNSEGMENTED 1u
/* WAH_NO_WAIT: the same stream as wah_compress_device by a route in which no workgroup ever waits for another: three
 * launches -- count (every tile's word count to a table), one exclusive scan of the table, place (every tile again,
 * its words to their place) -- and the bitmap is read twice (about 1.6 x the time).  The one-launch kernel resolves its
 * offsets with bounded in-kernel waits on workgroups that started earlier (reference: thrust::exclusive_scan between two
 * kernels, compress.cu:129-166); should such a wait ever expire the launch reports WAH_ERR_TIMEOUT, and this is the
 * route to take instead: compress() does so by itself, a caller of the device API passes the flag (or sets
 * WAH_FORCE_FALLBACK=1 in the environment, which sends every plain compress launch this way).  Not combinable with
 * WAH_UNSEGMENTED.  The route itself reads nothing of the workspace's earlier content; before the one-launch kernel is used
 * again after a WAH_ERR_TIMEOUT the workspace must be initialised again (wah_workspace_init_device). */
#define WAH_NO_WAIT 2u
int wah_compress_device_ex(const uint32_t *d_in, uint64_t n_words, uint32_t *d_out, uint64_t out_capacity_words,
                           uint64_t *d_out_words, unsigned flags, void *d_workspace, size_t workspace_bytes, void *stream);

/* Optional side output: when d_segment_offsets != NULL it receives, for every
 * 992-word segment s, the index of its first compressed word
 * (n_segments + 1 entries, the last one = C).  This is the reference's scanned
 * blockCounts array (compress.cu:146) kept instead of thrown away. */
int wah_compress_device_indexed(const uint32_t *d_in, uint64_t n_words, uint32_t *d_out, uint64_t out_capacity_words,
                                uint64_t *d_out_words, uint64_t *d_segment_offsets, void *d_workspace,
                                size_t workspace_bytes, void *stream);

/* Synchronises `stream`, returns WAH_OK or the error a compress launch recorded. */
int wah_compress_status(void *d_workspace, void *stream);

/* Column shards over several GPUs of a node (the reference is one device, default stream:
 * compress.cu:129,166).  Columns are independent bitmaps, so there is nothing to exchange: every shard is the column
 * matrix one device owns (n_columns columns of n_words_per_column words, back to back in that device's memory, every
 * column a whole number of 992-word segments so that no fill crosses a column), compressed by ONE launch on a stream of
 * its own by a host thread of its own that makes `device` current -- no collective, no peer access, no RCCL.  All the
 * other entry points of this header work on the device that is current in the calling thread; this one names its
 * devices.  Per shard: d_out receives the columns' streams back to back (capacity out_capacity_words;
 * wah_max_compressed_words(n_columns * n_words_per_column) always suffices), d_out_words their total,
 * d_segment_offsets (n_columns * n_words_per_column / 992 + 1 entries) the first word of every segment -- column c of
 * the shard starts at entry c * n_words_per_column / 992 --, d_workspace an initialised workspace of
 * wah_compress_workspace_bytes(n_columns * n_words_per_column) bytes on that device.  Returns when every shard has
 * finished: WAH_OK, or the first error (shard order); status[i], if status != NULL, receives shard i's own code.
 * The launches run on streams of the call's own: inputs and workspaces must be complete when it is made (synchronise
 * whatever other stream wrote them), and the outputs are complete when it returns.
 * Several shards may name the same device (they then share it like any two streams). */
typedef struct {
    int device;                 /* HIP device ordinal of this shard */
    uint64_t n_columns;
    const uint32_t *d_in;       /* n_columns * n_words_per_column words on `device`, 16-byte aligned */
    uint32_t *d_out;
    uint64_t out_capacity_words;
    uint64_t *d_out_words;      /* one uint64 on `device` */
    uint64_t *d_segment_offsets; /* may be NULL */
    void *d_workspace;
    size_t workspace_bytes;
} wah_column_shard;
int wah_compress_columns_multi_device(int n_shards, const wah_column_shard *shards, uint64_t n_words_per_column, int *status);

/* d_comp: c_words compressed words, 16-byte aligned.  d_out: room for
 * out_capacity_words decoded words.  d_out_info: two device uint64:
 * [0] = ceil(31*G/32) decoded words, [1] = G groups.
 * Two decoders, the same words out of both.  ONE PASS over the stream (decode_tile_kernel): it decides tile by tile
 * (8192 words), from the tile's own words, whether the workgroup that holds the tile expands it (up to about 7 groups
 * per word) or puts it on a list that a second launch shares out in work items of about 32 output segments (a highly
 * compressed stream: every tile; a long fill inside incompressible data: that tile) -- correct for every stream,
 * the faster one up to about 7 groups per word (by 13-19 %: the stream is read once) and from about 200 on (3-6 %).
 * TWO LAUNCHES (a scan of the stream + an expansion pass, as _scan_device + _expand_device): about 20 % faster between 8
 * and 30 groups per word, where the one pass finds out tile by tile that everything goes onto its list, and within 1.5 %
 * from there to 200.  Without a pass over the
 * stream the library cannot know which it holds, so the DEFAULT goes by what out_capacity_words allows it to be: at most
 * 7 words of output per word of stream, or more than 40: one pass; between: two launches.  A caller who knows the
 * stream passes WAH_ONE_PASS or WAH_TWO_LAUNCHES (wah_decompress_device_ex); decompress(), which has the stream in
 * host memory, samples it (one pass up to 7 groups per word and from 128 on).  A stream that is only 4-byte aligned and WAH_NO_WAIT: always the two launches.
 * wah_last_decode_route() says which one the calling thread's last call launched.
 * On ANY status other than WAH_OK the content of d_out is undefined: on WAH_ERR_CAPACITY the one-pass decoder has
 * written the part that fits (it learns the size while it writes; nothing is ever written behind out_capacity_words),
 * the two-launch routes nothing; on WAH_ERR_TIMEOUT / WAH_ERR_STREAM segments may have been written at positions
 * derived from an unresolved count, inside the capacity. */
int wah_decompress_device(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out, uint64_t out_capacity_words,
                          uint64_t *d_out_info, void *d_workspace, size_t workspace_bytes, void *stream);

/* The same with flags.  WAH_NO_WAIT: the scan of the stream by a route in which no workgroup waits for another (and the
 * two-launch form of the decoder, whatever the stream) -- every 4096-word
 * tile's group total to a table, then one scan launch over the table (the reference's getCounts ->
 * thrust::exclusive_scan, decompress.cu:66-80, with one entry per tile instead of one per word); the expand pass never
 * waits anyway.  It is what decompress() takes by itself when a bounded wait of the one-launch sums kernel has expired
 * (WAH_ERR_TIMEOUT), and what WAH_FORCE_FALLBACK=1 in the environment selects for every call.  The route reads nothing
 * of the workspace's earlier content; before the scan route is used again after a WAH_ERR_TIMEOUT the workspace must be
 * initialised again (wah_workspace_init_device). */
/* WAH_TWO_LAUNCHES: the scan of the stream (decode_sums_kernel) and the expansion (decode_expand_kernel) as two launches
 * with waits, as _scan_device + _expand_device: the stream is read twice.  WAH_ONE_PASS: decode_tile_kernel + the launch
 * over its list, whatever the capacity suggests.  (Not both.) */
#define WAH_TWO_LAUNCHES 4u
#define WAH_ONE_PASS 8u
int wah_decompress_device_ex(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out, uint64_t out_capacity_words,
                             uint64_t *d_out_info, unsigned flags, void *d_workspace, size_t workspace_bytes, void *stream);

/* Which decoder the calling thread's last wah_decompress* call launched: 0 none (an argument error), WAH_ROUTE_ONE_PASS,
 * WAH_ROUTE_TWO_LAUNCHES, WAH_ROUTE_NO_WAIT. */
#define WAH_ROUTE_ONE_PASS 1
#define WAH_ROUTE_TWO_LAUNCHES 2
#define WAH_ROUTE_NO_WAIT 3
int wah_last_decode_route(void);

/* Decoding through the segment index of wah_compress_device_indexed(): segments [first_segment, first_segment +
 * n_segments) of the bitmap (992 words each, the bitmap's last one shorter) are written to d_out[0 ..).  A stream of
 * compress() never lets a fill cross a segment (compress.cu:129-146: one block per 992 words, fills merged inside
 * it), so with the index every segment is an independent job: no scan of the stream (the getCounts + scan half of
 * decompress.cu:56-83 is what the index already holds), one pass, any sub-range.  n_words: length of the ORIGINAL
 * bitmap; the whole bitmap decodes to wah_decoded_words(wah_max_compressed_words(n_words)) words like
 * wah_decompress_device.  Only for streams of this library's (= the reference's) compress(): a range whose words do
 * not make up exactly its segment, or an empty fill word, is reported as WAH_ERR_STREAM by wah_decompress_status().
 * Workspace: wah_decompress_segments_workspace_bytes(), 256-byte aligned. */
size_t wah_decompress_segments_workspace_bytes(void);
int wah_decompress_segments_device(const uint32_t *d_comp, uint64_t c_words, const uint64_t *d_segment_offsets, uint64_t n_words,
                                   uint64_t first_segment, uint64_t n_segments, uint32_t *d_out, uint64_t out_capacity_words,
                                   void *d_workspace, size_t workspace_bytes, void *stream);

/* The segment index of a stream that came WITHOUT one (written by the reference, read from storage, ...):
 * d_segment_offsets receives ceil(G / 1024) + 1 entries, exactly what wah_compress_device_indexed() would have
 * written beside the stream, so that wah_decompress_segments_device / wah_bitop_indexed_device can be used on it from
 * then on.  One scan of the stream (as wah_decompress_scan_device: d_out_info = [decoded words, groups]) + one pass
 * that places every word.  A stream in which a fill crosses a 1024-group boundary, or that contains an empty fill,
 * has no such index: WAH_ERR_STREAM from wah_decompress_status(); too few entries: WAH_ERR_CAPACITY.
 * Workspace: wah_decompress_workspace_bytes(c_words, 0). */
int wah_build_index_device(const uint32_t *d_comp, uint64_t c_words, uint64_t *d_segment_offsets, uint64_t offsets_capacity,
                           uint64_t *d_out_info, void *d_workspace, size_t workspace_bytes, void *stream);

/* First half of the above only (getCounts + scan): fills d_out_info so a
 * caller that does not know the decoded size can allocate, then call
 * wah_decompress_device (which repeats the scan) or _expand_device. */
int wah_decompress_scan_device(const uint32_t *d_comp, uint64_t c_words, uint64_t *d_out_info, void *d_workspace,
                               size_t workspace_bytes, void *stream);
int wah_decompress_expand_device(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out,
                                 uint64_t out_capacity_words, uint64_t *d_out_info, void *d_workspace,
                                 size_t workspace_bytes, void *stream);

int wah_decompress_status(void *d_workspace, void *stream);

/* Stream checker.  The reference decoder accepts ANY sequence of words
 * (kernels.cu:332-354): fills of count 0, literals that should have been fills, fills that run across the
 * 1024-group segments the reference's encoder never crosses (kernels.cu:68, tests.cpp:166-172), adjacent fills it
 * would have merged.  wah_validate_device says what a stream contains, without decoding it.
 * d_report: 8 x uint64 in device memory:
 *   [0] groups the stream expands to          [1] decoded words = ceil(31 * groups / 32)
 *   [2] fill words of count 0                 [3] literal words equal to 0 or 0x7FFFFFFF
 *   [4] fills crossing a 1024-group boundary  [5] adjacent fills of the same kind inside one segment
 *   [6] 1 if [2], [3], [4] and [5] are all zero, else 0
 *   [7] reserved (0)
 * [6] == 1 exactly when the stream is what compress() of this library / of the reference emits for some bitmap
 * ("segment-canonical"): every check above clean.  Workspace: wah_decompress_workspace_bytes(c_words, 0).
 * Asynchronous on `stream`; errors are reported as for wah_decompress_device. */
#define WAH_REPORT_WORDS 8
int wah_validate_device(const uint32_t *d_comp, uint64_t c_words, uint64_t *d_report, void *d_workspace,
                        size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Unsegmented ("canonical") WAH.  compress() never lets a fill cross a 1024-group segment
 * (the reference's block structure, kernels.cu:68, tests.cpp:166-172): a long run costs one word per segment.
 * wah_merge_fills_device rewrites a stream with adjacent fills of the same kind merged into one word (and fills of
 * count 0 dropped), which turns the output of compress() into the classic unsegmented WAH form; the result decodes
 * to the same bitmap with wah_decompress* (whose decoder takes fills of any length), but it is no longer what the
 * reference's encoder would emit.  Runs are not merged across multiples of 2^29 groups, so every count fits 30 bits.
 *   d_out: capacity c_words is always enough.  Workspace: wah_merge_fills_workspace_bytes(c_words), 256-byte aligned.
 *   d_out_words: device uint64.  Errors are read back with wah_decompress_status(d_workspace, stream).
 * ------------------------------------------------------------------------- */
size_t wah_merge_fills_workspace_bytes(uint64_t c_words);
int wah_merge_fills_device(const uint32_t *d_comp, uint64_t c_words, uint32_t *d_out, uint64_t out_capacity_words,
                           uint64_t *d_out_words, void *d_workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Bitwise operations on two compressed bitmaps (not in the reference, whose README.md:10
 * names them as the reason bitmap indexes use WAH).  Both streams must describe bitmaps of n_words words.  The
 * result is what compress() gives for (A op B): decode A, decode B (decode_sums + decode_expand each) into scratch,
 * then ONE pass of the compress kernel that combines the two bitmaps while it stages them -- three streaming passes
 * at HBM speed, nothing leaves the device.
 *   d_scratch: wah_bitop_scratch_bytes(n_words, a_words, b_words) bytes, 256-byte aligned.
 *   Errors that only the device sees (a stream that does not expand to n_words words, output capacity) are read
 *   back by wah_bitop_status(), which synchronises the stream.
 * ------------------------------------------------------------------------- */
#define WAH_OP_AND 0
#define WAH_OP_OR 1
#define WAH_OP_XOR 2
#define WAH_OP_ANDNOT 3 /* A & ~B */
size_t wah_bitop_scratch_bytes(uint64_t n_words, uint64_t a_words, uint64_t b_words);
int wah_bitop_device(int op, uint64_t n_words, const uint32_t *d_a, uint64_t a_words, const uint32_t *d_b,
                     uint64_t b_words, uint32_t *d_out, uint64_t out_capacity_words, uint64_t *d_out_words,
                     void *d_scratch, size_t scratch_bytes, void *stream);
int wah_bitop_status(void *d_scratch, uint64_t n_words, uint64_t a_words, uint64_t b_words, void *stream);

/* The same for operands that come with their segment index (wah_compress_device_indexed): the two streams are walked
 * segment by segment through their indexes, combined group by group in registers, and ONE decoded bitmap is written,
 * which the compress kernel then reads -- no scan of the operands, no second read of them, half the intermediate
 * traffic of wah_bitop_device.  d_out_offsets (may be NULL) receives the result's own segment index, so results can
 * be combined further.  Operands must be streams of compress() for bitmaps of n_words words (anything else:
 * WAH_ERR_STREAM from wah_bitop_indexed_status()).
 *   d_scratch: wah_bitop_indexed_scratch_bytes(n_words) bytes, 256-byte aligned. */
size_t wah_bitop_indexed_scratch_bytes(uint64_t n_words);
int wah_bitop_indexed_device(int op, uint64_t n_words, const uint32_t *d_a, uint64_t a_words, const uint64_t *d_a_offsets,
                             const uint32_t *d_b, uint64_t b_words, const uint64_t *d_b_offsets, uint32_t *d_out,
                             uint64_t out_capacity_words, uint64_t *d_out_words, uint64_t *d_out_offsets, void *d_scratch,
                             size_t scratch_bytes, void *stream);
int wah_bitop_indexed_status(void *d_scratch, uint64_t n_words, void *stream);

/* Two routes, the same words out of both (the library chooses by the operands' lengths; wah_last_bitop_route() says which
 * the calling thread's last wah_bitop_indexed_device / wah_bitop_many_indexed_device call took):
 * WAH_BITOP_ROUTE_RUNS -- all operands together hold at most 112 words per 1024-group segment on average (sparse or
 * clustered bitmaps, what a bitmap index mostly holds): their run lists are MERGED in the compressed domain, one lane per
 * segment, the way WAH operations are done on a CPU; nothing is decoded, nothing of bitmap size is written, the cost goes
 * with the operands' words (count pass, scan of 1/256 of the segment counts, write pass: the operands are read twice).
 * WAH_BITOP_ROUTE_GROUPS -- anything else: every segment decoded into its 1024 groups, combined in registers and
 * compressed again (the description above); the cost goes with the bitmap's length. */
#define WAH_BITOP_ROUTE_RUNS 1
#define WAH_BITOP_ROUTE_GROUPS 2
int wah_last_bitop_route(void);

/* The same for up to 8 operands in ONE combining pass, left to right: A op B op C ... (WAH_OP_ANDNOT: A and not B
 * and not C ...) -- the conjunction of several predicates of a bitmap index in one go.  d_streams / stream_words /
 * d_offsets: HOST arrays of n_operands device pointers / lengths / index pointers.  Scratch and status as for
 * wah_bitop_indexed_device (wah_bitop_indexed_scratch_bytes, wah_bitop_indexed_status). */
int wah_bitop_many_indexed_device(int op, uint64_t n_words, int n_operands, const uint32_t *const *d_streams,
                                  const uint64_t *stream_words, const uint64_t *const *d_offsets, uint32_t *d_out,
                                  uint64_t out_capacity_words, uint64_t *d_out_words, uint64_t *d_out_offsets, void *d_scratch,
                                  size_t scratch_bytes, void *stream);

/* ------------------------------------------------------------------------- *
 * Benchmark support: synthetic bitmaps generated in HBM (include/wah_gen.h
 * states the bit-exact definition; replaces tests.cpp:42-64), and a plain
 * 16-byte-per-lane copy used as the on-box HBM ceiling.
 * ------------------------------------------------------------------------- */
int wah_gen_uniform_device(uint32_t *d_out, uint64_t n_words, uint64_t seed, uint64_t threshold, void *stream);
int wah_gen_clustered_device(uint32_t *d_out, uint64_t n_words, uint64_t seed, uint64_t threshold, void *stream);
int wah_copy_device(const uint32_t *d_in, uint32_t *d_out, uint64_t n_words, void *stream);

/* Last error text of the calling thread ("" if none). */
const char *wah_last_error(void);

/* Library / build identification, e.g. "wah-mi355x 0.1 gfx950". */
const char *wah_version(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* WAH_H_ */
