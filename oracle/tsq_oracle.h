/*
 * tsq_oracle.h -- CPU restatement of turbosqueeze's per-block codec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker.  The shipped library
 * (turbosqueeze_amd/csrc -> libturbosqueeze_amd.so) never links or calls it.
 *
 * Pinning: checked against the known-answer vectors K0..K7 of SURVEY.md
 * section 8c (tests/test_oracle_kat.py) and, in the build container, against
 * the reference's own tsq_encode.cpp / tsq_decode.cpp compiled unmodified
 * into oracle/_ref (tests/test_oracle_vs_ref.py, oracle/Makefile).
 */
#ifndef TSQ_ORACLE_H
#define TSQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSQO_BLOCK_BITS   22
#define TSQO_BLOCK_SZ     (1u << TSQO_BLOCK_BITS)            /* turbosqueeze.h:37-38 */
#define TSQO_OUTPUT_SZ    (TSQO_BLOCK_SZ + (TSQO_BLOCK_SZ >> 2)) /* turbosqueeze.h:39 */
#define TSQO_HASH_BITS    17                                 /* turbosqueeze.h:41 */
#define TSQO_HASH_ENTRIES (1u << TSQO_HASH_BITS)
#define TSQO_HALO         128  /* readable bytes required after every block */

/* Upper bound of the compressed size of an n-byte block (all 16-byte literals):
 * 3 (size) + n + one control per 8 symbols + one size byte per 2 symbols + 2. */
uint32_t tsqo_bound(uint32_t n);

/*
 * Encode one block under the canonical oracle conditions of SURVEY.md 8c:
 *   - `in` must have TSQO_HALO readable bytes after in[n-1] (the next block's
 *     first bytes, or zeros after the final block);
 *   - the output is produced as if `out` had been zero-filled beforehand
 *     (the function zero-fills tsqo_bound(n)+16 bytes itself);
 *   - `table` is the 2^17 x u16 position table; it is zeroed here
 *     (tsqInit, tsq_context.cpp:77-80) and left in its final state.
 * Returns the compressed size.  Follows tsq_encode.cpp:48-189 (ext == 0) and
 * tsq_encode.cpp:192-342 (ext != 0).
 */
uint32_t tsqo_encode_block(const uint8_t *in, uint32_t n, uint8_t *out,
                           uint32_t ext, uint16_t *table);

/*
 * Decode one block stream.  `in_len` bounds the reads (the reference ignores
 * it and over-reads; we stop instead).  Writes exactly the header's size
 * bytes to `out` (never beyond) and returns that size; returns 0 with
 * *status != 0 on an oversize header (tsq_decode.cpp:53,146) or a malformed
 * stream.  Follows tsq_decode.cpp:42-126 / 129-315.
 *   status: 0 ok, 1 size header > 4 MiB, 2 stream truncated,
 *           3 match source before block start, 4 match source overlaps pair.
 */
uint32_t tsqo_decode_block(const uint8_t *in, uint32_t in_len, uint8_t *out,
                           uint32_t ext, int *status);

/*
 * Container helpers (turbosqueeze.cpp:64-83, tsq_threads.cpp:218-239,333-335):
 * 16-byte header "TSQ1" | u32 n_blocks | u64 total, then per block a u24
 * frame (compressed size | ext << 23) followed by the block stream.
 * `in` needs TSQO_HALO zero bytes after in[n-1].  Returns bytes written.
 * `threads` > 1 splits the blocks over that many pthreads (block b -> b % T).
 */
size_t tsqo_compress_bound(size_t n);
size_t tsqo_compress(const uint8_t *in, size_t n, uint8_t *out, uint32_t ext, int threads);
/* Returns the decompressed size, or (size_t)-1 on a malformed container. */
size_t tsqo_decompressed_size(const uint8_t *in, size_t n);
size_t tsqo_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, int threads);

/* CPU baseline for bench.py: block-parallel encode+decode on `threads` pthreads, best of `reps`
 * warm passes.  enc/dec: addresses of the reference's tsqEncode/tsqDecode (oracle/_ref) or NULL
 * for the port.  Returns 0 when the round trip reproduced the input. */
int tsqo_cpubench2(void *enc, void *dec, const uint8_t *in, size_t n, uint32_t ext, int threads, int reps, int shape, int pin,
                   double *enc_seconds, double *dec_seconds, uint64_t *compressed_bytes);
int tsqo_cpubench(void *enc, void *dec, const uint8_t *in, size_t n, uint32_t ext, int threads, int reps,
                  double *enc_seconds, double *dec_seconds, uint64_t *compressed_bytes);

/* FNV-1a 64 (basis cbf29ce484222325, prime 100000001b3). */
uint64_t tsqo_fnv1a64(const uint8_t *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
