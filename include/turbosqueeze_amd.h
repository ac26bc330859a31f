/*
 * turbosqueeze_amd.h -- C ABI of libturbosqueeze_amd.so, the MI355X (gfx950) implementation
 * of turbosqueeze's per-block encode/decode hot path.
 *
 * Plain C: pointers, sizes and function pointers only.  Two groups of entry points:
 *
 *  (1) the reference's own block-codec and scheduler API, same names, argument meaning and
 *      error behaviour as /root/reference/turbosqueeze.h:441-674, so that a program built
 *      against the reference links against this library instead.  The two reference functions
 *      that take std::function are additionally exported as tsqa_*_cb twins with C function
 *      pointers (for cgo / ctypes / JNI callers); the std::function forms themselves are
 *      declared in include/turbosqueeze.h (C++).
 *
 *  (2) tsqa_* device-resident entry points: the same path with input and output already in
 *      HBM (what bench.py times, and what a GPU-side consumer of .tsq data would bind).
 *
 * Every call runs hand-written HIP kernels; there is no CPU fallback.  If no gfx950 device is
 * usable the calls fail (tsqa_* return TSQA_ERR_NO_DEVICE, the tsq* forms report failure the way
 * the reference reports a failed job) -- they never compute on the host.
 */
#ifndef TURBOSQUEEZE_AMD_H
#define TURBOSQUEEZE_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- format constants (reference turbosqueeze.h:37-43) ---- */
#ifndef TSQ_BLOCK_BITS
#define TSQ_BLOCK_BITS (22)
#define TSQ_BLOCK_SZ   (1 << TSQ_BLOCK_BITS)
#define TSQ_OUTPUT_SZ  ((1 << TSQ_BLOCK_BITS) + (1 << (TSQ_BLOCK_BITS - 2)))
#define TSQ_HASH_BITS  (17)
#define TSQ_HASH_SZ    ((1 << TSQ_HASH_BITS) * sizeof(uint16_t))
#define TSQ_HASH_MASK  ((1 << TSQ_HASH_BITS) - 1)
#endif

/* ---- status codes of the tsqa_* entry points ---- */
enum {
    TSQA_OK            = 0,
    TSQA_ERR_NO_DEVICE = 1,   /* no usable HIP device / kernels not loadable */
    TSQA_ERR_HIP       = 2,   /* a HIP runtime call failed (tsqa_last_error has the text) */
    TSQA_ERR_ARG       = 3,   /* null pointer, zero size, capacity too small */
    TSQA_ERR_FORMAT    = 4,   /* bad magic, n_blocks == 0, frame size 0 or > TSQ_OUTPUT_SZ, truncated */
    TSQA_ERR_STREAM    = 5,   /* a block stream is malformed (bad offset, overrun) */
    TSQA_ERR_OVERFLOW  = 6    /* a block expanded beyond TSQ_OUTPUT_SZ */
};

/* =====================================================================================
 * (2) Device-resident path
 * ================================================================================== */

typedef struct tsqa_ctx tsqa_ctx;   /* one per (process, device): streams + scratch in HBM */

/* device < 0: use the current HIP device.  Scratch is grown on demand and kept. */
int         tsqa_create(int device, tsqa_ctx **out);
void        tsqa_destroy(tsqa_ctx *ctx);
const char *tsqa_last_error(const tsqa_ctx *ctx);
int         tsqa_device_id(const tsqa_ctx *ctx);

/* ceil(n / TSQ_BLOCK_SZ): the job split of tsq_threads.cpp:313. */
size_t tsqa_block_count(size_t n);
/* Capacity that always holds the container of an n-byte input:
 * 16 + n_blocks * (3 + TSQ_OUTPUT_SZ)  (tsq_threads.cpp:339). */
size_t tsqa_container_bound(size_t n);

/*
 * Compress n bytes resident at d_in into a complete .tsq container at d_out
 * (header "TSQ1" | u32 n_blocks | u64 n, then per block u24 (size | ext<<23) + stream;
 * turbosqueeze.cpp:64-83).  Replaces the loop tsqInit + tsqEncode over all blocks
 * (tsq_threads.cpp:176-177) plus the writer's frame assembly (tsq_threads.cpp:218-239).
 * Bytes past d_in[n-1] are never read: the encoder's look-ahead beyond the input end sees
 * zeros (canonical conditions).  Work is enqueued on `hip_stream` (a hipStream_t, NULL =
 * the context's own stream); the call returns after the stream has drained and *out_size
 * holds the container size.
 */
int tsqa_compress_device(tsqa_ctx *ctx, const void *d_in, size_t n,
                         void *d_out, size_t out_cap, size_t *out_size,
                         uint32_t ext, void *hip_stream);

/* Asynchronous form: nothing is waited for; the container size lands in *d_out_size
 * (a device uint64).  Used by bench.py so that HIP events see only kernel time. */
int tsqa_compress_device_async(tsqa_ctx *ctx, const void *d_in, size_t n,
                               void *d_out, size_t out_cap, uint64_t *d_out_size,
                               int32_t *d_status, uint32_t ext, void *hip_stream);

/*
 * Decompress a .tsq container resident at d_in (n bytes) into d_out.  Replaces the frame
 * walk (tsq_threads.cpp:513-524) + tsqDecode per block (tsq_threads.cpp:590) + the ordered
 * writer (tsq_threads.cpp:648).  *out_size = header total.  Fails with TSQA_ERR_FORMAT /
 * TSQA_ERR_STREAM instead of over-running like tsq_decode.cpp does on corrupt input.
 */
int tsqa_decompress_device(tsqa_ctx *ctx, const void *d_in, size_t n,
                           void *d_out, size_t out_cap, size_t *out_size,
                           void *hip_stream);

/* Asynchronous form: the caller states n_blocks (the header's count, which the host cannot
 * read without a sync); the frame-walk kernel checks it against the header.  *d_status
 * (device int32) becomes nonzero (a TSQA_ERR_*) on a bad container or stream; the total
 * uncompressed size lands in *d_out_size. */
int tsqa_decompress_device_async(tsqa_ctx *ctx, const void *d_in, size_t n, uint32_t n_blocks,
                                 void *d_out, size_t out_cap, uint64_t *d_out_size,
                                 int32_t *d_status, void *hip_stream);

/* Kernel timing for bench.py: when enabled, every encode / decode kernel launch is bracketed by
 * HIP events recorded on the stream it is launched on (up to 256 launches are kept).
 * tsqa_profile_read waits for those events and returns, per kernel, the summed elapsed
 * milliseconds and the number of launches since the last read. */
int tsqa_profile_enable(tsqa_ctx *ctx, int on);
int tsqa_profile_read(tsqa_ctx *ctx, double *encode_ms, uint32_t *encode_launches,
                      double *decode_ms, uint32_t *decode_launches);

/* Kernel variant selection for A/B measurements: 0 = default (five-wave staged encoder, ring decoder),
 * 1 = serial kernels (one lane walks the block; correctness baseline), 2 = windowed scalar-walk encoder /
 * chunked decoder without history ring, and encode only: 3 = single-wave orbit, 4 = two-wave
 * parser/builder, 5 = three-wave tile pipeline, 6 = the lean layouts (encoder without the input window, decoder without the history ring: two
 * blocks per CU; variant 0 picks them by itself when there are more blocks than CUs), 7 = never lean. */
void tsqa_set_kernel_variant(tsqa_ctx *ctx, int encode_variant, int decode_variant);

/* =====================================================================================
 * (1) The reference API (turbosqueeze.h:441-674), C-callable subset
 * ================================================================================== */

struct TSQCompressionContext {      /* turbosqueeze.h:57-63; tests memset refhash (test/test.cpp:42) */
    uint16_t *refhash;
};
struct TSQCompressionContext_MT;    /* opaque here; first field is uint32_t num_cores (turbosqueeze.h:343) */
struct TSQDecompressionContext_MT;

/* turbosqueeze.h:625,634,643 -- the 256 KiB CPU-visible table is kept for source
 * compatibility; the device keeps its own tables in HBM. */
struct TSQCompressionContext *tsqAllocateContext(void);
void tsqDeallocateContext(struct TSQCompressionContext *ctx);
void tsqInit(struct TSQCompressionContext *ctx);

/* turbosqueeze.h:657 -- one block, synchronous.  inputSize <= TSQ_BLOCK_SZ; outputBlock must
 * hold TSQ_OUTPUT_SZ bytes.  The look-ahead past inputBlock[inputSize-1] sees zeros. */
void tsqEncode(struct TSQCompressionContext *ctx, uint8_t *inputBlock, uint8_t *outputBlock,
               uint32_t *outputSize, uint32_t inputSize, uint32_t withExtensions);
/* turbosqueeze.h:670 -- *outputSize = 0 on an oversize header or a malformed stream. */
void tsqDecode(uint8_t *inputBlock, uint8_t *outputBlock, uint32_t *outputSize,
               uint32_t inputSize, uint32_t withExtensions);

/* turbosqueeze.h:458,470 -- FILE* to FILE*; level is ignored as in the reference. */
void tsqCompress(FILE *in, FILE *out, bool useextensions, uint32_t level);
void tsqDecompress(FILE *in, FILE *out);

/* turbosqueeze.h:480,489,554,563 */
struct TSQCompressionContext_MT   *tsqAllocateContextCompression_MT(bool verbose);
void                               tsqDeallocateContextCompression_MT(struct TSQCompressionContext_MT *ctx);
struct TSQDecompressionContext_MT *tsqAllocateContextDecompression_MT(bool verbose);
void                               tsqDeallocateContextDecompression_MT(struct TSQDecompressionContext_MT *ctx);

/* turbosqueeze.h:508,580 -- infile: `in` is a path; outfile: `*out` is a path; memory output is
 * malloc()ed by the library and free()d by the caller. */
bool tsqCompress_MT(struct TSQCompressionContext_MT *ctx, uint8_t *in, size_t szin, bool infile,
                    uint8_t **out, size_t *szout, bool outfile, bool useextensions, uint32_t level);
bool tsqDecompress_MT(struct TSQDecompressionContext_MT *ctx, uint8_t *in, size_t szin, bool infile,
                      uint8_t **out, size_t *szout, bool outfile);

/* C twins of tsqCompressAsync_MT / tsqDecompressAsync_MT (turbosqueeze.h:543-544,615-616):
 * callbacks are plain function pointers + a user pointer; either may be NULL.  Return the
 * job id (>= 1) or 0 after calling done(0,false,user). */
typedef void (*tsqa_done_fn)(uint32_t jobid, bool ok, void *user);
typedef void (*tsqa_progress_fn)(uint32_t jobid, double fraction, void *user);
uint32_t tsqa_compress_async_cb(struct TSQCompressionContext_MT *ctx, uint8_t *in, size_t szin, bool infile,
                                uint8_t **out, size_t *szout, bool outfile, bool useextensions, uint32_t level,
                                tsqa_done_fn done, tsqa_progress_fn progress, void *user);
uint32_t tsqa_decompress_async_cb(struct TSQDecompressionContext_MT *ctx, uint8_t *in, size_t szin, bool infile,
                                  uint8_t **out, size_t *szout, bool outfile,
                                  tsqa_done_fn done, tsqa_progress_fn progress, void *user);

#ifdef __cplusplus
}
#endif
#endif /* TURBOSQUEEZE_AMD_H */
