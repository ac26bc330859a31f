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
    TSQA_ERR_OVERFLOW  = 6,   /* a block expanded beyond TSQ_OUTPUT_SZ */
    TSQA_ERR_STALL     = 7    /* a decode on several workgroups per block gave up waiting for a sibling workgroup (the GPU was busy
                                 with other work for seconds): the container may be fine.  The synchronous entry points and the
                                 reference-named API decode again on one workgroup per block by themselves; the stream-ordered
                                 (*_async) entry points report it in *d_status -- decode again with decode variant 4 */
};

/* =====================================================================================
 * (2) Device-resident path
 * ================================================================================== */

typedef struct tsqa_ctx tsqa_ctx;   /* one per (process, device): streams + scratch in HBM */

/* device < 0: use the current HIP device.  Scratch is grown on demand and kept.
 * A context owns ONE set of scratch buffers (block slots, sizes, frame tables): calls on the same context must be
 * ordered on one stream at a time -- issue the next call on the same stream, or after the previous one has drained.
 * Two streams that want to run concurrently take two contexts. */
int         tsqa_create(int device, tsqa_ctx **out);
void        tsqa_destroy(tsqa_ctx *ctx);
const char *tsqa_last_error(const tsqa_ctx *ctx);
int         tsqa_device_id(const tsqa_ctx *ctx);

/* ceil(n / TSQ_BLOCK_SZ): the job split of tsq_threads.cpp:313. */
size_t tsqa_block_count(size_t n);
/* Capacity that always holds the container of an n-byte input:
 * 16 + n_blocks * (3 + TSQ_OUTPUT_SZ)  (tsq_threads.cpp:339). */
size_t tsqa_container_bound(size_t n);

/* What this library was compiled as: a static string of the form
 *   "arch=gfx950 timing_only=0 instrumented=0 ab_variants=0 lm=4 lf=3 records=11 [switches: none]"
 * timing_only = 1 means a TSQ_X_* switch that gives WRONG streams on purpose (an experiment library, csrc/tsq_experiment.h);
 * the product library always reports timing_only=0 instrumented=0 and "switches: none" (tests/test_abi_cpu.py asserts it). */
const char* tsqa_build_info(void);

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

/*
 * Sharded operation: blocks are independent, so a job can be cut across devices or ranks --
 * block i -> worker i % num_cores in the reference (tsq_threads.cpp:71,463).  These two entry points work on
 * whatever subset of a job's blocks a device owns and leave the gather (frames in block order,
 * tsq_threads.cpp:192-275; blocks at their offsets, :648) to the caller.
 *
 * tsqa_encode_blocks_async: block b of the call is read at d_in + b * stride; every block is TSQ_BLOCK_SZ long
 * except the last (last_len), and each is IMMEDIATELY followed by its look-ahead bytes (the first 128 bytes of
 * the block that follows it in the job; zeros after the job's last block) -- so stride = TSQ_BLOCK_SZ for one
 * contiguous buffer, TSQ_BLOCK_SZ + 128 for a shard that holds every N-th block.  Stream b lands at
 * d_slots + b * TSQ_OUTPUT_SZ, its size in d_sizes[b] (both device memory).  Replaces tsqInit + tsqEncode per
 * owned block (tsq_threads.cpp:176-177).
 */
int tsqa_encode_blocks_async(tsqa_ctx *ctx, const void *d_in, uint32_t n_blocks, size_t stride, uint32_t last_len,
                             uint32_t ext, void *d_slots, uint32_t *d_sizes, int32_t *d_status, void *hip_stream);

/* One block stream to decode: where it starts (relative to d_streams), where its bytes go (relative to d_out). */
typedef struct tsqa_frame {
    uint64_t stream_at;   /* first byte of the block stream (the u24 size header) */
    uint64_t out_at;      /* where the decoded block goes */
    uint32_t stream_len;  /* frame & 0x7FFFFF (tsq_threads.cpp:513-517) */
    uint32_t ext;         /* frame >> 23 */
    uint32_t out_len;     /* the stream's u24 header: decoded bytes (<= TSQ_BLOCK_SZ) */
    uint32_t pad;
} tsqa_frame;

/* tsqa_decode_blocks_async: decode n_blocks streams described by d_frames (device memory).  Replaces tsqDecode
 * per owned block (tsq_threads.cpp:590).  *d_status becomes TSQA_ERR_STREAM on a malformed stream.
 * Trust: the descriptors may be taken from an untrusted container.  The kernels refuse (TSQA_ERR_STREAM) a stream_len
 * below 3 or above TSQ_OUTPUT_SZ and an out_len above TSQ_BLOCK_SZ, and never read a stream beyond stream_len bytes nor write
 * a block beyond out_len bytes.  stream_at and out_at are the CALLER's responsibility: stream_at + stream_len must lie inside
 * d_streams and out_at + out_len inside d_out (the library has no way to know the sizes of those allocations). */
int tsqa_decode_blocks_async(tsqa_ctx *ctx, const void *d_streams, const tsqa_frame *d_frames, uint32_t n_blocks,
                             void *d_out, int32_t *d_status, void *hip_stream);

/* The gather / scatter that goes with them: one DMA per owned block between its slot in HBM (b * TSQ_OUTPUT_SZ) and
 * its frame in a container in HOST memory (pinned or hipHostRegister'ed for full DMA speed).  frame_at[b] is the
 * container offset of block b's three frame bytes (16 + sum over earlier blocks of 3 + size); sizes and frame_at are
 * host arrays.  to_host also writes the frame bytes (size | ext << 23, tsq_threads.cpp:218-219). */
int tsqa_frames_to_host_async(tsqa_ctx *ctx, const void *d_slots, const uint32_t *sizes, const uint64_t *frame_at,
                              uint32_t n_blocks, uint32_t ext, void *host_container, void *hip_stream);
int tsqa_frames_from_host_async(tsqa_ctx *ctx, const void *host_container, const uint64_t *frame_at, const uint32_t *sizes,
                                uint32_t n_blocks, void *d_streams, void *hip_stream);

/* Kernel timing for bench.py: when enabled, every encode / decode kernel launch is bracketed by
 * HIP events recorded on the stream it is launched on (up to 256 launches are kept).
 * tsqa_profile_read waits for those events and returns, per kernel, the summed elapsed
 * milliseconds and the number of launches since the last read. */
int tsqa_profile_enable(tsqa_ctx *ctx, int on);
int tsqa_profile_read(tsqa_ctx *ctx, double *encode_ms, uint32_t *encode_launches,
                      double *decode_ms, uint32_t *decode_launches);
/* The same for whole calls: tsqa_compress_device_async (encode kernel + container pack) and
 * tsqa_decompress_device_async (frame walk + decode kernel). */
int tsqa_profile_read_calls(tsqa_ctx *ctx, double *compress_ms, uint32_t *compress_calls,
                            double *decompress_ms, uint32_t *decompress_calls);

/* One step of a block-sharded job on one rank (block b of the job belongs to rank b % world; what bench.py --gpus N and
 * turbosqueeze_amd/sharding.py run).  The only exchange between ranks is the all-gather of the u32 stream sizes, made by the
 * caller (torch.distributed / RCCL); everything else is here, one call per side:
 *
 * tsqa_frame_offsets / tsqa_walk_frames: host-only helpers (no device): the writer's prefix sum of 3 + size
 *   (tsq_threads.cpp:226-239) and the reader's frame walk with validation (tsq_threads.cpp:513-524) over a container in host
 *   memory.  TSQA_ERR_FORMAT on a malformed container.
 * tsqa_sharded_place_async: after tsqa_encode_blocks_async and the all-gather, every owned stream (block b's at
 *   d_slots + (b / world) * TSQ_OUTPUT_SZ) goes by DMA to its final place in ONE container in host memory, with its three
 *   frame bytes; rank 0 also writes the 16-byte header.  *container_size is the same on every rank.
 * tsqa_sharded_fetch_decode_async: walks the container, brings this rank's frames back to d_streams (k-th owned frame at
 *   k * TSQ_OUTPUT_SZ) and decodes them back to back into d_out (k-th owned block at k * TSQ_BLOCK_SZ).  The container is not
 *   trusted: its whole frame walk is validated against container_size and against streams_cap / out_cap (the bytes the two device
 *   buffers hold) before any copy is enqueued; TSQA_ERR_FORMAT otherwise, with nothing written.
 * Both replace the Python loops of round 2's sharding.py; the host container should be pinned / hipHostRegister'ed. */
int tsqa_frame_offsets(const uint32_t *sizes, uint32_t n_blocks, uint64_t *frame_at, uint64_t *container_size);
int tsqa_walk_frames(const void *container, size_t size, uint32_t cap_blocks, uint64_t *frame_at, uint32_t *sizes, uint32_t *ext,
                     uint32_t *out_len, uint32_t *n_blocks, uint64_t *total);
int tsqa_sharded_place_async(tsqa_ctx *ctx, const void *d_slots, const uint32_t *all_sizes, uint32_t n_blocks, uint64_t n_total,
                             uint32_t rank, uint32_t world, uint32_t ext, void *host_container, size_t host_cap,
                             uint64_t *container_size, void *hip_stream);
int tsqa_sharded_fetch_decode_async(tsqa_ctx *ctx, const void *host_container, size_t container_size, uint32_t rank, uint32_t world,
                                    void *d_streams, size_t streams_cap, void *d_out, size_t out_cap, int32_t *d_status, uint64_t *total,
                                    void *hip_stream);
/* When *d_status of tsqa_sharded_fetch_decode_async reads TSQA_ERR_STALL: the owned frames and their descriptors are still on the
 * device; this decodes them again with one workgroup per block (which waits for nobody) and reports in *d_status again. */
int tsqa_sharded_decode_again_async(tsqa_ctx *ctx, const void *d_streams, void *d_out, int32_t *d_status, void *hip_stream);

/* The measured copy bandwidth of this GPU (bytes read + bytes written per second, GB/s = 1e9 B/s) by a plain
 * grid-stride 16-byte copy kernel over `bytes` of HBM: the second denominator beside the 8 TB/s specification when a
 * kernel is priced against the HBM roofline. */
int tsqa_measure_copy(tsqa_ctx *ctx, size_t bytes, int reps, double *best_gbps, double *median_gbps);
/* which launch shape the last tsqa_measure_copy on this context chose (text; "" before the first probe) */
const char *tsqa_copy_probe_shape(const tsqa_ctx *ctx);

/* Kernel variant selection.  Encoder: 0 = default (staged encoder, one workgroup of fourteen working wavefronts per block; its lean layout -- twelve -- by itself when there are
 * more blocks than CUs), 1 = serial kernel (one lane walks the block; correctness baseline), 6 = force the lean layout
 * (two blocks per CU), 7 = never lean.  Decoder: 0 = default (byte-lane decoder; on two workgroups per block when a
 * launch has at most half as many blocks as the device has CUs, three -- two of them parsing alternate windows of the stream --
 * when it has at most a third), 1 = serial kernel, 3 = several workgroups per block whatever the block count (three when the
 * CUs allow), 4 = always one, 5 = always three, 6 = always two.  The previous round's production encoder (encoder 5) exists only in the A/B library
 * built by `make ab`; the product library rejects it. */
void tsqa_set_kernel_variant(tsqa_ctx *ctx, int encode_variant, int decode_variant);
/* How many polls (of ~0.1 us) a decode on several workgroups per block waits for a sibling workgroup before it reports
 * TSQA_ERR_STALL (default 2^24: seconds).  Tests set it to 1 to exercise the retry. */
void tsqa_set_decode_wait_limit(tsqa_ctx *ctx, uint32_t polls);

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
 * hold TSQ_OUTPUT_SZ bytes.  As in the reference (tsq_encode.cpp:74,108,126,162; its scheduler hands workers pointers
 * into the caller's contiguous buffer, tsq_threads.cpp:109) the encoder looks up to 128 bytes past
 * inputBlock[inputSize-1]: what is readable there is used (so a loop over the blocks of one buffer gives the
 * reference's bytes), what is not mapped is seen as zeros instead of faulting.  The look-ahead is taken with
 * process_vm_readv(2) on the calling process; where a sandbox refuses that call (EPERM / ENOSYS) every block sees zeros behind
 * it -- the library says so once on stderr -- and TSQ_AMD_ENCODE_NO_LOOKAHEAD=1 asks for that behaviour explicitly.  Callers that
 * must not depend on either use tsqa_encode_blocks_async, which takes the look-ahead bytes as part of its input layout. */
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

/* tsqEncode takes the reference encoder's look-ahead (the ~67 bytes it reads behind inputBlock[inputSize-1],
 * tsq_encode.cpp:74,108,126,162) with process_vm_readv, which stops at an unmapped page instead of faulting.  Where a seccomp
 * profile or sandbox refuses that call the look-ahead is seen as zeros and a loop over the blocks of ONE buffer no longer gives the
 * container's streams at block edges (every stream stays valid).  One line on stderr says so the first time; this call lets a
 * caller detect it: 0 = not yet known (no tsqEncode call so far), 1 = look-ahead read from the caller's memory,
 * 2 = refused by the system (zeros), 3 = switched off with TSQ_AMD_ENCODE_NO_LOOKAHEAD. */
int tsqa_encode_lookahead_state(void);

#ifdef __cplusplus
}
#endif
#endif /* TURBOSQUEEZE_AMD_H */
