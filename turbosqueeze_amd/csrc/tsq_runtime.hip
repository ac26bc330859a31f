// tsq_runtime.hip -- device context, kernel launches and the tsqa_* device-resident C ABI.
//
// One tsqa_ctx per (process, device).  All work of a call is enqueued on ONE HIP stream in
// this order (compress):  encode kernel (one workgroup per 4 MiB block)  ->  pack_scan (frame
// offsets, header, frame bytes)  ->  pack_copy (streams into the container);
// (decompress):  frame_walk  ->  decode kernel (one workgroup per block).
// There is no host computation on the data path and no CPU fallback.
#include "tsq_internal.h"

#include "tsq_common.cuh"
#include "tsq_container.cuh"
#include "tsq_serial.cuh"
#include "tsq_launch.cuh"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

using namespace tsq;

#define TSQ_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->set_error("%s failed: %s", #call, hipGetErrorString(e_));                     \
            return TSQA_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

void tsqa_ctx::set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, sizeof(err), fmt, ap);
    va_end(ap);
}

extern "C" size_t tsqa_block_count(size_t n) { return (n + kBlockSize - 1) / kBlockSize; }

extern "C" size_t tsqa_container_bound(size_t n) { return 16 + tsqa_block_count(n) * (size_t)(3 + kSlotSize); }

// What this library was compiled as (csrc/tsq_experiment.h): the product reports no timing-only switch and no instrumentation.
#define TSQ_STR2(x) #x
#define TSQ_STR(x) TSQ_STR2(x)
extern "C" const char* tsqa_build_info(void)
{
    return "arch=gfx950 timing_only=" TSQ_STR(TSQ_TIMING_ONLY_BUILD) " instrumented=" TSQ_STR(TSQ_INSTRUMENTED_BUILD)
#ifdef TSQ_AB_VARIANTS
           " ab_variants=1"
#else
           " ab_variants=0"
#endif
           " lm=" TSQ_STR(TSQ_LM) " lf=" TSQ_STR(TSQ_LF) " records=" TSQ_STR(TSQ_RECORDS) " [switches:"
#ifdef TSQ_X_NOHAZ
           " TSQ_X_NOHAZ"
#endif
#ifdef TSQ_X_NOCOMMITWAIT
           " TSQ_X_NOCOMMITWAIT"
#endif
#ifdef TSQ_X_NOPATCH
           " TSQ_X_NOPATCH"
#endif
#ifdef TSQ_X_FAKE_TABLE
           " TSQ_X_FAKE_TABLE"
#endif
#ifdef TSQ_X_FAKE_CAND
           " TSQ_X_FAKE_CAND"
#endif
#ifdef TSQ_X_FREE_QUERY
           " TSQ_X_FREE_QUERY"
#endif
#ifdef TSQ_X_DELAY_STAGE
           " TSQ_X_DELAY_STAGE"
#endif
#ifdef TSQ_STATS
           " TSQ_STATS"
#endif
#ifdef TSQ_SPINS
           " TSQ_SPINS"
#endif
#ifdef TSQ_TRACEONLY
           " TSQ_TRACEONLY"
#endif
#ifdef TSQ_JITTER
           " TSQ_JITTER"
#endif
#if !TSQ_TIMING_ONLY_BUILD && !TSQ_INSTRUMENTED_BUILD
           " none"
#endif
           "]";
}

extern "C" int tsqa_create(int device, tsqa_ctx** out)
{
    if (!out) return TSQA_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TSQA_ERR_NO_DEVICE;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return TSQA_ERR_NO_DEVICE; }
    if (device >= count) return TSQA_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return TSQA_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "turbosqueeze_amd: device %d is %s; this library carries gfx950 code only\n", device, prop.gcnArchName);
        return TSQA_ERR_NO_DEVICE;
    }
    tsqa_ctx* c = new (std::nothrow) tsqa_ctx();
    if (!c) return TSQA_ERR_ARG;
    c->device = device;
    c->n_cus = prop.multiProcessorCount;
    // (tests: the reference-named API makes its contexts itself; the wait limit of its multi-workgroup decodes comes from here)
    if (const char* e = getenv("TSQ_AMD_DECODE_WAIT_LIMIT")) { const long v = atol(e); if (v > 0) c->decode_wait_limit = (uint32_t)v; }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return TSQA_ERR_HIP;
    }
    if (hipMalloc(&c->d_size, sizeof(uint64_t)) != hipSuccess || hipMalloc(&c->d_status, sizeof(int32_t)) != hipSuccess) {
        tsqa_destroy(c);
        return TSQA_ERR_HIP;
    }
    *out = c;
    return TSQA_OK;
}

extern "C" void tsqa_destroy(tsqa_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (hipEvent_t e : c->prof_pool) if (e) (void)hipEventDestroy(e);
    (void)hipFree(c->slots); (void)hipFree(c->tables); (void)hipFree(c->sizes); (void)hipFree(c->frame_at);
    (void)hipFree(c->frames); (void)hipFree(c->d_size); (void)hipFree(c->d_status);
    (void)hipFree(c->duo_ring); (void)hipFree(c->duo_flags);
    if (c->host_frames) (void)hipHostFree(c->host_frames);
    if (c->host_frames_copied) (void)hipEventDestroy(c->host_frames_copied);
    delete c;
}

extern "C" const char* tsqa_last_error(const tsqa_ctx* c) { return c ? c->err : "null context"; }
extern "C" int tsqa_device_id(const tsqa_ctx* c) { return c ? c->device : -1; }
extern "C" void tsqa_set_kernel_variant(tsqa_ctx* c, int ev, int dv) { if (c) { c->enc_variant = ev; c->dec_variant = dv; } }
extern "C" void tsqa_set_decode_wait_limit(tsqa_ctx* c, uint32_t polls) { if (c) c->decode_wait_limit = polls ? polls : 1u; }

// Scratch in HBM, grown on demand and kept: slots (TSQ_OUTPUT_SZ per block, the reference's
// per-block output buffer, tsq_context.cpp:89-143), per-block sizes, frame offsets, frame
// descriptors and one 256 KiB position table per block for the encoders (want_tables).
int tsqa_ctx::reserve(size_t n_blocks, bool want_tables, bool want_slots)
{
    (void)hipSetDevice(device);
    if (n_blocks > cap_blocks) {
        size_t nb = n_blocks;
        (void)hipStreamSynchronize(stream);
        (void)hipFree(sizes); (void)hipFree(frame_at); (void)hipFree(frames);
        sizes = nullptr; frame_at = nullptr; frames = nullptr; cap_blocks = 0;
        forget_sharded();                                // (the descriptors of a sharded decode went with `frames`)
        TSQ_HIP(this, hipMalloc(&sizes, nb * sizeof(uint32_t)));
        TSQ_HIP(this, hipMalloc(&frame_at, (nb + 1) * sizeof(uint64_t)));
        TSQ_HIP(this, hipMalloc(&frames, nb * sizeof(FrameInfo)));
        cap_blocks = nb;
    }
    if (want_slots && n_blocks > cap_slots) {        // (callers that bring their own slots -- the sharded block API -- never pay for these)
        (void)hipStreamSynchronize(stream);
        (void)hipFree(slots); slots = nullptr; cap_slots = 0;
        TSQ_HIP(this, hipMalloc(&slots, n_blocks * (size_t)kSlotSize + 256));
        cap_slots = n_blocks;
    }
    if (want_tables && n_blocks > cap_tables) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(tables); tables = nullptr; cap_tables = 0;
        TSQ_HIP(this, hipMalloc(&tables, n_blocks * (size_t)kHashEntries * sizeof(uint16_t)));
        cap_tables = n_blocks;
    }
    return TSQA_OK;
}

// Scratch of the two-workgroup decoder (tsq_dec_duo.cuh): four chunk records and two counters per block.
int tsqa_ctx::reserve_duo(size_t n_blocks)
{
    (void)hipSetDevice(device);
    if (n_blocks <= cap_duo) return TSQA_OK;
    (void)hipStreamSynchronize(stream);
    (void)hipFree(duo_ring); (void)hipFree(duo_flags);
    duo_ring = nullptr; duo_flags = nullptr; cap_duo = 0;
    TSQ_HIP(this, hipMalloc(&duo_ring, n_blocks * (size_t)DuoCfg::SLOTS * DuoCfg::REC_WORDS * sizeof(uint32_t)));
    TSQ_HIP(this, hipMalloc(&duo_flags, n_blocks * (size_t)DuoCfg::FLAG_STRIDE * sizeof(uint32_t)));
    cap_duo = n_blocks;
    return TSQA_OK;
}

// Frame descriptors built on the host (sharded fetch + decode): pinned, so that their copy to the device is a DMA ordered on the
// caller's stream; the previous call's copy is waited for before they are overwritten.
int tsqa_ctx::reserve_host_frames(size_t n)
{
    (void)hipSetDevice(device);
    if (host_frames_pending) { (void)hipEventSynchronize(host_frames_copied); host_frames_pending = false; }
    if (!host_frames_copied) TSQ_HIP(this, hipEventCreateWithFlags(&host_frames_copied, hipEventDisableTiming));
    if (n > cap_host_frames) {
        if (host_frames) (void)hipHostFree(host_frames);
        host_frames = nullptr; cap_host_frames = 0;
        size_t want = 64; while (want < n) want *= 2;
        TSQ_HIP(this, hipHostMalloc(reinterpret_cast<void**>(&host_frames), want * sizeof(FrameInfo), hipHostMallocDefault));
        cap_host_frames = want;
    }
    host_frame_src.resize(cap_host_frames);
    return TSQA_OK;
}

// ---- kernel timing ----
bool tsqa_ctx::prof_begin(int kind, hipStream_t s)
{
    if (!profiling || prof_used[kind] >= (uint32_t)kProfPairs) return false;
    const size_t at = ((size_t)kind * kProfPairs + prof_used[kind]) * 2;
    prof_used[kind]++;
    (void)hipEventRecord(prof_pool[at], s);
    return true;
}
void tsqa_ctx::prof_end(int kind, hipStream_t s)
{
    const size_t at = ((size_t)kind * kProfPairs + prof_used[kind] - 1) * 2 + 1;
    (void)hipEventRecord(prof_pool[at], s);
}

extern "C" int tsqa_profile_enable(tsqa_ctx* c, int on)
{
    if (!c) return TSQA_ERR_ARG;
    (void)hipSetDevice(c->device);
    if (on && c->prof_pool.empty()) {
        c->prof_pool.resize((size_t)tsqa_ctx::kProfKinds * tsqa_ctx::kProfPairs * 2, nullptr);
        for (auto& e : c->prof_pool)
            if (hipEventCreate(&e) != hipSuccess) {
                // all or nothing: a half-made pool would hand null events to hipEventRecord later
                for (hipEvent_t made : c->prof_pool) if (made) (void)hipEventDestroy(made);
                c->prof_pool.clear();
                c->profiling = false;
                c->set_error("hipEventCreate failed");
                return TSQA_ERR_HIP;
            }
    }
    c->profiling = on != 0;
    return TSQA_OK;
}

static void drain_events(tsqa_ctx* c, int kind, double* ms, uint32_t* count)
{
    double sum = 0; uint32_t n = 0;
    for (uint32_t k = 0; k < c->prof_used[kind]; ++k) {
        const size_t at = ((size_t)kind * tsqa_ctx::kProfPairs + k) * 2;
        float t = 0;
        if (hipEventSynchronize(c->prof_pool[at + 1]) == hipSuccess && hipEventElapsedTime(&t, c->prof_pool[at], c->prof_pool[at + 1]) == hipSuccess) { sum += t; n++; }
    }
    c->prof_used[kind] = 0;
    if (ms) *ms = sum;
    if (count) *count = n;
}

extern "C" int tsqa_profile_read(tsqa_ctx* c, double* enc_ms, uint32_t* enc_n, double* dec_ms, uint32_t* dec_n)
{
    if (!c) return TSQA_ERR_ARG;
    (void)hipSetDevice(c->device);
    if (c->prof_pool.empty()) { if (enc_ms) *enc_ms = 0; if (enc_n) *enc_n = 0; if (dec_ms) *dec_ms = 0; if (dec_n) *dec_n = 0; return TSQA_OK; }
    drain_events(c, 0, enc_ms, enc_n);
    drain_events(c, 1, dec_ms, dec_n);
    return TSQA_OK;
}

extern "C" int tsqa_profile_read_calls(tsqa_ctx* c, double* comp_ms, uint32_t* comp_n, double* decomp_ms, uint32_t* decomp_n)
{
    if (!c) return TSQA_ERR_ARG;
    (void)hipSetDevice(c->device);
    if (c->prof_pool.empty()) { if (comp_ms) *comp_ms = 0; if (comp_n) *comp_n = 0; if (decomp_ms) *decomp_ms = 0; if (decomp_n) *decomp_n = 0; return TSQA_OK; }
    drain_events(c, 2, comp_ms, comp_n);
    drain_events(c, 3, decomp_ms, decomp_n);
    return TSQA_OK;
}

// ---- internal launches (also used by the reference-API layer in tsq_compat.hip) ----

int tsqa_ctx::launch_encode_to(const void* d_in, size_t n, size_t readable, size_t stride, uint32_t ext, uint8_t* slots_out,
                               uint32_t* sizes_out, int32_t* status, hipStream_t s)
{
    const uint32_t nb = (uint32_t)tsqa_block_count(n);
    int rc = reserve(nb, true, false);                   // (the streams go to the caller's slots: the context's own are not needed here)
    if (rc) return rc;
    const bool timed = prof_begin(0, s);
    rc = launch_encode_kernels(this, static_cast<const uint8_t*>(d_in), n, readable, stride, ext, slots_out, sizes_out, status, s);
    if (rc) { if (timed) prof_used[0]--; return rc; }     // (the pair's end event was never recorded: give the pair back)
    if (timed) prof_end(0, s);
    TSQ_HIP(this, hipGetLastError());
    return TSQA_OK;
}

int tsqa_ctx::launch_encode(const void* d_in, size_t n, size_t readable, uint32_t ext, int32_t* status, hipStream_t s)
{
    int rc = reserve((uint32_t)tsqa_block_count(n), true);
    if (rc) return rc;
    return launch_encode_to(d_in, n, readable, kBlockSize, ext, slots, sizes, status, s);
}

int tsqa_ctx::launch_pack(size_t n, uint32_t ext, void* d_out, size_t out_cap, uint64_t* d_out_size, int32_t* status, hipStream_t s)
{
    const uint32_t nb = (uint32_t)tsqa_block_count(n);
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(256), 0, s, sizes, nb, (uint64_t)n, ext,
                       static_cast<uint8_t*>(d_out), (uint64_t)out_cap, frame_at, d_out_size, status);
    const uint32_t pieces = (kSlotSize + kPackPiece - 1) / kPackPiece + 1;
    hipLaunchKernelGGL(pack_copy_kernel, dim3(pieces, nb), dim3(256), 0, s, slots, sizes, frame_at,
                       static_cast<uint8_t*>(d_out), status);
    TSQ_HIP(this, hipGetLastError());
    return TSQA_OK;
}

int tsqa_ctx::launch_decode_frames(const void* d_streams, const FrameInfo* d_frames, uint32_t n_blocks, void* d_out, int32_t* status, hipStream_t s, int variant)
{
    const bool timed = prof_begin(1, s);
    int rc = launch_decode_kernels(this, static_cast<const uint8_t*>(d_streams), d_frames, n_blocks, static_cast<uint8_t*>(d_out), status, s, variant);
    if (rc) { if (timed) prof_used[1]--; return rc; }
    if (timed) prof_end(1, s);
    TSQ_HIP(this, hipGetLastError());
    return TSQA_OK;
}

int tsqa_ctx::launch_decode(const void* d_container, uint32_t n_blocks, void* d_out, int32_t* status, hipStream_t s, int variant)
{
    return launch_decode_frames(d_container, frames, n_blocks, d_out, status, s, variant);
}

// ---- public device-resident entry points ----

extern "C" int tsqa_compress_device_async(tsqa_ctx* c, const void* d_in, size_t n, void* d_out, size_t out_cap,
                                          uint64_t* d_out_size, int32_t* d_status, uint32_t ext, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_in || !d_out || !d_out_size || !d_status || n == 0) { c->set_error("compress: null pointer or zero size"); return TSQA_ERR_ARG; }
    if (out_cap < 16 + 6 * tsqa_block_count(n)) { c->set_error("compress: output capacity too small"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    TSQ_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    const bool timed = c->prof_begin(2, s);
    int rc = c->launch_encode(d_in, n, n, ext, d_status, s);
    if (rc) { if (timed) c->prof_used[2]--; return rc; }
    rc = c->launch_pack(n, ext, d_out, out_cap, d_out_size, d_status, s);
    if (timed) c->prof_end(2, s);
    return rc;
}

static int status_to_rc(tsqa_ctx* c, int32_t st, const char* what)
{
    if (st == 0) return TSQA_OK;
    c->set_error("%s: device reported status %d", what, st);
    return st;
}

extern "C" int tsqa_compress_device(tsqa_ctx* c, const void* d_in, size_t n, void* d_out, size_t out_cap,
                                    size_t* out_size, uint32_t ext, void* hip_stream)
{
    if (!c || !out_size) return TSQA_ERR_ARG;
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    int rc = tsqa_compress_device_async(c, d_in, n, d_out, out_cap, c->d_size, c->d_status, ext, s);
    if (rc) return rc;
    uint64_t sz = 0; int32_t st = 0;
    TSQ_HIP(c, hipMemcpyAsync(&sz, c->d_size, sizeof(sz), hipMemcpyDeviceToHost, s));
    TSQ_HIP(c, hipMemcpyAsync(&st, c->d_status, sizeof(st), hipMemcpyDeviceToHost, s));
    TSQ_HIP(c, hipStreamSynchronize(s));
    *out_size = (size_t)sz;
    return status_to_rc(c, st, "compress");
}

// (variant < 0: the context's decode variant; the retry after TSQA_ERR_STALL passes 4 without touching the context's setting)
static int decompress_device_async_impl(tsqa_ctx* c, const void* d_in, size_t n, uint32_t n_blocks, void* d_out,
                                        size_t out_cap, uint64_t* d_out_size, int32_t* d_status, void* hip_stream, int variant)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_in || !d_out || !d_out_size || !d_status || n < 16 || n_blocks == 0) { c->set_error("decompress: bad argument"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    int rc = c->reserve(n_blocks, false, false);
    if (rc) return rc;
    c->forget_sharded();                                 // the frame walk below overwrites c->frames
    TSQ_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    const bool timed = c->prof_begin(3, s);
    hipLaunchKernelGGL(frame_walk_kernel, dim3(1), dim3(64), 0, s, static_cast<const uint8_t*>(d_in), (uint64_t)n, n_blocks,
                       (uint64_t)out_cap, c->frames, d_out_size, d_status);
    rc = c->launch_decode(d_in, n_blocks, d_out, d_status, s, variant);
    if (timed) c->prof_end(3, s);
    return rc;
}

extern "C" int tsqa_decompress_device_async(tsqa_ctx* c, const void* d_in, size_t n, uint32_t n_blocks, void* d_out,
                                            size_t out_cap, uint64_t* d_out_size, int32_t* d_status, void* hip_stream)
{
    return decompress_device_async_impl(c, d_in, n, n_blocks, d_out, out_cap, d_out_size, d_status, hip_stream, -1);
}

// ---- sharded operation: a device owns some of a job's blocks ----

extern "C" int tsqa_encode_blocks_async(tsqa_ctx* c, const void* d_in, uint32_t n_blocks, size_t stride, uint32_t last_len,
                                        uint32_t ext, void* d_slots, uint32_t* d_sizes, int32_t* d_status, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_in || !d_slots || !d_sizes || !d_status || n_blocks == 0 || last_len == 0 || last_len > kBlockSize || stride < kBlockSize) {
        c->set_error("encode_blocks: bad argument");
        return TSQA_ERR_ARG;
    }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    TSQ_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    // the virtual total gives every block but the last 4 MiB; what may be read ends with the last block's look-ahead
    const size_t n = (size_t)(n_blocks - 1) * kBlockSize + last_len;
    const size_t tail = stride > kBlockSize ? (stride - kBlockSize < 128 ? stride - kBlockSize : 128) : 0;
    const size_t readable = (size_t)(n_blocks - 1) * stride + last_len + tail;
    return c->launch_encode_to(d_in, n, readable, stride, ext, static_cast<uint8_t*>(d_slots), d_sizes, d_status, s);
}

extern "C" int tsqa_decode_blocks_async(tsqa_ctx* c, const void* d_streams, const tsqa_frame* d_frames, uint32_t n_blocks,
                                        void* d_out, int32_t* d_status, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_streams || !d_frames || !d_out || !d_status || n_blocks == 0) { c->set_error("decode_blocks: bad argument"); return TSQA_ERR_ARG; }
    static_assert(sizeof(tsqa_frame) == sizeof(FrameInfo), "public frame descriptor = kernel frame descriptor");
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    TSQ_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    return c->launch_decode_frames(d_streams, reinterpret_cast<const FrameInfo*>(d_frames), n_blocks, d_out, d_status, s);
}

extern "C" int tsqa_decompress_device(tsqa_ctx* c, const void* d_in, size_t n, void* d_out, size_t out_cap,
                                      size_t* out_size, void* hip_stream)
{
    if (!c || !out_size) return TSQA_ERR_ARG;
    if (!d_in || n < 16) { c->set_error("decompress: bad argument"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    uint8_t head[16];
    TSQ_HIP(c, hipMemcpyAsync(head, d_in, 16, hipMemcpyDeviceToHost, s));
    TSQ_HIP(c, hipStreamSynchronize(s));
    if (memcmp(head, "TSQ1", 4) != 0) { c->set_error("decompress: bad magic"); return TSQA_ERR_FORMAT; }   // tsq_threads.cpp:732-752
    uint32_t nb; uint64_t total;
    memcpy(&nb, head + 4, 4); memcpy(&total, head + 8, 8);
    if (nb == 0) { c->set_error("decompress: n_blocks == 0"); return TSQA_ERR_FORMAT; }                     // tsq_threads.cpp:759-768
    if (total > out_cap) { c->set_error("decompress: output capacity %zu < %llu", out_cap, (unsigned long long)total); return TSQA_ERR_ARG; }
    if ((size_t)nb > n / 6) { c->set_error("decompress: n_blocks larger than the container"); return TSQA_ERR_FORMAT; }
    int rc = tsqa_decompress_device_async(c, d_in, n, nb, d_out, out_cap, c->d_size, c->d_status, s);
    if (rc) return rc;
    uint64_t sz = 0; int32_t st = 0;
    TSQ_HIP(c, hipMemcpyAsync(&sz, c->d_size, sizeof(sz), hipMemcpyDeviceToHost, s));
    TSQ_HIP(c, hipMemcpyAsync(&st, c->d_status, sizeof(st), hipMemcpyDeviceToHost, s));
    TSQ_HIP(c, hipStreamSynchronize(s));
    if (st == kErrStall) {
        // a workgroup of a several-workgroups-per-block decode did not get onto the GPU in time (other work held the CUs): the
        // container is not at fault -- once more with one workgroup per block, which waits for nobody
        rc = decompress_device_async_impl(c, d_in, n, nb, d_out, out_cap, c->d_size, c->d_status, s, 4);
        if (rc) return rc;
        TSQ_HIP(c, hipMemcpyAsync(&sz, c->d_size, sizeof(sz), hipMemcpyDeviceToHost, s));
        TSQ_HIP(c, hipMemcpyAsync(&st, c->d_status, sizeof(st), hipMemcpyDeviceToHost, s));
        TSQ_HIP(c, hipStreamSynchronize(s));
    }
    *out_size = (size_t)sz;
    return status_to_rc(c, st, "decompress");
}

// Host gather / scatter of a shard's frames (what compression_write_worker and decompression_read_worker do with
// memcpy, tsq_threads.cpp:226-239,513-524): one DMA per owned block between its slot in HBM and its place in the
// container in host memory.  The three frame bytes are written / skipped here.
extern "C" int tsqa_frames_to_host_async(tsqa_ctx* c, const void* d_slots, const uint32_t* sizes, const uint64_t* frame_at,
                                         uint32_t n_blocks, uint32_t ext, void* host_container, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_slots || !sizes || !frame_at || !host_container) { c->set_error("frames_to_host: null pointer"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    uint8_t* base = static_cast<uint8_t*>(host_container);
    for (uint32_t b = 0; b < n_blocks; ++b) {
        if (sizes[b] < 3 || sizes[b] > kSlotSize) { c->set_error("frames_to_host: block %u has size %u", b, sizes[b]); return TSQA_ERR_ARG; }
        const uint32_t frame = sizes[b] | (ext ? 0x800000u : 0u);                     // tsq_threads.cpp:218-219
        uint8_t* p = base + frame_at[b];
        p[0] = (uint8_t)frame; p[1] = (uint8_t)(frame >> 8); p[2] = (uint8_t)(frame >> 16);
        TSQ_HIP(c, hipMemcpyAsync(p + 3, static_cast<const uint8_t*>(d_slots) + (size_t)b * kSlotSize, sizes[b], hipMemcpyDeviceToHost, s));
    }
    return TSQA_OK;
}

extern "C" int tsqa_frames_from_host_async(tsqa_ctx* c, const void* host_container, const uint64_t* frame_at, const uint32_t* sizes,
                                           uint32_t n_blocks, void* d_streams, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_streams || !sizes || !frame_at || !host_container) { c->set_error("frames_from_host: null pointer"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    const uint8_t* base = static_cast<const uint8_t*>(host_container);
    for (uint32_t b = 0; b < n_blocks; ++b) {
        if (sizes[b] < 3 || sizes[b] > kSlotSize) { c->set_error("frames_from_host: block %u has size %u", b, sizes[b]); return TSQA_ERR_FORMAT; }
        TSQ_HIP(c, hipMemcpyAsync(static_cast<uint8_t*>(d_streams) + (size_t)b * kSlotSize, base + frame_at[b] + 3, sizes[b], hipMemcpyHostToDevice, s));
    }
    return TSQA_OK;
}

// ---- one step of a block-sharded job on this rank (block b of the job belongs to rank b % world) ----
// Host-only helpers first (no device involved): the writer's frame offsets and the reader's frame walk
// (tsq_threads.cpp:226-239,513-524) over a container in host memory.
extern "C" int tsqa_frame_offsets(const uint32_t* sizes, uint32_t n_blocks, uint64_t* frame_at, uint64_t* container_size)
{
    if (!sizes || !frame_at || !container_size) return TSQA_ERR_ARG;
    uint64_t at = 16;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        if (sizes[b] < 3 || sizes[b] > kSlotSize) return TSQA_ERR_ARG;
        frame_at[b] = at;
        at += 3ull + sizes[b];
    }
    *container_size = at;
    return TSQA_OK;
}

extern "C" int tsqa_walk_frames(const void* container, size_t size, uint32_t cap_blocks, uint64_t* frame_at, uint32_t* sizes, uint32_t* ext,
                                uint32_t* out_len, uint32_t* n_blocks, uint64_t* total)
{
    if (!container || !frame_at || !sizes || !ext || !out_len || !n_blocks || !total) return TSQA_ERR_ARG;
    const uint8_t* p = static_cast<const uint8_t*>(container);
    if (size < 16 || memcmp(p, "TSQ1", 4) != 0) return TSQA_ERR_FORMAT;                    // tsq_threads.cpp:732-752
    uint32_t nb; uint64_t tot;
    memcpy(&nb, p + 4, 4); memcpy(&tot, p + 8, 8);
    if (nb == 0 || (size_t)nb > (size - 16) / 6 || nb > cap_blocks) return TSQA_ERR_FORMAT;  // tsq_threads.cpp:759-768
    uint64_t at = 16, sum = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        if (at + 6 > size) return TSQA_ERR_FORMAT;
        const uint32_t frame = (uint32_t)p[at] | ((uint32_t)p[at + 1] << 8) | ((uint32_t)p[at + 2] << 16);
        const uint32_t len = frame & 0x7FFFFFu;                                             // tsq_threads.cpp:513-517
        if (len < 3 || len > kSlotSize || at + 3 + len > size) return TSQA_ERR_FORMAT;
        const uint32_t usize = (uint32_t)p[at + 3] | ((uint32_t)p[at + 4] << 8) | ((uint32_t)p[at + 5] << 16);
        if (usize > kBlockSize) return TSQA_ERR_FORMAT;
        frame_at[b] = at; sizes[b] = len; ext[b] = frame >> 23; out_len[b] = usize;
        sum += usize;
        at += 3ull + len;
    }
    if (sum != tot) return TSQA_ERR_FORMAT;
    *n_blocks = nb; *total = tot;
    return TSQA_OK;
}

// After the encode of the owned blocks and the all-gather of every block's stream size: this rank's frames go to their final
// place in ONE container in host memory (rank 0 also writes the 16-byte header).  The whole "gather" of the writer thread
// (tsq_threads.cpp:192-275) is this prefix sum and one DMA per owned block.
extern "C" int tsqa_sharded_place_async(tsqa_ctx* c, const void* d_slots, const uint32_t* all_sizes, uint32_t n_blocks, uint64_t n_total,
                                        uint32_t rank, uint32_t world, uint32_t ext, void* host_container, size_t host_cap,
                                        uint64_t* container_size, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_slots || !all_sizes || !host_container || !container_size || world == 0 || rank >= world || n_blocks == 0) { c->set_error("sharded_place: bad argument"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    uint8_t* base = static_cast<uint8_t*>(host_container);
    // every size and the capacity are checked before anything is written or enqueued (no partial container on an error)
    uint64_t at = 16;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t sz = all_sizes[b];
        if (sz < 3 || sz > kSlotSize) { c->set_error("sharded_place: block %u has size %u", b, sz); return TSQA_ERR_ARG; }
        at += 3ull + sz;
    }
    if (at > host_cap) { c->set_error("sharded_place: the host container is too small (%llu > %zu)", (unsigned long long)at, host_cap); return TSQA_ERR_ARG; }
    at = 16;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        const uint32_t sz = all_sizes[b];
        if (b % world == rank) {
            const uint32_t frame = sz | (ext ? 0x800000u : 0u);                          // tsq_threads.cpp:218-219
            uint8_t* p = base + at;
            p[0] = (uint8_t)frame; p[1] = (uint8_t)(frame >> 8); p[2] = (uint8_t)(frame >> 16);
            TSQ_HIP(c, hipMemcpyAsync(p + 3, static_cast<const uint8_t*>(d_slots) + (size_t)(b / world) * kSlotSize, sz, hipMemcpyDeviceToHost, s));
        }
        at += 3ull + sz;
    }
    if (rank == 0) {                                                                      // tsq_threads.cpp:333-335
        memcpy(base, "TSQ1", 4); memcpy(base + 4, &n_blocks, 4); memcpy(base + 8, &n_total, 8);
    }
    *container_size = at;
    return TSQA_OK;
}

// The reader's side: walk the container's frames (host memory), bring this rank's frames to d_streams (frame k of the rank at
// k * TSQ_OUTPUT_SZ) and decode them back to back into d_out (block k of the rank at k * TSQ_BLOCK_SZ).
extern "C" int tsqa_sharded_fetch_decode_async(tsqa_ctx* c, const void* host_container, size_t container_size, uint32_t rank, uint32_t world,
                                               void* d_streams, size_t streams_cap, void* d_out, size_t out_cap, int32_t* d_status, uint64_t* total,
                                               void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!host_container || !d_streams || !d_out || !d_status || !total || world == 0 || rank >= world) { c->set_error("sharded_fetch_decode: bad argument"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    const uint8_t* p = static_cast<const uint8_t*>(host_container);
    if (container_size < 16 || memcmp(p, "TSQ1", 4) != 0) { c->set_error("sharded_fetch_decode: bad magic"); return TSQA_ERR_FORMAT; }
    uint32_t nb; uint64_t tot;
    memcpy(&nb, p + 4, 4); memcpy(&tot, p + 8, 8);
    if (nb == 0 || (size_t)nb > (container_size - 16) / 6) { c->set_error("sharded_fetch_decode: bad block count"); return TSQA_ERR_FORMAT; }
    const uint32_t n_local = nb > rank ? (nb - rank + world - 1) / world : 0;
    // The container is not trusted: the whole frame walk is validated -- against the container's own size and against what the caller's
    // buffers can hold -- before a single copy is enqueued (a container with more, shorter blocks than the job the buffers were sized
    // for must not overrun them).
    if ((uint64_t)n_local * kSlotSize > streams_cap) { c->set_error("sharded_fetch_decode: %u owned frames do not fit d_streams (%zu B)", n_local, streams_cap); return TSQA_ERR_FORMAT; }
    c->forget_sharded();                                 // whatever happens below, an older call's descriptors are not to be decoded again
    if (int rc = c->reserve(n_local ? n_local : 1, false, false)) return rc;
    if (int rc = c->reserve_host_frames(n_local ? n_local : 1)) return rc;
    uint64_t at = 16, sum = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        if (at + 6 > container_size) { c->set_error("sharded_fetch_decode: truncated container"); return TSQA_ERR_FORMAT; }
        const uint32_t frame = (uint32_t)p[at] | ((uint32_t)p[at + 1] << 8) | ((uint32_t)p[at + 2] << 16);
        const uint32_t len = frame & 0x7FFFFFu;
        if (len < 3 || len > kSlotSize || at + 3 + len > container_size) { c->set_error("sharded_fetch_decode: bad frame %u", b); return TSQA_ERR_FORMAT; }
        const uint32_t usize = (uint32_t)p[at + 3] | ((uint32_t)p[at + 4] << 8) | ((uint32_t)p[at + 5] << 16);
        if (usize > kBlockSize) { c->set_error("sharded_fetch_decode: bad block size in frame %u", b); return TSQA_ERR_FORMAT; }
        if (b % world == rank) {
            const uint32_t k = b / world;
            if ((uint64_t)k * kBlockSize + usize > out_cap) { c->set_error("sharded_fetch_decode: owned block %u does not fit d_out (%zu B)", k, out_cap); return TSQA_ERR_FORMAT; }
            FrameInfo f;
            f.stream_at = (uint64_t)k * kSlotSize; f.out_at = (uint64_t)k * kBlockSize; f.stream_len = len; f.ext = frame >> 23; f.out_len = usize; f.pad = 0;
            c->host_frames[k] = f;
            c->host_frame_src[k] = at + 3;
        }
        sum += usize;
        at += 3ull + len;
    }
    if (sum != tot) { c->set_error("sharded_fetch_decode: block sizes do not add up"); return TSQA_ERR_FORMAT; }
    *total = tot;
    TSQ_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    if (n_local == 0) return TSQA_OK;
    for (uint32_t k = 0; k < n_local; ++k)
        TSQ_HIP(c, hipMemcpyAsync(static_cast<uint8_t*>(d_streams) + (size_t)k * kSlotSize, p + c->host_frame_src[k], c->host_frames[k].stream_len, hipMemcpyHostToDevice, s));
    // (the descriptors live in pinned memory: the copy is a real DMA ordered on `s`, and reserve_host_frames has waited for the
    //  previous call's copy before they were overwritten)
    TSQ_HIP(c, hipMemcpyAsync(c->frames, c->host_frames, (size_t)n_local * sizeof(FrameInfo), hipMemcpyHostToDevice, s));
    TSQ_HIP(c, hipEventRecord(c->host_frames_copied, s));
    c->host_frames_pending = true;
    c->sharded_n_local = n_local; c->sharded_streams = d_streams; c->sharded_out = d_out;
    return c->launch_decode_frames(d_streams, c->frames, n_local, d_out, d_status, s);
}

// After *d_status of tsqa_sharded_fetch_decode_async came back TSQA_ERR_STALL: the owned frames and their descriptors are still on the
// device -- decode them again with one workgroup per block (decode variant 4), which waits for nobody.  A GPU of a sharded job holds
// few blocks, so its first attempt always takes the several-workgroups-per-block decoder; a busy GPU must not fail a valid container.
extern "C" int tsqa_sharded_decode_again_async(tsqa_ctx* c, const void* d_streams, void* d_out, int32_t* d_status, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_streams || !d_out || !d_status) { c->set_error("sharded_decode_again: null pointer"); return TSQA_ERR_ARG; }
    if (c->sharded_n_local == 0) { c->set_error("sharded_decode_again: no sharded decode to repeat on this context (none yet, or another call has used the context since)"); return TSQA_ERR_ARG; }
    // the descriptors hold offsets into the buffers of THAT call: a retry into other buffers would decode them against the wrong memory
    if (d_streams != c->sharded_streams || d_out != c->sharded_out) { c->set_error("sharded_decode_again: not the buffers of the sharded decode being repeated"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    TSQ_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    return c->launch_decode_frames(d_streams, c->frames, c->sharded_n_local, d_out, d_status, s, 4);
}

// ---- the second roofline denominator: what a plain device copy reaches on this GPU ----
// MODE 0: grid-stride, four independent 16-byte loads in flight per lane; 1: the same with non-temporal loads and stores;
// 2: one pass, every thread moves four words 256 apart (no loop: as many workgroups as the buffer needs).
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void copy_probe_kernel(const probe_u32x4* __restrict__ src, probe_u32x4* __restrict__ dst, size_t words)
{
    if (MODE == 2) {
        const size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x;
        if (i + 768u < words) {
            const probe_u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256), c = __builtin_nontemporal_load(src + i + 512),
                                d = __builtin_nontemporal_load(src + i + 768);
            __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + 256);
            __builtin_nontemporal_store(c, dst + i + 512); __builtin_nontemporal_store(d, dst + i + 768);
        } else for (size_t k = i; k < words && k < i + 1024u; k += 256u) dst[k] = src[k];
        return;
    }
    const size_t step = (size_t)gridDim.x * 256u;
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    for (; i + 3 * step < words; i += 4 * step) {
        if (MODE == 1) {
            const probe_u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + step), c = __builtin_nontemporal_load(src + i + 2 * step),
                                d = __builtin_nontemporal_load(src + i + 3 * step);
            __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + step);
            __builtin_nontemporal_store(c, dst + i + 2 * step); __builtin_nontemporal_store(d, dst + i + 3 * step);
        } else {
            const probe_u32x4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
            dst[i] = a; dst[i + step] = b; dst[i + 2 * step] = c; dst[i + 3 * step] = d;
        }
    }
    for (; i < words; i += step) dst[i] = src[i];
}

extern "C" int tsqa_measure_copy(tsqa_ctx* c, size_t bytes, int reps, double* best_gbps, double* median_gbps)
{
    if (!c || bytes < (size_t(1) << 20) || reps < 1 || reps > 64) return TSQA_ERR_ARG;
    (void)hipSetDevice(c->device);
    probe_u32x4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t words = bytes / 16;
    int rc = TSQA_OK;
    double rates[64];
    if (hipMalloc(&a, words * 16) != hipSuccess || hipMalloc(&b, words * 16) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
        hipMemsetAsync(a, 0x5a, words * 16, c->stream) != hipSuccess) rc = TSQA_ERR_HIP;
    // the shape that copies fastest is found first -- grid-stride kernels with 8, 16, 32 or 64 workgroups of 256 per CU, with default
    // and with non-temporal accesses, and the one-pass kernel -- then measured `reps` times
    uint32_t grid = (uint32_t)c->n_cus * 8u;
    int mode = 0;
    auto launch = [&](int m, uint32_t g) {
        if (m == 0) hipLaunchKernelGGL(copy_probe_kernel<0>, dim3(g), dim3(256), 0, c->stream, a, b, words);
        else if (m == 1) hipLaunchKernelGGL(copy_probe_kernel<1>, dim3(g), dim3(256), 0, c->stream, a, b, words);
        else hipLaunchKernelGGL(copy_probe_kernel<2>, dim3((uint32_t)((words + 1023u) / 1024u)), dim3(256), 0, c->stream, a, b, words);
    };
    {
        float best_ms = 1e30f;
        for (int m = 0; m < 3 && rc == TSQA_OK; ++m)
            for (uint32_t per_cu = 8; per_cu <= (m == 2 ? 8u : 64u) && rc == TSQA_OK; per_cu *= 2) {
                const uint32_t g = (uint32_t)c->n_cus * per_cu;
                float ms = 1e30f;
                for (int k = 0; k < 2; ++k) {
                    (void)hipEventRecord(e0, c->stream);
                    launch(m, g);
                    (void)hipEventRecord(e1, c->stream);
                    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = TSQA_ERR_HIP; break; }
                }
                if (ms < best_ms) { best_ms = ms; grid = g; mode = m; }
            }
    }
    for (int r = -1; r < reps && rc == TSQA_OK; ++r) {          // r == -1 warms up
        (void)hipEventRecord(e0, c->stream);
        launch(mode, grid);
        (void)hipEventRecord(e1, c->stream);
        float ms = 0;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0) { rc = TSQA_ERR_HIP; break; }
        if (r >= 0) rates[r] = 2.0 * (double)(words * 16) / (ms * 1e-3) / 1e9;    // bytes read + bytes written
    }
    if (rc == TSQA_OK) {
        for (int i = 1; i < reps; ++i) for (int j = i; j > 0 && rates[j] < rates[j - 1]; --j) { double t = rates[j]; rates[j] = rates[j - 1]; rates[j - 1] = t; }
        if (best_gbps) *best_gbps = rates[reps - 1];
        if (median_gbps) *median_gbps = rates[reps / 2];
        snprintf(c->probe_shape, sizeof(c->probe_shape), "copy probe: mode %d (0 grid-stride, 1 grid-stride non-temporal, 2 one pass non-temporal), %u workgroups", mode, mode == 2 ? (uint32_t)((words + 1023u) / 1024u) : grid);
    } else c->set_error("copy probe failed");
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    return rc;
}

extern "C" const char* tsqa_copy_probe_shape(const tsqa_ctx* c) { return c ? c->probe_shape : ""; }

#ifdef TSQ_SPINS
extern "C" int tsqa_debug_spins(uint32_t* out16)
{
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(tsq::g_enc_spins), 20 * sizeof(uint32_t)) == hipSuccess ? TSQA_OK : TSQA_ERR_HIP;
}
#endif
#if defined(TSQ_STATS) || defined(TSQ_TRACEONLY)
extern "C" int tsqa_debug_trace(uint32_t* out4096)
{
    return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(tsq::g_enc_trace), 4096 * sizeof(uint32_t)) == hipSuccess ? TSQA_OK : TSQA_ERR_HIP;
}
#endif
#ifdef TSQ_STATS
// instrumented builds only: counters published by block 0 of the last encode / decode launch
extern "C" int tsqa_debug_stats(unsigned long long* enc64, unsigned long long* dec16)
{
    if (enc64 && hipMemcpyFromSymbol(enc64, HIP_SYMBOL(tsq::g_enc_stats), 64 * sizeof(unsigned long long)) != hipSuccess) return TSQA_ERR_HIP;
    if (dec16 && hipMemcpyFromSymbol(dec16, HIP_SYMBOL(tsq::g_dec_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return TSQA_ERR_HIP;
    return TSQA_OK;
}
extern "C" int tsqa_debug_dec_waves(unsigned long long* out48)
{
    return hipMemcpyFromSymbol(out48, HIP_SYMBOL(tsq::g_dec_wave), 48 * sizeof(unsigned long long)) == hipSuccess ? TSQA_OK : TSQA_ERR_HIP;
}
extern "C" int tsqa_debug_duo_xcc(uint32_t* out2048)
{
    return hipMemcpyFromSymbol(out2048, HIP_SYMBOL(tsq::g_duo_xcc), 2048 * sizeof(uint32_t)) == hipSuccess ? TSQA_OK : TSQA_ERR_HIP;
}
extern "C" int tsqa_debug_syms(uint32_t* out8192)
{
    return hipMemcpyFromSymbol(out8192, HIP_SYMBOL(tsq::g_dbg_syms), 8192 * sizeof(uint32_t)) == hipSuccess ? TSQA_OK : TSQA_ERR_HIP;
}
#endif
