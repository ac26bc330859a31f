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
#include "tsq_fast.cuh"

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
    (void)hipFree(c->slots); (void)hipFree(c->tables); (void)hipFree(c->sizes); (void)hipFree(c->frame_at);
    (void)hipFree(c->frames); (void)hipFree(c->d_size); (void)hipFree(c->d_status);
    delete c;
}

extern "C" const char* tsqa_last_error(const tsqa_ctx* c) { return c ? c->err : "null context"; }
extern "C" int tsqa_device_id(const tsqa_ctx* c) { return c ? c->device : -1; }
extern "C" void tsqa_set_kernel_variant(tsqa_ctx* c, int ev, int dv) { if (c) { c->enc_variant = ev; c->dec_variant = dv; } }

// Scratch in HBM, grown on demand and kept: slots (TSQ_OUTPUT_SZ per block, the reference's
// per-block output buffer, tsq_context.cpp:89-143), per-block sizes, frame offsets, frame
// descriptors and -- for the serial encoder variant only -- one 256 KiB table per block.
int tsqa_ctx::reserve(size_t n_blocks, bool want_tables)
{
    (void)hipSetDevice(device);
    if (n_blocks > cap_blocks) {
        size_t nb = n_blocks;
        (void)hipStreamSynchronize(stream);
        (void)hipFree(slots); (void)hipFree(sizes); (void)hipFree(frame_at); (void)hipFree(frames);
        slots = nullptr; sizes = nullptr; frame_at = nullptr; frames = nullptr; cap_blocks = 0;
        TSQ_HIP(this, hipMalloc(&slots, nb * (size_t)kSlotSize + 256));
        TSQ_HIP(this, hipMalloc(&sizes, nb * sizeof(uint32_t)));
        TSQ_HIP(this, hipMalloc(&frame_at, (nb + 1) * sizeof(uint64_t)));
        TSQ_HIP(this, hipMalloc(&frames, nb * sizeof(FrameInfo)));
        cap_blocks = nb;
    }
    if (want_tables && n_blocks > cap_tables) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(tables); tables = nullptr; cap_tables = 0;
        TSQ_HIP(this, hipMalloc(&tables, n_blocks * (size_t)kHashEntries * sizeof(uint16_t)));
        cap_tables = n_blocks;
    }
    return TSQA_OK;
}

// ---- kernel timing ----
hipEvent_t tsqa_ctx::prof_begin(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, hipStream_t s)
{
    if (!profiling || v.size() >= 256) return nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return nullptr;
    v.emplace_back(a, b);
    (void)hipEventRecord(a, s);
    return a;
}
void tsqa_ctx::prof_end(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, hipStream_t s)
{
    if (!v.empty()) (void)hipEventRecord(v.back().second, s);
}

extern "C" int tsqa_profile_enable(tsqa_ctx* c, int on)
{
    if (!c) return TSQA_ERR_ARG;
    c->profiling = on != 0;
    return TSQA_OK;
}

static void drain_events(std::vector<std::pair<hipEvent_t, hipEvent_t>>& v, double* ms, uint32_t* count)
{
    double sum = 0; uint32_t n = 0;
    for (auto& p : v) {
        float t = 0;
        if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) { sum += t; n++; }
        (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    v.clear();
    if (ms) *ms = sum;
    if (count) *count = n;
}

extern "C" int tsqa_profile_read(tsqa_ctx* c, double* enc_ms, uint32_t* enc_n, double* dec_ms, uint32_t* dec_n)
{
    if (!c) return TSQA_ERR_ARG;
    (void)hipSetDevice(c->device);
    drain_events(c->enc_events, enc_ms, enc_n);
    drain_events(c->dec_events, dec_ms, dec_n);
    return TSQA_OK;
}

// ---- internal launches (also used by the reference-API layer in tsq_compat.cpp) ----

int tsqa_ctx::launch_encode(const void* d_in, size_t n, size_t readable, uint32_t ext, int32_t* status, hipStream_t s)
{
    const uint32_t nb = (uint32_t)tsqa_block_count(n);
    const bool serial = enc_variant == 1;
    int rc = reserve(nb, serial);
    if (rc) return rc;
    const uint8_t* in = static_cast<const uint8_t*>(d_in);
    const bool timed = prof_begin(enc_events, s) != nullptr;
    if (serial) {
        if (ext) hipLaunchKernelGGL(enc_serial_kernel<true>, dim3(nb), dim3(64), 0, s, in, (uint64_t)n, (uint64_t)readable, slots, sizes, tables, status);
        else     hipLaunchKernelGGL(enc_serial_kernel<false>, dim3(nb), dim3(64), 0, s, in, (uint64_t)n, (uint64_t)readable, slots, sizes, tables, status);
    } else {
        rc = launch_encode_fast(this, in, n, readable, ext, status, s);
        if (rc) return rc;
    }
    if (timed) prof_end(enc_events, s);
    TSQ_HIP(this, hipGetLastError());
    return TSQA_OK;
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

int tsqa_ctx::launch_decode(const void* d_container, uint32_t n_blocks, void* d_out, int32_t* status, hipStream_t s)
{
    const uint8_t* in = static_cast<const uint8_t*>(d_container);
    const bool timed = prof_begin(dec_events, s) != nullptr;
    if (dec_variant == 1) {
        hipLaunchKernelGGL(dec_serial_kernel, dim3(n_blocks), dim3(64), 0, s, in, frames, static_cast<uint8_t*>(d_out), status);
    } else {
        int rc = launch_decode_fast(this, in, n_blocks, static_cast<uint8_t*>(d_out), status, s);
        if (rc) return rc;
    }
    if (timed) prof_end(dec_events, s);
    TSQ_HIP(this, hipGetLastError());
    return TSQA_OK;
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
    int rc = c->launch_encode(d_in, n, n, ext, d_status, s);
    if (rc) return rc;
    return c->launch_pack(n, ext, d_out, out_cap, d_out_size, d_status, s);
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

extern "C" int tsqa_decompress_device_async(tsqa_ctx* c, const void* d_in, size_t n, uint32_t n_blocks, void* d_out,
                                            size_t out_cap, uint64_t* d_out_size, int32_t* d_status, void* hip_stream)
{
    if (!c) return TSQA_ERR_ARG;
    if (!d_in || !d_out || !d_out_size || !d_status || n < 16 || n_blocks == 0) { c->set_error("decompress: bad argument"); return TSQA_ERR_ARG; }
    hipStream_t s = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->stream;
    (void)hipSetDevice(c->device);
    int rc = c->reserve(n_blocks, false);
    if (rc) return rc;
    TSQ_HIP(c, hipMemsetAsync(d_status, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(frame_walk_kernel, dim3(1), dim3(64), 0, s, static_cast<const uint8_t*>(d_in), (uint64_t)n, n_blocks,
                       (uint64_t)out_cap, c->frames, d_out_size, d_status);
    return c->launch_decode(d_in, n_blocks, d_out, d_status, s);
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
    *out_size = (size_t)sz;
    return status_to_rc(c, st, "decompress");
}

#ifdef TSQ_STATS
// instrumented builds only: counters published by block 0 of the last encode / decode launch
extern "C" int tsqa_debug_stats(unsigned long long* enc48, unsigned long long* dec16)
{
    if (enc48 && hipMemcpyFromSymbol(enc48, HIP_SYMBOL(tsq::g_enc_stats), 48 * sizeof(unsigned long long)) != hipSuccess) return TSQA_ERR_HIP;
    if (dec16 && hipMemcpyFromSymbol(dec16, HIP_SYMBOL(tsq::g_dec_stats), 16 * sizeof(unsigned long long)) != hipSuccess) return TSQA_ERR_HIP;
    return TSQA_OK;
}
extern "C" int tsqa_debug_syms(uint32_t* out8192)
{
    return hipMemcpyFromSymbol(out8192, HIP_SYMBOL(tsq::g_dbg_syms), 8192 * sizeof(uint32_t)) == hipSuccess ? TSQA_OK : TSQA_ERR_HIP;
}
#endif
