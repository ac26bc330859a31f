// tsq_internal.h -- host-side context shared by the runtime and the reference-API layer.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <utility>
#include <vector>

#include "../../include/turbosqueeze_amd.h"

namespace tsq { struct FrameInfo; }

struct tsqa_ctx {
    int device = 0;
    int n_cus = 0;
    hipStream_t stream = nullptr;
    // scratch in HBM
    uint8_t* slots = nullptr;          // n_blocks x TSQ_OUTPUT_SZ encoded block streams
    uint32_t* sizes = nullptr;         // n_blocks stream sizes
    uint64_t* frame_at = nullptr;      // n_blocks + 1 frame offsets in the container
    tsq::FrameInfo* frames = nullptr;  // n_blocks frame descriptors (decode)
    uint16_t* tables = nullptr;        // n_blocks x 2^17 u16 position tables of the encoders
    size_t cap_blocks = 0, cap_tables = 0, cap_slots = 0;
    tsq::FrameInfo* host_frames = nullptr;     // frame descriptors built on the host (sharded fetch + decode): pinned
    size_t cap_host_frames = 0;
    std::vector<uint64_t> host_frame_src;      // where each owned frame's stream starts in the host container
    hipEvent_t host_frames_copied = nullptr;   // recorded behind the descriptors' copy to the device
    bool host_frames_pending = false;
    // what the last tsqa_sharded_fetch_decode_async left on the device for tsqa_sharded_decode_again_async: the descriptor count and
    // the buffers they refer to.  Zero / null whenever `frames` may hold anything else (every other writer of `frames`, a reallocation).
    uint32_t sharded_n_local = 0;
    const void* sharded_streams = nullptr;
    void* sharded_out = nullptr;
    void forget_sharded() { sharded_n_local = 0; sharded_streams = nullptr; sharded_out = nullptr; }
    char probe_shape[160] = {0};               // what tsqa_measure_copy chose (tsqa_copy_probe_shape)
    uint32_t* duo_ring = nullptr;      // two-workgroup decoder: chunk records handed from the PARSE to the COPY workgroup of a block
    uint32_t* duo_flags = nullptr;     // and their progress counters
    size_t cap_duo = 0;
    uint64_t* d_size = nullptr;        // result words of the synchronous entry points
    int32_t* d_status = nullptr;
    int enc_variant = 0, dec_variant = 0;
    uint32_t decode_wait_limit = 1u << 24;   // polls before a multi-workgroup decode gives up on a sibling workgroup (TSQA_ERR_STALL)
    char err[256] = {0};
    // optional timing (tsqa_profile_*): HIP event pairs on the launch stream, taken from a pool made when profiling is
    // switched on (nothing is created or destroyed between the events).  Kinds: 0 encode kernel, 1 decode kernel,
    // 2 whole compress call (encode + container pack), 3 whole decompress call (frame walk + decode).
    static constexpr int kProfKinds = 4, kProfPairs = 256;
    bool profiling = false;
    std::vector<hipEvent_t> prof_pool;                     // kProfKinds * kProfPairs * 2 events
    uint32_t prof_used[kProfKinds] = {0, 0, 0, 0};
    bool prof_begin(int kind, hipStream_t s);
    void prof_end(int kind, hipStream_t s);

    void set_error(const char* fmt, ...) __attribute__((format(printf, 2, 3)));
    int reserve(size_t n_blocks, bool want_tables, bool want_slots = true);
    int reserve_duo(size_t n_blocks);
    int reserve_host_frames(size_t n);
    // `readable` >= n: bytes of d_in that may be read (look-ahead halo); zeros are seen beyond it
    int launch_encode(const void* d_in, size_t n, size_t readable, uint32_t ext, int32_t* status, hipStream_t s);
    // general form: block b at d_in + b * stride, streams to slots_out[b * TSQ_OUTPUT_SZ], sizes to sizes_out[b]
    int launch_encode_to(const void* d_in, size_t n, size_t readable, size_t stride, uint32_t ext, uint8_t* slots_out,
                         uint32_t* sizes_out, int32_t* status, hipStream_t s);
    // (variant < 0: the context's decode variant; the retry after TSQA_ERR_STALL passes 4 -- one workgroup per block -- without touching
    //  the context's setting, which other threads of the scheduler read)
    int launch_decode_frames(const void* d_streams, const tsq::FrameInfo* d_frames, uint32_t n_blocks, void* d_out, int32_t* status, hipStream_t s, int variant = -1);
    int launch_pack(size_t n, uint32_t ext, void* d_out, size_t out_cap, uint64_t* d_out_size, int32_t* status, hipStream_t s);
    int launch_decode(const void* d_container, uint32_t n_blocks, void* d_out, int32_t* status, hipStream_t s, int variant = -1);
};
