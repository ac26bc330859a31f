// tsq_common.cuh -- device-side constants and helpers shared by all gfx950 kernels.
// (HIP source; the .cuh suffix only marks "device header", nothing here is CUDA.)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsq {

constexpr uint32_t kBlockBits   = 22;                       // turbosqueeze.h:37
constexpr uint32_t kBlockSize   = 1u << kBlockBits;         // 4 MiB
constexpr uint32_t kSlotSize    = kBlockSize + (kBlockSize >> 2);   // TSQ_OUTPUT_SZ, turbosqueeze.h:39
constexpr uint32_t kHashBits    = 17;                       // turbosqueeze.h:41
constexpr uint32_t kHashEntries = 1u << kHashBits;
constexpr uint32_t kHashMask    = kHashEntries - 1u;
constexpr uint32_t kWave        = 64;

// status codes mirrored from include/turbosqueeze_amd.h
constexpr int32_t kOk = 0, kErrFormat = 4, kErrStream = 5, kErrOverflow = 6, kErrStall = 7;

// Per-block frame description produced by the frame-walk kernel for the decoders.
struct FrameInfo {
    uint64_t stream_at;   // byte offset of the block stream inside the container
    uint64_t out_at;      // byte offset of the block in the decompressed output
    uint32_t stream_len;  // compressed bytes (frame & 0x7FFFFF)
    uint32_t ext;         // frame bit 23
    uint32_t out_len;     // u24 header of the stream
    uint32_t pad;
};

// ---- unaligned little-endian loads (gfx950 global memory handles them in hardware) ----
__device__ __forceinline__ uint32_t ldu16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
__device__ __forceinline__ uint32_t ldu32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ldu64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

// Loads that see zeros past `avail` bytes (the canonical zero halo after the last block).
__device__ __forceinline__ uint32_t ldu32z(const uint8_t* base, uint64_t at, uint64_t avail)
{
    if (at + 4 <= avail) return ldu32(base + at);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4; ++k) if (at + k < avail) v |= (uint32_t)base[at + k] << (8 * k);
    return v;
}
__device__ __forceinline__ uint64_t ldu64z(const uint8_t* base, uint64_t at, uint64_t avail)
{
    if (at + 8 <= avail) return ldu64(base + at);
    uint64_t v = 0;
    for (uint32_t k = 0; k < 8; ++k) if (at + k < avail) v |= (uint64_t)base[at + k] << (8 * k);
    return v;
}
__device__ __forceinline__ uint32_t ldu8z(const uint8_t* base, uint64_t at, uint64_t avail)
{
    return at < avail ? base[at] : 0u;
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// hash of the 4 bytes at a position (tsq_encode.cpp:75)
__device__ __forceinline__ uint32_t hash4(uint32_t w) { return (w ^ (w >> 12)) & kHashMask; }

// candidate = the position congruent to `lo` (mod 65536) in [i-65536, i-1] (tsq_encode.cpp:76-78)
__device__ __forceinline__ uint32_t candidate_of(uint32_t lo, uint32_t i)
{
    uint32_t pos = (i & 0xFFFF0000u) + lo;
    return lo >= (i & 0xFFFFu) ? pos - 65536u : pos;
}

__device__ __forceinline__ bool offset_ok(uint32_t offset) { return (offset - 4u) < 0xFFFBu; }   // tsq_encode.cpp:100

// match length -> size nibble (tsq_encode.cpp:44-45), and bytes a nibble spans (tsq_encode.cpp:154)
__device__ __forceinline__ uint32_t length_nibble(uint32_t k)
{
    return k >= 64 ? 2u : k >= 48 ? 1u : k >= 32 ? 0u : k >= 17 ? 15u : k - 1u;
}
__device__ __forceinline__ uint32_t nibble_span(uint32_t m) { return m < 3u ? (m + 2u) << 4 : m + 1u; }

// common prefix in bytes from two 8-byte words; 8 when equal (platform.h:30-38 returns 64 for 0)
__device__ __forceinline__ uint32_t prefix8(uint64_t a, uint64_t b)
{
    uint64_t x = a ^ b;
    return x ? (uint32_t)__builtin_ctzll(x) >> 3 : 8u;
}

// Inclusive scans over the 64 lanes of a wavefront with DPP row shifts and row broadcasts: six VALU instructions, no LDS round
// trips (a __shfl_up is a ds_bpermute: an LDS-pipe operation with its latency, and the LDS pipe is what this kernel is short of).
#define TSQ_DPP_SCAN_STEP(op, ctrl, rows)                                                                                        \
    { const uint32_t t_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false); v = op(v, t_); }
__device__ __forceinline__ uint32_t dpp_add(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t dpp_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v)
{
    TSQ_DPP_SCAN_STEP(dpp_add, 0x111, 0xF) TSQ_DPP_SCAN_STEP(dpp_add, 0x112, 0xF) TSQ_DPP_SCAN_STEP(dpp_add, 0x114, 0xF) TSQ_DPP_SCAN_STEP(dpp_add, 0x118, 0xF)
    TSQ_DPP_SCAN_STEP(dpp_add, 0x142, 0xA) TSQ_DPP_SCAN_STEP(dpp_add, 0x143, 0xC)      // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ uint32_t dpp_last(uint32_t own, uint32_t earlier) { return own ? own : earlier; }
__device__ __forceinline__ uint32_t wave_scan_last(uint32_t v)                         // every lane: the last non-zero value at or before it (0: none)
{
    TSQ_DPP_SCAN_STEP(dpp_last, 0x111, 0xF) TSQ_DPP_SCAN_STEP(dpp_last, 0x112, 0xF) TSQ_DPP_SCAN_STEP(dpp_last, 0x114, 0xF) TSQ_DPP_SCAN_STEP(dpp_last, 0x118, 0xF)
    TSQ_DPP_SCAN_STEP(dpp_last, 0x142, 0xA) TSQ_DPP_SCAN_STEP(dpp_last, 0x143, 0xC)
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v)                          // values >= 0; 0 is the identity
{
    TSQ_DPP_SCAN_STEP(dpp_max, 0x111, 0xF) TSQ_DPP_SCAN_STEP(dpp_max, 0x112, 0xF) TSQ_DPP_SCAN_STEP(dpp_max, 0x114, 0xF) TSQ_DPP_SCAN_STEP(dpp_max, 0x118, 0xF)
    TSQ_DPP_SCAN_STEP(dpp_max, 0x142, 0xA) TSQ_DPP_SCAN_STEP(dpp_max, 0x143, 0xC)
    return v;
}

}  // namespace tsq
