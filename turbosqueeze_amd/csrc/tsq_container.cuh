// tsq_container.cuh -- device-side .tsq container assembly and frame walk.
//
// Container (turbosqueeze.cpp:64-83, tsq_threads.cpp:218-239,333-335): 16-byte header
// "TSQ1" | u32 n_blocks | u64 total, then per block a u24 frame (size | ext << 23) and the stream.
#pragma once

#include "tsq_common.cuh"

namespace tsq {

// One workgroup: exclusive scan of (3 + size_b), header + frame bytes, total size.
// Replaces compression_write_worker's serial frame emission (tsq_threads.cpp:192-275).
__global__ __launch_bounds__(256) void pack_scan_kernel(const uint32_t* __restrict__ sizes, uint32_t n_blocks,
                                                        uint64_t n_total, uint32_t ext, uint8_t* __restrict__ container,
                                                        uint64_t out_cap, uint64_t* __restrict__ frame_at,
                                                        uint64_t* __restrict__ out_size, int32_t* __restrict__ status)
{
    __shared__ uint64_t wave_sum[4];
    __shared__ uint64_t carry;
    const uint32_t t = threadIdx.x, lane = t & 63u, wid = t >> 6;
    if (t == 0) carry = 16;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        uint32_t b = base + t;
        uint64_t v = b < n_blocks ? 3ull + sizes[b] : 0ull;
        uint64_t incl = v;                                   // inclusive wave scan
        for (uint32_t d = 1; d < 64; d <<= 1) {
            uint64_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_sum[wid] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wid; ++w) before += wave_sum[w];
        if (b < n_blocks) frame_at[b] = before + incl - v;
        __syncthreads();
        if (t == 255) carry = before + incl;
        __syncthreads();
    }
    const uint64_t total = carry;
    if (t == 0) {
        frame_at[n_blocks] = total;
        *out_size = total;
        if (total > out_cap) atomicMax(status, kErrOverflow);
    }
    if (total > out_cap) return;
    if (t < 16) {
        uint8_t v;
        if (t < 4) v = (uint8_t)"TSQ1"[t];
        else if (t < 8) v = (uint8_t)(n_blocks >> (8 * (t - 4)));
        else v = (uint8_t)(n_total >> (8 * (t - 8)));
        container[t] = v;
    }
    for (uint32_t b = t; b < n_blocks; b += 256) {
        uint32_t frame = sizes[b] | (ext ? 0x800000u : 0u);     // tsq_threads.cpp:218-219
        uint8_t* p = container + frame_at[b];
        p[0] = (uint8_t)frame; p[1] = (uint8_t)(frame >> 8); p[2] = (uint8_t)(frame >> 16);
    }
}

// Copy each block stream from its slot to its place in the container.  Destination offsets are
// arbitrary bytes, so each thread stores one destination-aligned 16-byte word assembled from an
// unaligned 16-byte load; head and tail bytes go singly.  grid = (pieces, n_blocks).
constexpr uint32_t kPackPiece = 32768;
__global__ __launch_bounds__(256) void pack_copy_kernel(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ sizes,
                                                        const uint64_t* __restrict__ frame_at, uint8_t* __restrict__ container,
                                                        const int32_t* __restrict__ status)
{
    const uint32_t b = blockIdx.y;
    const uint32_t size = sizes[b];
    const uint32_t piece_at = blockIdx.x * kPackPiece;
    if (piece_at >= size || *status != 0) return;
    const uint8_t* src = slots + (size_t)b * kSlotSize;
    uint8_t* dst = container + frame_at[b] + 3;
    // Piece p moves the destination-aligned words [head + p*P, head + (p+1)*P); piece 0 also
    // moves the `head` bytes in front of the first aligned word.
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    const uint32_t begin = piece_at + head;
    const uint32_t end_all = size;
    uint32_t end = begin + kPackPiece;
    if (end > end_all) end = end_all;
    if (piece_at == 0) for (uint32_t k = threadIdx.x; k < head && k < size; k += 256) dst[k] = src[k];
    if (begin >= end_all) return;
    const uint32_t words = (end - begin) >> 4;
    for (uint32_t w = threadIdx.x; w < words; w += 256) {
        uint4 v;
        __builtin_memcpy(&v, src + begin + (w << 4), 16);
        *reinterpret_cast<uint4*>(dst + begin + (w << 4)) = v;
    }
    // the last piece that reaches the end carries the tail bytes
    if (end == end_all) for (uint32_t k = begin + (words << 4) + threadIdx.x; k < end_all; k += 256) dst[k] = src[k];
}

// Serial frame walk of a container (tsq_threads.cpp:444-543: block k starts at 16 + sum(3+size_j)).
// One wavefront; lane 0 walks, since each frame position depends on the previous one.
__global__ __launch_bounds__(64) void frame_walk_kernel(const uint8_t* __restrict__ container, uint64_t n, uint32_t n_blocks,
                                                        uint64_t out_cap, FrameInfo* __restrict__ frames,
                                                        uint64_t* __restrict__ out_size, int32_t* __restrict__ status)
{
    if (threadIdx.x != 0) return;
    int32_t bad = 0;
    uint64_t total = 0, at = 16, oat = 0;
    if (n < 16 || container[0] != 'T' || container[1] != 'S' || container[2] != 'Q' || container[3] != '1') bad = kErrFormat;
    if (!bad) {
        uint32_t nb = ldu32(container + 4);
        total = ldu64(container + 8);
        if (nb != n_blocks || nb == 0) bad = kErrFormat;          // tsq_threads.cpp:759-768
        if (total > out_cap) bad = kErrFormat;
    }
    for (uint32_t b = 0; b < n_blocks && !bad; ++b) {
        if (at + 6 > n) { bad = kErrFormat; break; }
        uint32_t frame = (uint32_t)container[at] | ((uint32_t)container[at + 1] << 8) | ((uint32_t)container[at + 2] << 16);
        uint32_t len = frame & 0x7FFFFFu;                          // tsq_threads.cpp:513-517
        if (len < 3 || len > kSlotSize || at + 3 + len > n) { bad = kErrFormat; break; }
        uint32_t usize = (uint32_t)container[at + 3] | ((uint32_t)container[at + 4] << 8) | ((uint32_t)container[at + 5] << 16);
        if (usize > kBlockSize || oat + usize > total) { bad = kErrFormat; break; }
        FrameInfo f;
        f.stream_at = at + 3; f.out_at = oat; f.stream_len = len; f.ext = frame >> 23; f.out_len = usize; f.pad = 0;
        frames[b] = f;
        oat += usize;
        at += 3 + len;
    }
    if (!bad && oat != total) bad = kErrFormat;
    *out_size = bad ? 0 : total;
    if (bad) atomicMax(status, bad);
}

}  // namespace tsq
