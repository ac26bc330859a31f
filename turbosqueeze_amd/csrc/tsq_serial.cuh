// tsq_serial.cuh -- correctness-baseline kernels: one wavefront per block, the parse walked
// serially (wave-uniform control flow), lanes only used to move bytes.  Kept as kernel
// variant 1 (tsqa_set_kernel_variant) for A/B checks of the fast kernels on the GPU itself.
#pragma once

#include "tsq_common.cuh"
#include "tsq_emit.cuh"

namespace tsq {

// ---------------------------------------------------------------------------------------------
// Encoder (tsq_encode.cpp:48-189 no-ext, :192-342 ext), hash table in HBM (256 KiB per block).
// ---------------------------------------------------------------------------------------------
template <bool EXT>
__global__ __launch_bounds__(64) void enc_serial_kernel(const uint8_t* __restrict__ in, uint64_t n_total, uint64_t readable, uint64_t stride,
                                                        uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                        uint16_t* __restrict__ tables, int32_t* __restrict__ status)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint64_t start = (uint64_t)b * stride;  // (see enc_stage_kernel: contiguous blocks or a shard's blocks with their look-ahead)
    const uint64_t avail = readable - start;      // bytes that may be read from src; zeros beyond
    const uint64_t vstart = (uint64_t)b << kBlockBits;
    const uint32_t n = n_total - vstart < kBlockSize ? (uint32_t)(n_total - vstart) : kBlockSize;
    const uint8_t* src = in + start;
    uint8_t* out = slots + (size_t)b * kSlotSize;
    uint16_t* table = tables + (size_t)b * kHashEntries;
    const bool writer = lane == 0;
    constexpr uint32_t kCap = EXT ? 64u : 16u;

    // tsqInit (tsq_context.cpp:77-80)
    {
        uint4* t4 = reinterpret_cast<uint4*>(table);
        for (uint32_t k = lane; k < kHashEntries * 2 / 16; k += kWave) t4[k] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }

    Emitter e;
    e.begin(out, n, writer);
    uint32_t i = 0, pending, pos, word, offset;

    auto probe = [&](uint32_t at) {
        word = uniform(ldu32z(src, at, avail));
        uint32_t h = hash4(word);
        uint32_t lo = uniform((uint32_t)table[h]);
        if (writer) table[h] = (uint16_t)at;
        pos = candidate_of(lo, at);
    };
    auto literals = [&](uint32_t from, uint32_t to) {
        while (to != from) {
            uint32_t len = to - from > 16u ? 16u : to - from;
            if (lane < len) out[e.j + lane] = src[from + lane];     // from+lane < n here
            e.lit_out = e.j; e.lit_src = from;
            from += len; e.j += len;
            e.account(1u, len - 1u, from, writer);
        }
        return from;
    };

    do {
        pending = i;
        do {                                                   // tsq_encode.cpp:70-100
            i++;
            probe(i);
            offset = e.origin - pos;                           // before the forced flush
            if (i - pending > 31u) pending = literals(pending, i);
        } while (i < n && !(word == uniform(ldu32z(src, pos, avail)) && offset_ok(offset)));

        pending = literals(pending, i);                        // tsq_encode.cpp:103-118
        if (!(i < n)) break;

        do {                                                   // tsq_encode.cpp:123-170
            uint32_t k = prefix8(ldu64z(src, i, avail), ldu64z(src, pos, avail));
            if (k == 8) {
                uint32_t nb, a = i, c = pos;
                do { a += 8; c += 8; nb = prefix8(ldu64z(src, a, avail), ldu64z(src, c, avail)); k += nb; }
                while (nb == 8 && k < kCap);
            }
            k = uniform(k);
            uint32_t room = e.origin - pos;
            if (k > room) k = room - 1u;
            if (k < 4u) break;
            offset = e.origin - pos;
            if (!offset_ok(offset)) break;
            uint32_t m = length_nibble(k);
            if (writer) { out[e.j] = (uint8_t)offset; out[e.j + 1] = (uint8_t)(offset >> 8); }
            e.j += 2;
            i += nibble_span(m);
            e.account(0u, m, i, writer);
            probe(i);
            offset = e.origin - pos;
        } while (i < n - 5u && word == uniform(ldu32z(src, pos, avail)) && offset_ok(offset));

        if (e.j + 80u > kSlotSize) { if (writer) atomicMax(status, kErrOverflow); sizes[b] = 3; return; }
    } while (i < n);

    uint32_t total = e.finish(src, avail, writer);
    if (writer) sizes[b] = total;
}

// ---------------------------------------------------------------------------------------------
// Decoder (tsq_decode.cpp:42-126 no-ext, :129-315 ext).  Output clamped at the header size;
// offsets and stream bounds are checked (the reference checks neither).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dec_serial_kernel(const uint8_t* __restrict__ container,
                                                        const FrameInfo* __restrict__ frames,
                                                        uint8_t* __restrict__ outbuf,
                                                        int32_t* __restrict__ status)
{
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (uniform((uint32_t)*status) != 0u) return;           // (one wavefront: the first lane's reading decides for all)
    const FrameInfo f = frames[b];
    if (f.stream_len < 3u || f.stream_len > kSlotSize || f.out_len > kBlockSize) {       // untrusted descriptors: see dec_sym_kernel
        if (lane == 0) atomicMax(status, kErrStream);
        return;
    }
    const uint8_t* in = container + f.stream_at;
    uint8_t* out = outbuf + f.out_at;
    const uint32_t in_len = f.stream_len, size = f.out_len, ext = f.ext;
    uint32_t i = 3, j = 0;
    int32_t bad = 0;

    while (j < size && !bad) {
        if (i >= in_len) { bad = kErrStream; break; }
        uint32_t control = uniform((uint32_t)in[i++]);
        for (uint32_t p = 0; p < 4 && j < size && !bad; ++p) {
            if (i >= in_len) { bad = kErrStream; break; }
            uint32_t sizes = uniform((uint32_t)in[i++]);
            uint32_t origin = j;
            for (uint32_t s = 0; s < 2 && j < size; ++s) {
                uint32_t nib = s == 0 ? sizes >> 4 : sizes & 15u;
                uint32_t is_lit = (control >> (7u - (2u * p + s))) & 1u;
                uint32_t len, take;
                if (is_lit) {
                    len = nib + 1u;
                    take = len < size - j ? len : size - j;
                    if (i + take > in_len) { bad = kErrStream; break; }
                    if (lane < take) out[j + lane] = in[i + lane];
                    i += len;
                } else {
                    if (i + 2u > in_len) { bad = kErrStream; break; }
                    uint32_t off = uniform(ldu16(in + i));
                    i += 2;
                    len = (ext && nib < 3u) ? (nib + 2u) << 4 : nib + 1u;
                    if (off > origin) { bad = kErrStream; break; }
                    uint32_t from = origin - off;
                    take = len < size - j ? len : size - j;
                    if (from + take > origin) { bad = kErrStream; break; }
                    if (lane < take) out[j + lane] = out[from + lane];
                }
                j += take;
            }
        }
    }
    if (bad && lane == 0) atomicMax(status, bad);
}

}  // namespace tsq
