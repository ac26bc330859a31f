// tsq_enc_util.cuh -- small device helpers shared by the block encoders: instrumentation macros, zero-padded
// 16-byte loads, common-prefix length, lane/mask arithmetic, symbol record encodings and the LDS pointer types.
#pragma once

#include "tsq_common.cuh"

namespace tsq {

// Instrumented builds (-DTSQ_STATS, make stats): block 0 publishes cycle and event counters.
// (-DTSQ_TRACEONLY: the timeline alone, at production timing -- tools/tile_trace.py)
#if defined(TSQ_STATS) || defined(TSQ_TRACEONLY)
// a timeline: when each stage handed over each of 256 tiles (block 0, tiles kTraceFrom ..), low 32 bits of s_memtime
__device__ uint32_t g_enc_trace[16 * 256];
constexpr uint32_t kTraceFrom = 20000;
#define TSQ_TRACE(stage, t) do { if (blockIdx.x == 0 && (t) - kTraceFrom < 256u) { if ((threadIdx.x & 63u) == 0) g_enc_trace[(stage) * 256 + ((t) - kTraceFrom)] = (uint32_t)__builtin_amdgcn_s_memtime(); } } while (0)
// (the same stored by every lane: for the scalar stages, where a one-lane branch makes the compiler move uniform values to vector registers)
#define TSQ_TRACE_ALL(stage, t) do { if (blockIdx.x == 0 && (t) - kTraceFrom < 256u) g_enc_trace[(stage) * 256 + ((t) - kTraceFrom)] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#endif
#ifdef TSQ_STATS
__device__ unsigned long long g_enc_stats[64];
#define TSQ_T0() unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define TSQ_ACC(slot) do { unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st_[slot] += t1_ - t0_; t0_ = t1_; } while (0)
#define TSQ_CNT(slot, v) st_[slot] += (v)
#define TSQ_SUB(slot) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); TSQ_ACC(slot); } while (0)
#else
#ifdef TSQ_MARKS
#define TSQ_ACC(slot) asm volatile("; TSQ_MARK " #slot ::: "memory")
#else
#define TSQ_ACC(slot) do {} while (0)
#endif
#define TSQ_T0() do {} while (0)
#ifndef TSQ_TRACEONLY
#define TSQ_TRACE(stage, t) do {} while (0)
#define TSQ_TRACE_ALL(stage, t) do {} while (0)
#endif
#define TSQ_CNT(slot, v) do {} while (0)
#define TSQ_SUB(slot) TSQ_ACC(slot)
#endif

// Light instrumentation (-DTSQ_SPINS, make spins, tools/spin_counts.py): every unsuccessful poll of every wavefront is counted in
// LDS (one ds_add per spin, nothing on the paths that do not wait), so that who waits for whom shows at production timing.
#ifdef TSQ_SPINS
__device__ uint32_t g_enc_spins[20];
#define TSQ_SPIN(ctl) do { if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(&(ctl)[48u + (threadIdx.x >> 6)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#define TSQ_SPIN_AT(ctl, slot) do { if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(&(ctl)[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } while (0)
#else
#define TSQ_SPIN(ctl) do {} while (0)
#define TSQ_SPIN_AT(ctl, slot) do {} while (0)
#endif

// Bottleneck finder (-DTSQ_X_DELAY_STAGE=k, tools/xbuild.sh): stage k sleeps 128 cycles per tile (per item / per batch for BUILDER / EMIT); the
// stage whose delay shows one to one in the kernel time paces the pipeline.
#ifdef TSQ_X_DELAY_STAGE
#define TSQ_DELAY(k) do { if ((k) == TSQ_X_DELAY_STAGE) __builtin_amdgcn_s_sleep(2); } while (0)
#else
#define TSQ_DELAY(k) do {} while (0)
#endif

// 16 bytes at src+at, zeros past `avail`
__device__ __forceinline__ uint4 ld128z(const uint8_t* src, uint64_t at, uint64_t avail)
{
    uint4 v;
    if (__builtin_expect(at + 16 <= avail, 1)) { __builtin_memcpy(&v, src + at, 16); return v; }
    uint32_t w[4] = {0, 0, 0, 0};
#pragma nounroll
    for (uint32_t k = 0; k < 16; ++k) if (at + k < avail) w[k >> 2] |= (uint32_t)src[at + k] << (8u * (k & 3u));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t prefix16(uint4 a, uint4 b)
{
    uint32_t k = prefix8((uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)b.x | ((uint64_t)b.y << 32));
    if (k == 8) k += prefix8((uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)b.z | ((uint64_t)b.w << 32));
    return k;
}

__device__ __forceinline__ uint64_t below(uint32_t bit) { return bit >= 64u ? ~0ull : (1ull << bit) - 1ull; }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// Symbol records (one u32 per symbol):
//   literal: bit 31 = 1, bits 22..25 = length-1, bits 0..21 = input position of the first byte
//   match:   bit 31 = 0, bits 16..19 = size nibble, bits 0..15 = offset
__device__ __forceinline__ uint32_t rec_literal(uint32_t pos, uint32_t len) { return 0x80000000u | ((len - 1u) << 22) | pos; }
__device__ __forceinline__ uint32_t rec_match(uint32_t offset, uint32_t nib) { return (nib << 16) | offset; }

#ifdef TSQ_STATS
__device__ uint32_t g_dbg_syms[8192];
#endif
__device__ __forceinline__ uint32_t msb64(uint64_t m) { return 63u - (uint32_t)__builtin_clzll(m); }
__device__ __forceinline__ uint32_t lsb64(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
// number of consecutive set bits of `mask` starting at bit `from` (a run that reaches bit 63 included)
__device__ __forceinline__ uint32_t ones_from(uint64_t mask, uint32_t from)
{
    const uint64_t inv = ~(mask >> from);            // zero only when from == 0 and every bit is set
    return inv ? (uint32_t)__builtin_ctzll(inv) : 64u - from;
}

// LDS pointers with an explicit address space: volatile accesses through a generic pointer are
// not rewritten by address-space inference and would compile to flat_* plus a vmcnt(0) wait each.
typedef __attribute__((address_space(3))) uint8_t lds_u8_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;

// Items the parser hands the stream builder through the LDS queue.
// header words: 0 kind, 1 base, 2 V lo, 3 V hi, 4 nsym at entry, 5 origin at entry, 6 lit_from at entry,
// 7 certain lo, 8 certain hi, 9 record (kItemSym)        lane words: cand0 | nibble << 24
enum : uint32_t { kItemSeg = 1, kItemSym = 2, kItemEnd = 3 };

}  // namespace tsq
