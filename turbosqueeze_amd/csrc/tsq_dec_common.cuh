// tsq_dec_common.cuh -- what the block decoders share: the symbol record and the instrumentation macros.
#pragma once

#include "tsq_common.cuh"

namespace tsq {

struct DecSym {            // 8 bytes
    uint16_t out_rel;      // position inside the chunk image
    uint8_t len;           // bytes to produce (already clamped at the block size)
    uint8_t kind;          // 0 none, 1 literal, 2 match
    uint32_t a;            // literal: chunk-relative stream offset; match: block-absolute source position
};

#ifdef TSQ_STATS
__device__ unsigned long long g_dec_stats[16];
__device__ unsigned long long g_dec_wave[48];      // per wavefront of block 0: cycles in the pointer jumping, waiting bytes, loop iterations
#define TSQD_T0() unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define TSQD_ACC(slot) do { unsigned long long t1_ = __builtin_amdgcn_s_memtime(); st_[slot] += t1_ - t0_; t0_ = t1_; } while (0)
#define TSQD_CNT(slot, v) st_[slot] += (v)
#else
#define TSQD_T0() do {} while (0)
#define TSQD_ACC(slot) do {} while (0)
#define TSQD_CNT(slot, v) do {} while (0)
#endif

}  // namespace tsq
