// tsq_emit.cuh -- wave-uniform output-stream bookkeeping shared by the encoder kernels.
//
// Restates the symbol accounting of tsq_encode.cpp:57-61,93-95,113-115,157-159,176-188 with the
// control and size bytes held in registers and written once when complete, instead of the
// reference's read-modify-write of output memory.  The reference's 16-byte literal stores spill
// past the literal; the only place the spill is observable is in the one or two control/size
// bytes that are allocated but never filled at the end of the stream.  We
// reproduce those bytes from the position of the last literal chunk instead of spilling.
#pragma once

#include "tsq_common.cuh"

namespace tsq {

struct Emitter {
    uint8_t* out;        // block slot
    uint32_t j;          // next free output byte
    uint32_t ctl_at, sz_at;
    uint32_t ctl_val, sz_val;
    uint32_t nsym;
    uint32_t origin;     // input position at the start of the current pair (rep_last_i)
    uint32_t lit_out;    // where the last literal chunk was stored
    uint32_t lit_src;    // and the input position it came from

    __device__ __forceinline__ void begin(uint8_t* slot, uint32_t n, bool writer)
    {
        out = slot;
        if (writer) { slot[0] = (uint8_t)n; slot[1] = (uint8_t)(n >> 8); slot[2] = (uint8_t)(n >> 16); }
        ctl_at = 3; sz_at = 4; j = 5; ctl_val = 0; sz_val = 0; nsym = 0; origin = 0;
        lit_out = 0xFFFFFFFFu; lit_src = 0;
    }

    // One symbol: its literal/match bit and its nibble; a new control byte every 8 symbols,
    // a new size byte every 2, control first.  `writer` = the lane that performs the stores.
    __device__ __forceinline__ void account(uint32_t is_literal, uint32_t nibble,
                                            uint32_t origin_if_pair_closes, bool writer)
    {
        nsym++;
        ctl_val = (ctl_val << 1) | is_literal;
        if ((nsym & 7u) == 0) {
            if (writer) out[ctl_at] = (uint8_t)ctl_val;
            ctl_at = j++; ctl_val = 0;
        }
        sz_val = (sz_val << 4) | nibble;
        if ((nsym & 1u) == 0) {
            if (writer) out[sz_at] = (uint8_t)sz_val;
            sz_at = j++; sz_val = 0; origin = origin_if_pair_closes;
        }
    }

    // Value a never-filled byte at output position p holds in the reference: the spill of the
    // last 16-byte literal store if it reaches p, else the zero the buffer was filled with.
    __device__ __forceinline__ uint32_t stale(uint32_t p, const uint8_t* src, uint64_t avail) const
    {
        uint32_t d = p - lit_out;
        return (lit_out != 0xFFFFFFFFu && d < 16u) ? ldu8z(src, (uint64_t)lit_src + d, avail) : 0u;
    }

    // Tail of tsq_encode.cpp:176-188.  Returns the stream size.
    __device__ __forceinline__ uint32_t finish(const uint8_t* src, uint64_t avail, bool writer)
    {
        uint32_t used = nsym & 7u;
        if (used == 0) {
            // a fresh control byte then a fresh size byte, both untouched
            if (writer) { out[ctl_at] = (uint8_t)stale(ctl_at, src, avail); out[sz_at] = (uint8_t)stale(sz_at, src, avail); }
        } else {
            uint32_t pad = 8u - used;
            if (writer) out[ctl_at] = (uint8_t)((ctl_val << pad) | ((1u << pad) - 1u));
            uint32_t s = (nsym & 1u) ? (sz_val << 4) : (stale(sz_at, src, avail) << 4);
            if (writer) out[sz_at] = (uint8_t)s;
        }
        return j;
    }
};

}  // namespace tsq
