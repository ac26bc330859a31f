// tsq_enc_builder.cuh -- the stream BUILDER wave of the staged encoder and its batch layout.
//
// The parser hands over, through a single-producer/single-consumer queue in LDS, one item per hazard-free
// segment of a tile (visited mask + which visited lanes are matches + per-lane candidate|nibble words + the parse
// state at segment entry) or per explicitly pushed symbol.  The builder derives the symbols in parallel -- matches =
// visited match lanes, literals = 16-byte chunks of visited literal runs, symbol indices by popcount, pair origin of
// an odd symbol = start of the previous one -- and every 64 symbols lays the stream out (emit_batch): control bytes
// from a ballot, size bytes from a shuffle, payload offsets from a prefix sum (tsq_encode.cpp:57-59,94-95,103-118,
// 152-159).  The never-filled trailing control/size bytes get the reference's stale values (tsq_encode.cpp:176-188).
#pragma once

#include "tsq_common.cuh"
#include "tsq_enc_util.cuh"

namespace tsq {

// Lay out and store the `cnt` (<= 64) symbols held one per lane in `rec`, starting at output
// position j0 which is the start of a group of 8 (tsq_encode.cpp:57-59,94-95: control byte, then per
// pair a size byte and the two payloads).  Returns the output position after the last payload.
// lit_out / lit_src report the last literal chunk of the batch (for the never-filled trailing
// bytes); lit_out == 0xFFFFFFFF when the batch holds no literal.  Not inlined: it runs once per
// 64 symbols and must not bloat the walk loop; the caller re-uniforms the results.
struct EmitResult { uint32_t end, lit_out, lit_src; };
#ifdef TSQ_OLD_EMIT
__device__ __noinline__ EmitResult emit_batch(uint32_t rec, uint32_t cnt, uint32_t j0, uint8_t* out, const uint8_t* src,
                                              uint64_t avail, uint32_t lane)
{
    uint32_t lit_out = 0xFFFFFFFFu, lit_src = 0;
    const bool live = lane < cnt;
    const uint32_t lit = live ? rec >> 31 : 1u;                   // padding symbols count as literals (tsq_encode.cpp:180)
    const uint32_t nib = live ? (lit ? (rec >> 22) & 15u : (rec >> 16) & 15u) : 0u;
    const uint32_t pay = live ? (lit ? nib + 1u : 2u) : 0u;
    const uint32_t extra = live ? (uint32_t)((lane & 7u) == 0u) + (uint32_t)((lane & 1u) == 0u) : 0u;
    uint32_t incl = pay + extra;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t up = __shfl_up(incl, d); if (lane >= d) incl += up; }
    const uint32_t at = j0 + incl - (pay + extra);                // where this symbol's control/size/payload region starts
    const uint32_t end = j0 + rdlane(incl, 63);

    const uint64_t lits = __ballot(lit != 0u);
    if (live && (lane & 7u) == 0u) {                               // control byte: first symbol of the group in bit 7
        uint32_t bits = (uint32_t)(lits >> lane) & 0xFFu;
        bits = __builtin_bitreverse32(bits) >> 24;
        out[at] = (uint8_t)bits;
    }
    const uint32_t nib_next = __shfl_down(nib, 1);
    if (live && (lane & 1u) == 0u) out[at + (uint32_t)((lane & 7u) == 0u)] = (uint8_t)((nib << 4) | nib_next);
    const uint32_t pay_at = at + extra;
    if (live) {
        if (lit) {
            const uint32_t pos = rec & 0x3FFFFFu;
            const uint4 v = ld128z(src, pos, avail);
            const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t t = 0; t < 16; ++t)
                if (t <= nib) out[pay_at + t] = (uint8_t)(wds[t >> 2] >> (8u * (t & 3u)));
        } else {
            const uint16_t off = (uint16_t)rec;
            __builtin_memcpy(out + pay_at, &off, 2);
        }
    }
    const uint64_t live_lits = lits & below(cnt);
    if (live_lits) {
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(live_lits);
        lit_out = rdlane(pay_at, last);
        lit_src = rdlane(rec, last) & 0x3FFFFFu;
    }
    return EmitResult{end, lit_out, lit_src};
}

#else
__device__ __noinline__ EmitResult emit_batch(uint32_t rec, uint32_t cnt, uint32_t j0, uint8_t* out, const uint8_t* src,
                                              uint64_t avail, uint32_t lane)
{
    uint32_t lit_out = 0xFFFFFFFFu, lit_src = 0;
    const bool live = lane < cnt;
    const uint32_t lit = live ? rec >> 31 : 1u;                   // padding symbols count as literals (tsq_encode.cpp:180)
    const uint32_t nib = live ? (lit ? (rec >> 22) & 15u : (rec >> 16) & 15u) : 0u;
    const uint32_t pay = live ? (lit ? nib + 1u : 2u) : 0u;
    const uint32_t extra = live ? (uint32_t)((lane & 7u) == 0u) + (uint32_t)((lane & 1u) == 0u) : 0u;
    // the literal bytes are on their way while the layout is computed
    uint4 v = make_uint4(0, 0, 0, 0);
    if (live && lit) v = ld128z(src, rec & 0x3FFFFFu, avail);
    const uint32_t incl = wave_scan_add(pay + extra);
    const uint32_t at = j0 + incl - (pay + extra);                // where this symbol's control/size/payload region starts
    const uint32_t end = j0 + rdlane(incl, 63);

    const uint64_t lits = __ballot(lit != 0u);
    if (live && (lane & 7u) == 0u) {                               // control byte: first symbol of the group in bit 7
        uint32_t bits = (uint32_t)(lits >> lane) & 0xFFu;
        bits = __builtin_bitreverse32(bits) >> 24;
        out[at] = (uint8_t)bits;
    }
    const uint32_t nib_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nib, 0x130, 0xF, 0xF, false);   // wave_shl:1 : the next lane's nibble
    if (live && (lane & 1u) == 0u) out[at + (uint32_t)((lane & 7u) == 0u)] = (uint8_t)((nib << 4) | nib_next);
    const uint32_t pay_at = at + extra;
    if (live) {
        if (lit) {
            // nib + 1 bytes in pieces of 16 / 8 / 4 / 2 / 1 (unaligned stores are fine on gfx950; nothing is written past the literal)
            const uint32_t len = nib + 1u;
            uint8_t* q = out + pay_at;
            if (len == 16u) { __builtin_memcpy(q, &v, 16); }
            else {
                uint32_t w0 = v.x, w1 = v.y, w2 = v.z, w3 = v.w;
                if (len & 8u) { const uint2 d = make_uint2(w0, w1); __builtin_memcpy(q, &d, 8); q += 8; w0 = w2; w1 = w3; }
                if (len & 4u) { __builtin_memcpy(q, &w0, 4); q += 4; w0 = w1; }
                if (len & 2u) { const uint16_t d = (uint16_t)w0; __builtin_memcpy(q, &d, 2); q += 2; w0 >>= 16; }
                if (len & 1u) *q = (uint8_t)w0;
            }
        } else {
            const uint16_t off = (uint16_t)rec;
            __builtin_memcpy(out + pay_at, &off, 2);
        }
    }
    const uint64_t live_lits = lits & below(cnt);
    if (live_lits) {
        const uint32_t last = 63u - (uint32_t)__builtin_clzll(live_lits);
        lit_out = rdlane(pay_at, last);
        lit_src = rdlane(rec, last) & 0x3FFFFFu;
    }
    return EmitResult{end, lit_out, lit_src};
}

#endif
// ctl words shared by the two waves below: kCtlSyms = symbols recorded so far, kCtlBatches = batches of 64 laid out,
// kCtlEnd = 1 + the block's symbol count once the last symbol is recorded (0 before).
enum : uint32_t { kCtlSyms = 7, kCtlBatches = 8, kCtlEnd = 9 };

// BUILDER: items -> symbol records in the ring.  An item carries the parse state at its entry, so items are independent: with
// Cfg::DUAL_BUILDER two wavefronts take them alternately (wavefront `which` the items k = which mod 2).  Each publishes the next
// item it will take (kCtlTail0 / kCtlTail1: ACCOUNT reuses a queue slot once both are past it), and the symbol count (kCtlSyms,
// what EMIT waits for) in item order: item k's count goes out once the other wavefront is past item k-1.
enum : uint32_t { kCtlTail0 = 1, kCtlTail1 = 38 };
template <class Cfg>
__device__ __forceinline__ void stream_builder(lds_u8_t* lds, uint32_t lane, uint32_t which)
{
    volatile lds_u32_t* queue = (volatile lds_u32_t*)(lds + Cfg::off_queue);
    volatile lds_u32_t* ring = (volatile lds_u32_t*)(lds + Cfg::off_ring);
    lds_u32_t* ctl = (lds_u32_t*)(lds + Cfg::off_ctl);
    constexpr uint32_t STEP = Cfg::DUAL_BUILDER ? 2u : 1u;
    uint32_t tail = which, batches_seen = 0, final_nsym = 0, other_seen = 0;
    const uint32_t my_word = which ? kCtlTail1 : kCtlTail0, other_word = which ? kCtlTail0 : kCtlTail1;

#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
    const unsigned long long begin_ = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        if (__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= tail) {
#ifdef TSQ_STATS
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            bool ended = false;
            while (__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= tail) {
                // (the other wavefront took the end item: nothing more is coming)
                if (Cfg::DUAL_BUILDER && uniform(__hip_atomic_load(&ctl[kCtlEnd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) { ended = true; break; }
                TSQ_SPIN(ctl); __builtin_amdgcn_s_sleep(1);
            }
#ifdef TSQ_STATS
            st_[17] += __builtin_amdgcn_s_memtime() - w0_;
#endif
            if (ended) break;
        }
        volatile lds_u32_t* it = queue + (tail % Cfg::Q) * Cfg::ITEM_WORDS;
        TSQ_DELAY(8);
        const uint32_t kind = uniform(it[0]);
        const uint32_t nsym_entry = uniform(it[4]);
        uint32_t nsym_after = nsym_entry;
        // room in the ring for everything this item can add (at most 65 symbols) behind the batches not laid out yet
        if (nsym_entry + 66u - 64u * batches_seen > Cfg::RING) {
#ifdef TSQ_STATS
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            for (;;) {
                batches_seen = uniform(__hip_atomic_load(&ctl[kCtlBatches], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (nsym_entry + 66u - 64u * batches_seen <= Cfg::RING) break;
                TSQ_SPIN_AT(ctl, 47u);
                __builtin_amdgcn_s_sleep(2);
            }
#ifdef TSQ_STATS
            st_[19] += __builtin_amdgcn_s_memtime() - w0_;
#endif
        }
        if (kind == kItemSym) {
            if (lane == 0) ring[nsym_entry & (Cfg::RING - 1u)] = it[9];
            nsym_after = nsym_entry + 1u;
        } else if (kind == kItemSeg) {
            const uint32_t base = uniform(it[1]);
            const uint64_t V = (uint64_t)uniform(it[2]) | ((uint64_t)uniform(it[3]) << 32);
            const uint32_t origin_entry = uniform(it[5]);
            uint32_t lit_from = uniform(it[6]);
            const uint64_t certain_m = (uint64_t)uniform(it[7]) | ((uint64_t)uniform(it[8]) << 32);
            const uint32_t lane_word = it[16 + lane];
            const uint32_t cand0 = lane_word & 0xFFFFFFu, nib = lane_word >> 24;
            const uint32_t p = base + lane;
            const uint64_t M = V & certain_m, N = V & ~certain_m;
            const uint32_t Ls = lsb64(V);
            uint32_t idx0 = nsym_entry;
            if (((M >> Ls) & 1ull) && lit_from < base + Ls) {
                // a literal run ended exactly at the segment boundary: its pending bytes close here
                if (lane == 0) ring[idx0 & (Cfg::RING - 1u)] = rec_literal(lit_from, base + Ls - lit_from);
                idx0++;
                lit_from = base + Ls;
            }
            // the pair origin seen by an odd first symbol: the parser's origin, unless the literal above was symbol idx0-1
            const uint32_t first_prev_start = (idx0 != nsym_entry) ? uniform(it[6]) : origin_entry;
            const uint32_t len_first = ((N >> Ls) & 1ull) ? ones_from(N, Ls) : 0u;
            const uint64_t startN = N & ~(N << 1);
            const bool isM = (M >> lane) & 1ull, isN = (N >> lane) & 1ull;
            const bool in_first = lane >= Ls && lane < Ls + len_first;
            const uint64_t sb = startN & below(lane + 1u);
            const uint32_t rs_lane = sb ? msb64(sb) : 0u;
            const uint32_t rs_pos = in_first ? lit_from : base + rs_lane;
            const uint32_t off = p - rs_pos;
            const bool next_isM = lane < 63u && ((M >> (lane + 1u)) & 1ull);
            const bool ownerN = isN && ((off & 15u) == 15u || next_isM);
            const bool sym = isM || ownerN;
            const uint64_t SS = __ballot(sym);
            const uint64_t before = SS & below(lane);
            const uint32_t idx = idx0 + (uint32_t)__builtin_popcountll(before);
            const uint32_t sym_start = isM ? p : p - (off & 15u);
            const uint32_t prev_start_v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((before ? msb64(before) : 0u) << 2), (int)sym_start);
            const uint32_t pair_origin = (idx & 1u) ? (before ? prev_start_v : first_prev_start) : sym_start;
#ifdef TSQ_STATS
            if (blockIdx.x == 0 && sym && idx < 2048u) g_dbg_syms[2048u + idx] = isM ? rec_match(pair_origin - cand0, nib) : rec_literal(sym_start, (off & 15u) + 1u);
            if (blockIdx.x == 0 && lane == 0 && tail < 400u) { g_dbg_syms[4096u + 4u * tail] = nsym_entry; g_dbg_syms[4097u + 4u * tail] = base; g_dbg_syms[4098u + 4u * tail] = (uint32_t)V; g_dbg_syms[4099u + 4u * tail] = batches_seen; }
#endif
            if (sym) ring[idx & (Cfg::RING - 1u)] = isM ? rec_match(pair_origin - cand0, nib) : rec_literal(sym_start, (off & 15u) + 1u);
            nsym_after = idx0 + (uint32_t)__builtin_popcountll(SS);
        }
        // the item is consumed (its words are in registers, the records are on their way to the ring: the LDS executes a
        // wavefront's operations in order, so the counters below become visible after them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // in item order: the other wavefront is past item tail - 1 (its records are in the ring and its count is out)
        if (Cfg::DUAL_BUILDER && tail != 0u && other_seen < tail + 1u) {
#ifdef TSQ_STATS
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            for (;;) {
                other_seen = uniform(__hip_atomic_load(&ctl[other_word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (other_seen >= tail + 1u) break;
                TSQ_SPIN(ctl);
            }
            asm volatile("" ::: "memory");
#ifdef TSQ_STATS
            st_[17] += __builtin_amdgcn_s_memtime() - w0_;
#endif
        }
        tail += STEP;
        // (stores by every lane, and the loop exit on a plain uniform test: a `lane == 0` block that also tests `kind` in front of
        // the break was compiled into a divergent exit that dropped lanes 1..63 after the first item)
        // (the count first: when the other wavefront sees this one past the item, the count it then stores comes later)
        __hip_atomic_store(&ctl[kCtlSyms], nsym_after, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(&ctl[my_word], tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        final_nsym = nsym_after;
        if (kind == kItemEnd) {
            asm volatile("" ::: "memory");
            __hip_atomic_store(&ctl[kCtlEnd], final_nsym + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0 && which == 0u) { g_enc_stats[17] = st_[17] + st_[19]; g_enc_stats[18] = __builtin_amdgcn_s_memtime() - begin_; g_enc_stats[37] = st_[19]; }
#endif
}

// EMIT wave: every 64 symbols of the ring -> control bytes, size bytes and payloads of the block stream; the block's tail
// (tsq_encode.cpp:176-188) and its size.
template <class Cfg>
__device__ __forceinline__ void stream_emitter(const uint8_t* src, uint64_t avail, uint8_t* out, lds_u8_t* lds, uint32_t lane,
                                               uint32_t b, uint32_t* sizes, int32_t* status)
{
    volatile lds_u32_t* ring = (volatile lds_u32_t*)(lds + Cfg::off_ring);
    lds_u32_t* ctl = (lds_u32_t*)(lds + Cfg::off_ctl);
    uint32_t batches = 0, nsym = 0, j0 = 3, lit_out = 0xFFFFFFFFu, lit_src = 0;
    bool overflow = false;
#ifdef TSQ_STATS
    unsigned long long waited_ = 0;
    const unsigned long long begin_ = __builtin_amdgcn_s_memtime();
#endif
    auto flush_batch = [&](uint32_t first_index, uint32_t cnt) {
        if (overflow) return;
        const uint32_t rec = ring[(first_index + lane) & (Cfg::RING - 1u)];
#ifdef TSQ_STATS
        if (blockIdx.x == 0 && first_index + lane < 2048u && lane < cnt) g_dbg_syms[first_index + lane] = rec;
#endif
        const EmitResult r = emit_batch(rec, cnt, j0, out, src, avail, lane);
        j0 = uniform(r.end);
        const uint32_t lo = uniform(r.lit_out);
        if (lo != 0xFFFFFFFFu) { lit_out = lo; lit_src = uniform(r.lit_src); }
        if (j0 + 1200u > kSlotSize) overflow = true;
    };
    for (;;) {
        // (the end marker is read first: the builder sets it after the symbol counter, so a counter read behind it is final)
        const uint32_t end = uniform(__hip_atomic_load(&ctl[kCtlEnd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        asm volatile("" ::: "memory");
        const uint32_t have = uniform(__hip_atomic_load(&ctl[kCtlSyms], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        asm volatile("" ::: "memory");
        if (have >= 64u * (batches + 1u)) {
            flush_batch(64u * batches, 64u);
            TSQ_DELAY(9);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the records are read: the ring entries may be reused
            batches++;
            if (lane == 0) __hip_atomic_store(&ctl[kCtlBatches], batches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
        }
        if (end != 0u) { nsym = end - 1u; break; }
#ifdef TSQ_STATS
        const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
        TSQ_SPIN(ctl);
        __builtin_amdgcn_s_sleep(8);
#ifdef TSQ_STATS
        waited_ += __builtin_amdgcn_s_memtime() - w0_;
#endif
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[38] = waited_; g_enc_stats[39] = __builtin_amdgcn_s_memtime() - begin_; }
#endif
    if (overflow) { if (lane == 0) { atomicMax(status, kErrOverflow); sizes[b] = 3; } return; }
    const uint32_t rest = nsym - 64u * batches;
    if (rest) flush_batch(nsym - rest, rest);
    if (overflow) { if (lane == 0) { atomicMax(status, kErrOverflow); sizes[b] = 3; } return; }
    auto stale = [&](uint32_t pos) -> uint32_t {
        const uint32_t d = pos - lit_out;
        return (lit_out != 0xFFFFFFFFu && d < 16u) ? ldu8z(src, (uint64_t)lit_src + d, avail) : 0u;
    };
    uint32_t total = j0;
    if ((nsym & 7u) == 0u) {
        if (lane == 0) { out[j0] = (uint8_t)stale(j0); out[j0 + 1] = (uint8_t)stale(j0 + 1); }
        total = j0 + 2;
    } else if ((nsym & 1u) == 0u) {
        if (lane == 0) out[j0] = (uint8_t)(stale(j0) << 4);
        total = j0 + 1;
    }
    if (lane == 0) sizes[b] = total;
}

}  // namespace tsq
