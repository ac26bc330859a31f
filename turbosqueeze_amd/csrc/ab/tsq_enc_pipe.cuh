// tsq_enc_pipe.cuh -- two-wave pipelined block encoder for gfx950 (A/B variant 4: superseded, not in the product library).
//
// One workgroup of two wavefronts per 4 MiB block, specialised by role; they run on different
// SIMDs of the CU, so their instruction streams issue concurrently (a lone wavefront issues about
// one instruction per 5 cycles, which is what bounds the single-wave encoders):
//
//   wave 0  PARSER   window gathers + lane classification (see tsq_enc_orbit.cuh for the classes),
//                    the orbit loop, the exact scalar hazard path, table commit.  It keeps the
//                    parse state (symbol count, pair origin, pending literal start, current run)
//                    incrementally inside the orbit loop -- a few scalar ops per visit -- so it
//                    never needs anything back from the builder.
//   wave 1  BUILDER  receives, through a single-producer/single-consumer queue in LDS, one item per
//                    hazard-free segment (visited mask + per-lane candidate data + the parser's
//                    state at segment entry) or per scalar symbol, builds the symbol records in
//                    parallel, and lays out the stream (emit_batch) every 64 symbols.
//
// The serial dependence of the format (tsq_encode.cpp: the table depends on the parse) stays on
// the parser; everything that is a pure function of the parse moved to the builder.
#pragma once

#include "../tsq_common.cuh"
#include "../tsq_enc_util.cuh"
#include "../tsq_enc_builder.cuh"
#include "tsq_enc_orbit.cuh"

namespace tsq {

struct PipeCfg {
    static constexpr uint32_t Q = 32;                 // queue items
    static constexpr uint32_t ITEM_WORDS = 80;        // 16 header words + 64 lane words
    static constexpr uint32_t RING = 128;             // symbol records (builder private)
    static constexpr uint32_t off_owner = 0;                                   // u8[kHashEntries]
    static constexpr uint32_t off_queue = kHashEntries;                        // u32[Q * ITEM_WORDS]
    static constexpr uint32_t off_ring = off_queue + Q * ITEM_WORDS * 4;       // u32[RING]
    static constexpr uint32_t off_ctl = off_ring + RING * 4;                   // u32[4]: head, tail
    static constexpr uint32_t total = off_ctl + 64;
};


// The orbit: L += span[L] until a stop lane (span >= 128) or the end of the window, collecting
// the visited lanes.  Hand-written because hipcc turns the equivalent C++ loop into ~18 scalar
// instructions and 4 branches per visit; here a visit is 7 instructions, the exit test is a
// not-taken branch and the loop is unrolled so that a taken branch is paid once per 4 visits.
// A stop lane carries span 128: adding it leaves the window, and it is backed out afterwards.
// (SGPR lane select written by SALU, readlane result consumed by SALU: no manual wait states.)
constexpr uint32_t kStopSpan = 128;
#define TSQ_ORBIT_STEP                                   \
    "v_readlane_b32 %[s], %[span], %[L]\n\t"             \
    "s_lshl_b64 %[bit], 1, %[L]\n\t"                     \
    "s_mov_b32 %[prev], %[L]\n\t"                        \
    "s_or_b64 %[V], %[V], %[bit]\n\t"                    \
    "s_add_u32 %[L], %[L], %[s]\n\t"                     \
    "s_cmp_gt_u32 %[L], 63\n\t"                          \
    "s_cbranch_scc1 2f\n\t"
// returns true when the orbit stopped on a stop lane (L = that lane, not in V), false at window end
__device__ __forceinline__ bool orbit_run(uint32_t span, uint32_t& L, uint64_t& V)
{
    uint32_t s = 0, prev = 0;
    uint64_t bit;
    asm volatile(
        "s_cmp_gt_u32 %[L], 63\n\t"
        "s_cbranch_scc1 2f\n"
        "1:\n\t"
        TSQ_ORBIT_STEP TSQ_ORBIT_STEP TSQ_ORBIT_STEP TSQ_ORBIT_STEP
        "s_branch 1b\n"
        "2:\n\t"
        : [L] "+s"(L), [V] "+s"(V), [s] "+s"(s), [prev] "+s"(prev), [bit] "=&s"(bit)
        : [span] "v"(span)
        : "scc");
    if (s >= kStopSpan) { L = prev; V &= ~(1ull << prev); return true; }
    return false;
}

template <bool EXT>
__device__ __forceinline__ void pipe_parser(const uint8_t* src, uint64_t avail, uint32_t n, uint16_t* table,
                                            lds_u8_t* lds, uint32_t lane, uint32_t b)
{
    volatile lds_u8_t* bucket_owner = lds + PipeCfg::off_owner;
    volatile lds_u32_t* queue = (volatile lds_u32_t*)(lds + PipeCfg::off_queue);
    lds_u32_t* ctl = (lds_u32_t*)(lds + PipeCfg::off_ctl);
    constexpr uint32_t kDMin = EXT ? 128u : 64u;
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;
    (void)b;

    uint32_t head = 0, tail_seen = 0;   // items published so far; last value read of the builder's progress
    uint32_t v = 1, nsym = 0, origin = 0, lit_from = 0;
    bool after_match = false;
    uint32_t run0 = 0, origin_r0 = 0, odd_r0 = 0;
    bool done = false;

    // reserve the next queue slot (back-pressure on the builder), fill it, publish it
    auto slot_begin = [&]() -> volatile lds_u32_t* {
        while (head - tail_seen >= PipeCfg::Q) {
            tail_seen = uniform(__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (head - tail_seen >= PipeCfg::Q) __builtin_amdgcn_s_sleep(2);
        }
        return queue + (head % PipeCfg::Q) * PipeCfg::ITEM_WORDS;
    };
    auto slot_publish = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        head++;
        __hip_atomic_store(&ctl[0], head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // one scalar symbol (hazard path): counted here, materialised by the builder
    auto push = [&](uint32_t record, uint32_t origin_if_pair_closes) {
        volatile lds_u32_t* it = slot_begin();
        if (lane == 0) { it[0] = kItemSym; it[4] = nsym; it[9] = record; }
        slot_publish();
        nsym++;
        if ((nsym & 1u) == 0u) origin = origin_if_pair_closes;
    };

#ifdef TSQ_STATS
    unsigned long long st_[16] = {0};
#endif
    TSQ_T0();
    bool first = true;
    while (!done) {
        // ------------------------------------------------------------------ load + classify the window
        const uint32_t base = first ? 0u : v;
        first = false;
        const uint32_t p = base + lane;
        const uint4 w16 = ld128z(src, p, avail);
        const uint32_t w = w16.x;
        const uint32_t h = hash4(w);
        TSQ_ACC(13);
        bucket_owner[h] = (uint8_t)lane;
        const uint32_t t = table[h];
        const uint32_t cand0 = candidate_of(t, p);
        uint32_t k0 = prefix16(w16, ld128z(src, cand0, avail));
        if (EXT) {
            uint32_t more = 16;
            while (__ballot(k0 == more) != 0ull && more < 64u) {
                if (k0 == more) k0 += prefix16(ld128z(src, (uint64_t)p + more, avail), ld128z(src, (uint64_t)cand0 + more, avail));
                more += 16;
            }
        }
        uint64_t shared = __ballot(bucket_owner[h] != (uint8_t)lane);
        uint64_t flagged = 0;
        while (shared) {
            const uint64_t grp = __ballot(h == rdlane(h, lsb64(shared)));
            flagged |= grp & (grp - 1ull);
            shared &= ~grp;
        }
        const uint32_t dist = p - cand0;
        const bool eq4 = k0 >= 4u;
        const bool far_enough = dist >= kDMin && dist <= 0xFFFEu;
        const bool tail = p >= tail_from;
        const bool certain = eq4 && far_enough && !tail;
        const uint32_t nib = length_nibble(k0 < 4u ? 4u : k0);
        const uint32_t span_nat = certain ? nibble_span(nib) : 1u;
        const uint64_t hard = __ballot((eq4 && !far_enough) || tail);
        const uint64_t certain_m = __ballot(certain);
        const uint32_t span = (((hard | flagged) >> lane) & 1ull) ? kStopSpan : span_nat;
        const uint32_t lane_word = cand0 | (nib << 24);
        uint64_t vall = 0;
        TSQ_ACC(0); TSQ_CNT(4, 1);

        // hand one hazard-free segment to the builder; seg_* are the parser's state at its entry
        auto send_segment = [&](uint64_t V, uint32_t seg_nsym, uint32_t seg_origin, uint32_t seg_lit_from) {
            if (V == 0ull) return;
            volatile lds_u32_t* it = slot_begin();
            if (lane == 0) {
                it[0] = kItemSeg; it[1] = base; it[2] = (uint32_t)V; it[3] = (uint32_t)(V >> 32);
                it[4] = seg_nsym; it[5] = seg_origin; it[6] = seg_lit_from;
                it[7] = (uint32_t)certain_m; it[8] = (uint32_t)(certain_m >> 32);
            }
            it[16 + lane] = lane_word;
            slot_publish();
        };

        // exact scalar handling of one visit (hazard lanes); same logic as tsq_enc_orbit.cuh
        auto visit_serial = [&](uint32_t L) {
            const uint32_t i = base + L;
            uint32_t cand = rdlane(cand0, L);
            uint32_t k = rdlane(k0, L);
            bool e4 = k >= 4u;
            if ((flagged >> L) & 1ull) {
                const uint64_t grp = __ballot(h == rdlane(h, L)) & vall & below(L);
                if (grp) {
                    const uint32_t q = msb64(grp);
                    cand = base + q;
                    e4 = rdlane(w, L) == rdlane(w, q);
                    k = 0xFFu;
                }
            }
            vall |= 1ull << L;
            auto new_run = [&]() { after_match = false; run0 = i; origin_r0 = origin; odd_r0 = nsym & 1u; lit_from = i; v = i + 1u; };
            if (after_match) {
                if (!(i < n - 5u && e4 && offset_ok(origin - cand))) {               // tsq_encode.cpp:170
                    after_match = false;
                    if (!(i < n)) { done = true; return; }                           // tsq_encode.cpp:173
                    new_run();
                    return;
                }
            } else {
                const uint32_t f = (i - 1u - run0) >> 5;
                const uint32_t o_ref = f == 0u ? origin_r0 : (odd_r0 ? run0 + 32u * f - 16u : run0 + 32u * f);
                const bool ok = e4 && offset_ok(o_ref - cand);                        // tsq_encode.cpp:80,100
                if (i < n && !ok) {
                    v = i + 1u;
                    if (v - lit_from == 16u) { push(rec_literal(lit_from, 16u), v); lit_from = v; }
                    return;
                }
                if (i > lit_from) { push(rec_literal(lit_from, i - lit_from), i); lit_from = i; }   // tsq_encode.cpp:103-118
                if (!(i < n)) { done = true; return; }                               // tsq_encode.cpp:120
            }
            if (k == 0xFFu) {
                k = uniform(prefix16(ld128z(src, i, avail), ld128z(src, cand, avail)));
                if (EXT) {
                    while (k >= 16u && k < 64u && (k & 15u) == 0u) {
                        const uint32_t add = uniform(prefix16(ld128z(src, (uint64_t)i + k, avail), ld128z(src, (uint64_t)cand + k, avail)));
                        k += add;
                        if (add < 16u) break;
                    }
                }
            }
            const uint32_t room = origin - cand;
            if (k > room) k = room - 1u;
            if (k < 4u || !offset_ok(room)) { new_run(); return; }                   // the chain breaks without a symbol
            const uint32_t m = length_nibble(k);
            const uint32_t ni = i + nibble_span(m);
            push(rec_match(room, m), ni);
            after_match = true;
            lit_from = ni;
            v = ni;
        };

        // exact per-visit replay of a segment's effect on the parse state (used when a literal
        // run reaches a 16-byte chunk boundary inside the segment, which is rare on real data)
        auto replay_segment = [&](uint64_t V) {
            for (uint64_t m = V; m; m &= m - 1ull) {
                const uint32_t L = lsb64(m), q = base + L;
                if (!((certain_m >> L) & 1ull)) {            // a literal byte
                    if (after_match) { run0 = q; origin_r0 = origin; odd_r0 = nsym & 1u; after_match = false; }
                    if (q + 1u - lit_from == 16u) {          // a 16-byte chunk completes
                        nsym++;
                        if ((nsym & 1u) == 0u) origin = q + 1u;
                        lit_from = q + 1u;
                    }
                } else {                                     // a certain match
                    const uint32_t sp = rdlane(span_nat, L);
                    if (lit_from < q) { nsym++; if ((nsym & 1u) == 0u) origin = q; }     // the pending partial chunk closes
                    nsym++;
                    if ((nsym & 1u) == 0u) origin = q + sp;
                    lit_from = q + sp;
                    after_match = true;
                }
            }
        };
        // the same effect in O(1) from the masks: symbols are the matches plus one literal per run
        // that a match closes, the last symbol is always the last match, a trailing run stays open
        auto account_segment = [&](uint64_t V) {
            if (V == 0ull) return;
            const uint64_t M = V & certain_m, N = V & ~certain_m;
            const uint32_t Ls = lsb64(V), Le = msb64(V);
            const bool first_isN = (N >> Ls) & 1ull;
            uint64_t t = N & (N >> 1); t &= t >> 2; t &= t >> 4; t &= t >> 8;      // a run of 16 literal lanes
            const uint32_t carried = first_isN ? base + Ls - lit_from : 0u;
            const uint32_t first_len = first_isN ? ones_from(N, Ls) : 0u;
            if (t != 0ull || carried + first_len >= 16u) { replay_segment(V); return; }
            if (M == 0ull) {
                if (after_match) { run0 = base + Ls; origin_r0 = origin; odd_r0 = nsym & 1u; after_match = false; }
                return;
            }
            const uint32_t pre = (!first_isN && lit_from < base + Ls) ? 1u : 0u;
            const uint32_t Lm = msb64(M);
            const uint32_t endm = base + Lm + rdlane(span_nat, Lm);
            nsym += (uint32_t)__builtin_popcountll(M) + (uint32_t)__builtin_popcountll(M & (N << 1)) + pre;
            origin = (nsym & 1u) ? base + Lm : endm;
            lit_from = endm;
            if (Le == Lm) { after_match = true; }
            else { after_match = false; run0 = endm; origin_r0 = origin; odd_r0 = nsym & 1u; }
        };

        // ------------------------------------------------------------------ orbit over the window
        uint32_t L = v - base;
        while (!done) {
            uint64_t V = 0;
            const uint32_t seg_nsym = nsym, seg_origin = origin, seg_lit_from = lit_from;
            bool window_end = false;
            for (;;) {
                if (!orbit_run(span, L, V)) { window_end = true; break; }
                if ((hard >> L) & 1ull) break;
                // flagged only: harmless unless an earlier lane with the same hash was visited
                const uint64_t grp = __ballot(h == rdlane(h, L)) & (vall | V) & below(L);
                if (grp != 0ull) break;
                V |= 1ull << L;
                L += rdlane(span_nat, L);
            }
            TSQ_ACC(1);
            account_segment(V);
            TSQ_ACC(3);
            send_segment(V, seg_nsym, seg_origin, seg_lit_from);
            TSQ_ACC(14); TSQ_CNT(5, V ? 1 : 0);
            vall |= V;
            if (window_end) { v = base + L; break; }
            TSQ_CNT(7, 1);
            visit_serial(L);
            TSQ_ACC(8);
            L = v - base;
        }
        if (done) break;

        // ------------------------------------------------------------------ commit the window
        if (((vall & ~flagged) >> lane) & 1ull) table[h] = (uint16_t)p;
        uint64_t late = vall & flagged;
        while (late) {
            if (lane == lsb64(late)) table[h] = (uint16_t)p;
            late &= late - 1ull;
        }
        TSQ_ACC(2);
    }
#ifdef TSQ_STATS
    if (b == 0 && lane == 0) { st_[9] = nsym; for (int q = 0; q < 16; ++q) g_enc_stats[q] = st_[q]; }
#endif
    {   // end of stream
        volatile lds_u32_t* it = slot_begin();
        if (lane == 0) { it[0] = kItemEnd; it[4] = nsym; }
        slot_publish();
    }
}

template <bool EXT>
__global__ __launch_bounds__(128) void enc_pipe_kernel(const uint8_t* __restrict__ in, uint64_t n_total, uint64_t readable,
                                                       uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                       uint16_t* __restrict__ tables, int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pipe_lds[];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t role = uniform(threadIdx.x >> 6);
    const uint64_t start = (uint64_t)b << kBlockBits;
    const uint64_t avail = readable - start;
    const uint32_t n = n_total - start < kBlockSize ? (uint32_t)(n_total - start) : kBlockSize;
    const uint8_t* src = in + start;
    uint8_t* out = slots + (size_t)b * kSlotSize;
    uint16_t* table = tables + (size_t)b * kHashEntries;

    {   // tsqInit (tsq_context.cpp:77-80), both waves
        uint4* t4 = reinterpret_cast<uint4*>(table);
        for (uint32_t k = threadIdx.x; k < kHashEntries * 2 / 16; k += 128) t4[k] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < 4) reinterpret_cast<uint32_t*>(pipe_lds + PipeCfg::off_ctl)[threadIdx.x] = 0;
        if (threadIdx.x == 0) { out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); }
    }
    __syncthreads();
    lds_u8_t* lds3 = (lds_u8_t*)pipe_lds;
    if (role == 0) pipe_parser<EXT>(src, avail, n, table, lds3, lane, b);
    else stream_builder<PipeCfg>(src, avail, out, lds3, lane, b, sizes, status);
}

}  // namespace tsq
