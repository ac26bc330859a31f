// tsq_enc_tile.cuh -- three-wave tile-pipelined block encoder for gfx950 (A/B variant 5: superseded, not in the product library).
//
// One workgroup of three wavefronts per 4 MiB block, each on its own SIMD:
//
//   wave 0  FRONT    classifies 64-aligned tiles of positions one tile AHEAD of the parser: input
//                    words, hashes, table gather, candidate bytes, common prefix, lane classes
//                    (tsq_enc_orbit.cuh), same-hash ("twin") masks; it also performs the table
//                    commits, so that its own later gathers are ordered behind them.
//   wave 1  PARSER   the irreducible serial part: orbit over the tile's span register, O(1) state
//                    accounting per segment, the exact scalar hazard path (tsq_enc_pipe.cuh).
//   wave 2  BUILDER  symbol records + stream layout (tsq_enc_pipe.cuh: pipe_builder).
//
// The front classifies tile t against a table that holds the commits of tiles <= t-2; whether the
// commit of tile t-1 is already in it is a race that does not matter: a lane of tile t whose hash
// equals that of a lane of tile t-1 (a "previous-tile twin") is a stop lane, and the parser takes
// the most recent VISITED twin as its candidate -- exactly what the committed table would hold --
// or keeps the gathered one when no twin was visited.  Twins inside a tile work the same way.
// Twins are found with the byte-per-bucket owner image in LDS: every lane reads its bucket (who
// wrote it last: a lane of the previous tile?), writes itself, and reads it back (another lane of
// this tile?).
#pragma once

#include "../tsq_common.cuh"
#include "tsq_enc_pipe.cuh"

namespace tsq {

struct TileCfg {
    static constexpr uint32_t Q = 32;
    static constexpr uint32_t ITEM_WORDS = 80;
    static constexpr uint32_t RING = 128;
    static constexpr uint32_t REC_WORDS = 16 + 10 * 64;        // header + 10 per-lane arrays
    static constexpr uint32_t off_owner = 0;                                   // u8[kHashEntries]
    static constexpr uint32_t off_queue = kHashEntries;                        // u32[Q * ITEM_WORDS]
    static constexpr uint32_t off_ring = off_queue + Q * ITEM_WORDS * 4;       // u32[RING]
    static constexpr uint32_t off_rec = off_ring + RING * 4;                   // u32[2 * REC_WORDS]
    static constexpr uint32_t off_ctl = off_rec + 2 * REC_WORDS * 4;           // u32[16]
    static constexpr uint32_t total = off_ctl + 64;
};
// ctl words: 0 queue head, 1 queue tail, 2 tiles classified, 3 tiles parsed, 4 stop,
//            8 + 2*(t&3): visited mask of tile t (lo, hi)
// record header: 0,1 hard  2,3 flagged  4,5 certain  6,7 near-twin      arrays: 0 spanword 1 cand|nib 2 orbit halt 3 w 4,5 twin_in 6,7 twin_prev 8,9 orbit mask
// spanword: natural span (bits 0..7) | stop (bit 8) | common prefix (bits 16..23)

template <bool EXT>
__device__ __forceinline__ void tile_front(const uint8_t* src, uint64_t avail, uint32_t n, uint16_t* table, lds_u8_t* lds, uint32_t lane)
{
    volatile lds_u8_t* owner = lds + TileCfg::off_owner;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + TileCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + TileCfg::off_ctl);
    constexpr uint32_t kDMin = EXT ? 128u : 64u;
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;
    const uint32_t n_tiles = (n >> 6) + 3u;            // visits reach at most n + 63

    uint32_t h_m1 = 0xFFFFFFFFu, h_m2 = 0xFFFFFFFFu;    // hashes of tiles t-1 and t-2 (per lane)
    uint64_t twins_m1 = 0, twins_m2 = 0;                // in-tile "has an earlier twin" masks of those tiles

#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_T0();
    for (uint32_t t = 0; t < n_tiles; ++t) {
        // ---- commit tile t-2 once the parser has its visited mask (then the table holds tiles <= t-2)
        if (t >= 2u) {
            for (;;) {
                if (uniform(__hip_atomic_load(&ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= t - 1u) break;
                if (uniform(__hip_atomic_load(&ctl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) {
#ifdef TSQ_STATS
                    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[0] = st_[0]; g_enc_stats[1] = st_[1]; g_enc_stats[2] = st_[2]; g_enc_stats[3] = t; }
#endif
                    return;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            TSQ_ACC(0);
            const uint32_t slot = 8u + 2u * ((t - 2u) & 3u);
            const uint64_t vis = (uint64_t)uniform(__hip_atomic_load(&ctl[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) |
                                 ((uint64_t)uniform(__hip_atomic_load(&ctl[slot + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) << 32);
            const uint32_t p2 = ((t - 2u) << 6) + lane;
            // among equal hashes the highest visited lane must win: lanes with an earlier twin store afterwards, in order
            if (((vis & ~twins_m2) >> lane) & 1ull) table[h_m2] = (uint16_t)p2;
            uint64_t late = vis & twins_m2;
            while (late) {
                if (lane == lsb64(late)) table[h_m2] = (uint16_t)p2;
                late &= late - 1ull;
            }
            TSQ_ACC(1);
        }
        // ---- classify tile t
        const uint32_t base = t << 6;
        const uint32_t p = base + lane;
        const uint4 w16 = ld128z(src, p, avail);
        const uint32_t w = w16.x;
        const uint32_t h = hash4(w);
        TSQ_SUB(16);
        const uint32_t tag = 0x80u | ((t & 1u) << 6) | lane;        // bit 7: a valid entry (the image starts zeroed)
        // retire the entries of tile t-2 (unless tile t-1 has taken the bucket over): only tile t-1's stay valid
        if (t >= 2u && (owner[h_m2] & 0xC0u) == (0x80u | ((t & 1u) << 6))) owner[h_m2] = 0;
        const uint32_t before = owner[h];                // non-zero: a lane of tile t-1 has this hash
        owner[h] = (uint8_t)tag;
        const uint32_t tv = table[h];
        const uint32_t cand0 = candidate_of(tv, p);
        TSQ_SUB(17);
        uint32_t k0 = prefix16(w16, ld128z(src, cand0, avail));
        if (EXT) {
            uint32_t more = 16;
            while (__ballot(k0 == more) != 0ull && more < 64u) {
                if (k0 == more) k0 += prefix16(ld128z(src, (uint64_t)p + more, avail), ld128z(src, (uint64_t)cand0 + more, avail));
                more += 16;
            }
        }
        TSQ_SUB(18);
        // twins inside the tile: for each lane the mask of EARLIER lanes with the same hash
        uint64_t twin_in = 0, twins_here = 0;
        {
            uint64_t shared = __ballot(owner[h] != (uint8_t)tag);
            while (shared) {
                const uint32_t hl = rdlane(h, lsb64(shared));
                const uint64_t grp = __ballot(h == hl);
                if (h == hl) twin_in = grp & below(lane);
                twins_here |= grp & (grp - 1ull);
                shared &= ~grp;
            }
        }
        // twins in the previous tile: for each lane the mask of tile t-1 lanes with the same hash
        uint64_t twin_prev = 0;
        if (t > 0u) {
            uint64_t maybe = __ballot(before != 0u);
            while (maybe) {
                const uint32_t hl = rdlane(h, lsb64(maybe));
                const uint64_t grp_prev = __ballot(h_m1 == hl);
                const uint64_t grp_cur = __ballot(h == hl);
                if (h == hl) twin_prev = grp_prev;
                maybe &= ~grp_cur;
            }
        }
        TSQ_SUB(19);
        const uint32_t dist = p - cand0;
        const bool eq4 = k0 >= 4u;
        const bool far_enough = dist >= kDMin && dist <= 0xFFFEu;
        const bool tail = p >= tail_from;
        // a twin at most 3 positions back (runs of equal bytes): if it is visited it becomes the candidate and,
        // being closer than 4, can never match.  Such lanes are classed "no match" optimistically; the parser
        // verifies after the orbit that a near twin was indeed visited (else the lane goes through the exact path).
        bool neart = false;
        if (__ballot((twin_in | twin_prev) != 0ull) != 0ull) {          // wave-uniform: most tiles of text have no twin this close
            const uint64_t near_in = twin_in & ~below(lane >= 3u ? lane - 3u : 0u);
            const uint64_t near_prev = lane < 3u ? twin_prev & ~below(61u + lane) : 0ull;
            neart = (near_in | near_prev) != 0ull && !tail;
        }
        const bool certain = eq4 && far_enough && !tail && !neart;
        const uint32_t nib = length_nibble(k0 < 4u ? 4u : k0);
        const uint32_t span_nat = certain ? nibble_span(nib) : 1u;
        // offset = origin - cand <= p - cand: a candidate closer than 4 bytes can never pass (offset-4) < 0xFFFB
        // (tsq_encode.cpp:100), whatever the pair origin: such a lane is a plain "no match", not a hazard
        const bool hard_l = (eq4 && !far_enough && dist >= 4u && !neart) || tail;
        const bool twin_l = (twin_in | twin_prev) != 0ull;
        const uint64_t hard = __ballot(hard_l), flagged = __ballot(twin_l), certain_m = __ballot(certain), neart_m = __ballot(neart);
        // the whole orbit of every lane, by pointer doubling: `nx` = where the orbit started at this lane halts
        // (lane, or position past the tile: 7 bits | halted: bit 7), `orb` = the lanes it visits before that.
        // A hop halts when it lands on a hard lane or past the tile.  Twin lanes do not halt: the parser takes
        // the orbit optimistically and checks the visited twins afterwards.
        uint32_t nx;
        uint64_t orb = 1ull << lane;
        {
            const uint32_t self = lane | (hard_l ? 0x80u : 0u);                    // arriving at this lane: halts?
            const uint32_t c = lane + span_nat;                                     // < 128
            const uint32_t there = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((c & 63u) << 2), (int)self);
            nx = c >= 64u ? (c | 0x80u) : there;
            for (int round = 0; round < 6; ++round) {
                if (__ballot((nx & 0x80u) == 0u) == 0ull) break;                    // every orbit has halted
                const int at = (int)((nx & 63u) << 2);
                const uint32_t nx2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)nx);
                const uint32_t olo = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)(uint32_t)orb);
                const uint32_t ohi = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)(uint32_t)(orb >> 32));
                if ((nx & 0x80u) == 0u) { nx = nx2; orb |= (uint64_t)olo | ((uint64_t)ohi << 32); }
            }
        }

        TSQ_SUB(20);
        volatile lds_u32_t* rec = recs + (t & 1u) * TileCfg::REC_WORDS;
        if (lane == 0) {
            rec[0] = (uint32_t)hard; rec[1] = (uint32_t)(hard >> 32);
            rec[2] = (uint32_t)flagged; rec[3] = (uint32_t)(flagged >> 32);
            rec[4] = (uint32_t)certain_m; rec[5] = (uint32_t)(certain_m >> 32);
            rec[6] = (uint32_t)neart_m; rec[7] = (uint32_t)(neart_m >> 32);
        }
        rec[16 + 0 * 64 + lane] = span_nat | ((hard_l || twin_l) ? 256u : 0u) | (k0 << 16);
        rec[16 + 1 * 64 + lane] = cand0 | (nib << 24);
        rec[16 + 2 * 64 + lane] = nx;
        rec[16 + 3 * 64 + lane] = w;
        rec[16 + 4 * 64 + lane] = (uint32_t)twin_in;
        rec[16 + 5 * 64 + lane] = (uint32_t)(twin_in >> 32);
        rec[16 + 6 * 64 + lane] = (uint32_t)twin_prev;
        rec[16 + 7 * 64 + lane] = (uint32_t)(twin_prev >> 32);
        rec[16 + 8 * 64 + lane] = (uint32_t)orb;
        rec[16 + 9 * 64 + lane] = (uint32_t)(orb >> 32);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ctl[2], t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (prefetching the next tile's input words here was measured and does not pay: loads and stores retire in
        //  order, so the prefetch either delays this tile's gathers or makes the next step wait for the commit stores)
        h_m2 = h_m1; h_m1 = h;
        twins_m2 = twins_m1; twins_m1 = twins_here;
        TSQ_ACC(21);
#ifdef TSQ_STATS
        st_[2] = st_[16] + st_[17] + st_[18] + st_[19] + st_[20] + st_[21];
        if (blockIdx.x == 0 && lane == 0 && (t & 1023u) == 1023u) { g_enc_stats[0] = st_[0]; g_enc_stats[1] = st_[1]; g_enc_stats[2] = st_[2]; g_enc_stats[3] = t + 1; for (int q = 16; q < 22; ++q) g_enc_stats[q] = st_[q]; }
#endif
    }
}

// Orbit over precomputed four-hop words: one v_readlane per four visits.  On entry lane L (< 64) is
// a normal lane not yet in V.  Returns with L = the lane (or position past the tile, >= 64) the
// orbit halted on, which is NOT added to V.
__device__ __forceinline__ void orbit_run4(uint32_t hops, uint32_t& L, uint64_t& V)
{
    uint32_t w;
    uint64_t bit;
    asm volatile(
        "1:\n\t"
        "v_readlane_b32 %[w], %[hops], %[L]\n\t"
        "s_lshl_b64 %[bit], 1, %[L]\n\t"
        "s_or_b64 %[V], %[V], %[bit]\n\t"
        "s_and_b32 %[L], %[w], 0x7f\n\t"
        "s_bitcmp1_b32 %[w], 7\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_lshl_b64 %[bit], 1, %[L]\n\t"
        "s_or_b64 %[V], %[V], %[bit]\n\t"
        "s_bfe_u32 %[L], %[w], 0x70008\n\t"
        "s_bitcmp1_b32 %[w], 15\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_lshl_b64 %[bit], 1, %[L]\n\t"
        "s_or_b64 %[V], %[V], %[bit]\n\t"
        "s_bfe_u32 %[L], %[w], 0x70010\n\t"
        "s_bitcmp1_b32 %[w], 23\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_lshl_b64 %[bit], 1, %[L]\n\t"
        "s_or_b64 %[V], %[V], %[bit]\n\t"
        "s_bfe_u32 %[L], %[w], 0x70018\n\t"
        "s_bitcmp1_b32 %[w], 31\n\t"
        "s_cbranch_scc0 1b\n"
        "2:\n\t"
        : [L] "+s"(L), [V] "+s"(V), [w] "=&s"(w), [bit] "=&s"(bit)
        : [hops] "v"(hops)
        : "scc");
}

template <bool EXT>
__device__ __forceinline__ void tile_parser(const uint8_t* src, uint64_t avail, uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    volatile lds_u32_t* queue = (volatile lds_u32_t*)(lds + TileCfg::off_queue);
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + TileCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + TileCfg::off_ctl);

    uint32_t head = 0, tail_seen = 0;  // tail_seen: last value read of the builder's progress (re-read only when the queue looks full)
    uint32_t v = 1, nsym = 0, origin = 0, lit_from = 0;
    bool after_match = false;
    uint32_t run0 = 0, origin_r0 = 0, odd_r0 = 0;
    bool done = false;
    uint64_t vall_prev = 0;            // visited lanes of the previous tile

    auto slot_begin = [&]() -> volatile lds_u32_t* {
        while (head - tail_seen >= TileCfg::Q) {
            tail_seen = uniform(__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (head - tail_seen >= TileCfg::Q) __builtin_amdgcn_s_sleep(2);
        }
        return queue + (head % TileCfg::Q) * TileCfg::ITEM_WORDS;
    };
    auto slot_publish = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        head++;
        __hip_atomic_store(&ctl[0], head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto push = [&](uint32_t record, uint32_t origin_if_pair_closes) {
        volatile lds_u32_t* it = slot_begin();
        if (lane == 0) { it[0] = kItemSym; it[4] = nsym; it[9] = record; }
        slot_publish();
        nsym++;
        if ((nsym & 1u) == 0u) origin = origin_if_pair_closes;
    };

#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_T0();
    for (uint32_t t = 0; !done; ++t) {
        const uint32_t base = t << 6;
        uint64_t vall = 0;
        if (v < base + 64u) {
            // ---- the tile's record, produced by the front
            TSQ_ACC(5);
            while (uniform(__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <= t) __builtin_amdgcn_s_sleep(1);
            TSQ_ACC(4); TSQ_CNT(6, 1);
            volatile lds_u32_t* rec = recs + (t & 1u) * TileCfg::REC_WORDS;
            const uint64_t hard = (uint64_t)uniform(rec[0]) | ((uint64_t)uniform(rec[1]) << 32);
            const uint64_t flagged_m = (uint64_t)uniform(rec[2]) | ((uint64_t)uniform(rec[3]) << 32);   // lanes with any twin
            const uint64_t certain_m = (uint64_t)uniform(rec[4]) | ((uint64_t)uniform(rec[5]) << 32);
            const uint64_t neart_m = (uint64_t)uniform(rec[6]) | ((uint64_t)uniform(rec[7]) << 32);
            const uint32_t spanword = rec[16 + 0 * 64 + lane];
            const uint32_t nx = rec[16 + 2 * 64 + lane];
            const uint32_t orb_lo = rec[16 + 8 * 64 + lane], orb_hi = rec[16 + 9 * 64 + lane];
            const uint32_t lane_word = rec[16 + 1 * 64 + lane];
            const uint32_t w = rec[16 + 3 * 64 + lane];
            const uint32_t tin_lo = rec[16 + 4 * 64 + lane], tin_hi = rec[16 + 5 * 64 + lane];
            const uint32_t tpv_lo = rec[16 + 6 * 64 + lane], tpv_hi = rec[16 + 7 * 64 + lane];
            const uint32_t span_nat = spanword & 0xFFu;
            const uint32_t k0 = (spanword >> 16) & 0xFFu;
            const uint32_t cand0 = lane_word & 0xFFFFFFu;

            // visited twins of lane L: in this tile (before L) and in the previous tile
            auto visited_twins = [&](uint32_t L, uint64_t vis_here, uint64_t& in_tile, uint64_t& in_prev) {
                in_tile = ((uint64_t)rdlane(tin_lo, L) | ((uint64_t)rdlane(tin_hi, L) << 32)) & vis_here;
                in_prev = ((uint64_t)rdlane(tpv_lo, L) | ((uint64_t)rdlane(tpv_hi, L) << 32)) & vall_prev;
            };

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

            auto visit_serial = [&](uint32_t L) {
                const uint32_t i = base + L;
                uint32_t cand = rdlane(cand0, L);
                uint32_t k = rdlane(k0, L);
                bool e4 = k >= 4u;
                {
                    uint64_t in_tile, in_prev;
                    visited_twins(L, vall, in_tile, in_prev);
                    if (in_tile | in_prev) {                    // the most recent visited twin is the candidate
                        uint32_t wq;
                        if (in_tile) { const uint32_t q = msb64(in_tile); cand = base + q; wq = rdlane(w, q); }
                        else { const uint32_t q = msb64(in_prev); cand = base - 64u + q; wq = uniform(ldu32z(src, cand, avail)); }
                        e4 = rdlane(w, L) == wq;
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
                if (k < 4u || !offset_ok(room)) { new_run(); return; }
                const uint32_t m = length_nibble(k);
                const uint32_t ni = i + nibble_span(m);
                push(rec_match(room, m), ni);
                after_match = true;
                lit_from = ni;
                v = ni;
            };

            // exact effect of a segment on the parse state, one step per literal RUN or match (used when a run
            // reaches a 16-byte chunk boundary inside the segment: incompressible data, runs of equal bytes)
            auto replay_segment = [&](uint64_t V) {
                const uint64_t N = V & ~certain_m;
                const uint32_t Le = msb64(V);
                uint32_t L = lsb64(V);
                while (L <= Le) {
                    const uint32_t q = base + L;
                    if ((N >> L) & 1ull) {                       // a run of literal bytes starting at lane L
                        const uint32_t len = ones_from(N, L);
                        if (after_match) { run0 = q; origin_r0 = origin; odd_r0 = nsym & 1u; after_match = false; }
                        const uint32_t full = (q + len - lit_from) >> 4;            // 16-byte chunks that complete inside the run
                        if (full) {
                            nsym += full;
                            if ((nsym & 1u) == 0u) origin = lit_from + 16u * full;   // the last chunk closed a pair
                            else if (full >= 2u) origin = lit_from + 16u * (full - 1u);   // the one before it did
                            lit_from += 16u * full;
                        }
                        L += len;
                    } else {                                     // a certain match
                        const uint32_t sp = rdlane(span_nat, L);
                        if (lit_from < q) { nsym++; if ((nsym & 1u) == 0u) origin = q; }
                        nsym++;
                        if ((nsym & 1u) == 0u) origin = q + sp;
                        lit_from = q + sp;
                        after_match = true;
                        L += sp;
                    }
                }
            };
            auto account_segment = [&](uint64_t V) {
                if (V == 0ull) return;
                const uint64_t M = V & certain_m, N = V & ~certain_m;
                const uint32_t Ls = lsb64(V), Le = msb64(V);
                const bool first_isN = (N >> Ls) & 1ull;
                uint64_t r = N & (N >> 1); r &= r >> 2; r &= r >> 4; r &= r >> 8;
                const uint32_t carried = first_isN ? base + Ls - lit_from : 0u;
                const uint32_t first_len = first_isN ? ones_from(N, Ls) : 0u;
                if (r != 0ull || carried + first_len >= 16u) { TSQ_CNT(28, 1); replay_segment(V); return; }
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

            uint32_t L = v - base;
            TSQ_ACC(7);
            while (!done) {
                uint64_t V = 0;
                const uint32_t seg_nsym = nsym, seg_origin = origin, seg_lit_from = lit_from;
                bool tile_end = false;
                if (L < 64u && !((hard >> L) & 1ull)) {                          // the orbit from L: halts on a hard lane or past the tile
                    V = (uint64_t)rdlane(orb_lo, L) | ((uint64_t)rdlane(orb_hi, L) << 32);
                    L = rdlane(nx, L) & 0x7Fu;
                }
                if (V & flagged_m) {
                    // the orbit treated twin lanes as ordinary lanes.  That is wrong for a visited lane that has a
                    // VISITED twin before it (earlier in this tile, or in the previous tile): its gathered candidate
                    // is not current.  The first such lane ends the segment; everything before it is exact.
                    // Near-twin lanes (classed "no match" on the assumption that a twin at most 3 back is visited) are the
                    // other way round: they are right exactly when such a twin was visited.
                    const uint64_t seen = vall | V;
                    const uint32_t in_lo = tin_lo & (uint32_t)seen, in_hi = tin_hi & (uint32_t)(seen >> 32);
                    const uint32_t pv_lo = tpv_lo & (uint32_t)vall_prev, pv_hi = tpv_hi & (uint32_t)(vall_prev >> 32);
                    const bool has_in = (in_lo | in_hi) != 0u, has_prev = (pv_lo | pv_hi) != 0u;
                    const uint32_t nearest = in_hi ? 63u - (uint32_t)__builtin_clz(in_hi) : 31u - (uint32_t)__builtin_clz(in_lo | 1u);
                    const uint32_t nearest_prev = pv_hi ? 63u - (uint32_t)__builtin_clz(pv_hi) : 31u - (uint32_t)__builtin_clz(pv_lo | 1u);
                    const bool near_visited = (has_in && lane - nearest < 4u) || (has_prev && lane + 64u - nearest_prev < 4u);
                    const bool is_near = (neart_m >> lane) & 1ull;
                    const bool stale = ((V >> lane) & 1ull) && (is_near ? !near_visited : (has_in || has_prev));
                    const uint64_t bad = __ballot(stale);
                    if (bad) { L = lsb64(bad); V &= below(L); TSQ_CNT(23, 1); }
                    TSQ_CNT(22, 1);
                }
                tile_end = L >= 64u;
                TSQ_CNT(24, 1); TSQ_CNT(25, V != 0ull ? 1 : 0);
                TSQ_ACC(8);
                account_segment(V);
                TSQ_ACC(10);
                send_segment(V, seg_nsym, seg_origin, seg_lit_from);
                TSQ_ACC(11);
                vall |= V;
                if (tile_end) { v = base + L; break; }
                TSQ_CNT(26, 1); TSQ_CNT(27, ((hard >> L) & 1ull) ? 1 : 0);
                visit_serial(L);
                TSQ_ACC(12);
                L = v - base;
            }
        }
        // ---- hand the tile's visited mask to the front (it commits the table) and move on
        if (lane == 0) {
            const uint32_t slot = 8u + 2u * (t & 3u);
            __hip_atomic_store(&ctl[slot], (uint32_t)vall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&ctl[slot + 1u], (uint32_t)(vall >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ctl[3], t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        vall_prev = vall;
    }
#ifdef TSQ_STATS
    TSQ_ACC(5);
    if (blockIdx.x == 0 && lane == 0) { for (int q = 4; q < 13; ++q) g_enc_stats[q] = st_[q]; g_enc_stats[9] = nsym; for (int q = 22; q < 29; ++q) g_enc_stats[q] = st_[q]; }
#endif
    if (lane == 0) __hip_atomic_store(&ctl[4], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    {
        volatile lds_u32_t* it = slot_begin();
        if (lane == 0) { it[0] = kItemEnd; it[4] = nsym; }
        slot_publish();
    }
}

template <bool EXT>
__global__ __launch_bounds__(192) void enc_tile_kernel(const uint8_t* __restrict__ in, uint64_t n_total, uint64_t readable,
                                                       uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                       uint16_t* __restrict__ tables, int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t tile_lds[];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t role = uniform(threadIdx.x >> 6);
    const uint64_t start = (uint64_t)b << kBlockBits;
    const uint64_t avail = readable - start;
    const uint32_t n = n_total - start < kBlockSize ? (uint32_t)(n_total - start) : kBlockSize;
    const uint8_t* src = in + start;
    uint8_t* out = slots + (size_t)b * kSlotSize;
    uint16_t* table = tables + (size_t)b * kHashEntries;

    {   // tsqInit (tsq_context.cpp:77-80), all three waves
        uint4* t4 = reinterpret_cast<uint4*>(table);
        for (uint32_t k = threadIdx.x; k < kHashEntries * 2 / 16; k += 192) t4[k] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < 16) reinterpret_cast<uint32_t*>(tile_lds + TileCfg::off_ctl)[threadIdx.x] = 0;
        uint4* o4 = reinterpret_cast<uint4*>(tile_lds + TileCfg::off_owner);          // owner image: no valid entries
        for (uint32_t k = threadIdx.x; k < kHashEntries / 16; k += 192) o4[k] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) { out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); }
    }
    __syncthreads();
    lds_u8_t* lds3 = (lds_u8_t*)tile_lds;
    if (role == 0) tile_front<EXT>(src, avail, n, table, lds3, lane);
    else if (role == 1) tile_parser<EXT>(src, avail, n, lds3, lane);
    else stream_builder<TileCfg>(src, avail, out, lds3, lane, b, sizes, status);
}

}  // namespace tsq
