// tsq_enc_orbit.cuh -- "orbit" block encoder for gfx950 (A/B variant 3: superseded, not in the product library).
//
// Same window structure as tsq_enc_fast.cuh (64 positions per window, one wavefront per block,
// gathers for candidate + common prefix per lane), but the serial part of the parse is reduced to
// its irreducible core.  A single wavefront issues roughly one instruction every 5 cycles, so the
// walk of tsq_enc_fast.cuh (about 200 scalar instructions per visit) is issue-bound; here the
// per-visit work is one v_readlane and an add:
//
//   classify  every lane is put in one of three classes that do NOT depend on the walk's state:
//               M  "certain match"  the 4 bytes match and the candidate is 64..65534 bytes back
//                  (128.. with extensions): whatever the pair origin is at that moment, the
//                  offset test passes and the no-overlap clamp cannot bite (tsq_encode.cpp:100,
//                  140-145), so a visit emits a match of the precomputed length;
//               N  "no match"       the 4 bytes differ: a visit makes the byte a literal;
//               H  "hazard"         everything whose outcome depends on the pair origin, on an
//                  earlier lane of the same window with the same hash, or on the block end.
//   orbit     next(L) = L + span[L] (span = match length for M, 1 for N, 0 for H).  The set of
//             visited lanes is the orbit of the entry lane: `while (L < 64) L += readlane(span, L)`.
//   build     from the visited mask alone, all lanes build the symbols in parallel: matches are
//             the visited M lanes; literal symbols are the 16-byte chunks of each run of visited
//             N lanes (a forced flush every 32 literals, tsq_encode.cpp:82-98, changes when
//             symbols are counted but not which symbols there are); symbol indices are popcounts,
//             the pair origin of an odd symbol is the start of the symbol before it.
//   hazard    an H lane ends the segment; it is resolved by the exact scalar logic with the
//             reference-time pair origin reconstructed in closed form, then the orbit resumes.
//
// Symbol records go to a 128-entry LDS ring; every 64 symbols emit_batch() lays them out.
#pragma once

#include "../tsq_common.cuh"
#include "../tsq_enc_util.cuh"
#include "../tsq_enc_builder.cuh"

namespace tsq {

constexpr uint32_t kOrbRing = 128;
constexpr uint32_t kOrbLds = kHashEntries + kOrbRing * 4;

template <bool EXT>
__global__ __launch_bounds__(64) void enc_orbit_kernel(const uint8_t* __restrict__ in, uint64_t n_total, uint64_t readable,
                                                       uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                       uint16_t* __restrict__ tables, int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t orb_lds[];
    volatile uint8_t* bucket_owner = orb_lds;                                        // kHashEntries bytes
    volatile uint32_t* ring = reinterpret_cast<volatile uint32_t*>(orb_lds + kHashEntries);
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint64_t start = (uint64_t)b << kBlockBits;
    const uint64_t avail = readable - start;
    const uint32_t n = n_total - start < kBlockSize ? (uint32_t)(n_total - start) : kBlockSize;
    const uint8_t* src = in + start;
    uint8_t* out = slots + (size_t)b * kSlotSize;
    uint16_t* table = tables + (size_t)b * kHashEntries;
    constexpr uint32_t kDMin = EXT ? 128u : 64u;       // candidate distance from which a match is origin-independent
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;  // the probe test is i < n-5 (tsq_encode.cpp:170): handle the tail exactly

    {   // tsqInit (tsq_context.cpp:77-80)
        uint4* t4 = reinterpret_cast<uint4*>(table);
        for (uint32_t k = lane; k < kHashEntries * 2 / 16; k += kWave) t4[k] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
    }
    if (lane == 0) { out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); }

#ifdef TSQ_STATS
    unsigned long long st_[16] = {0};
#endif
    TSQ_T0();

    // ---- parse state, all wave-uniform ----
    uint32_t v = 1;                // next position to visit (position 0 is never probed, tsq_encode.cpp:72)
    uint32_t nsym = 0;             // symbols produced so far
    uint32_t origin = 0;           // rep_last_i once every produced symbol is counted
    uint32_t lit_from = 0;         // first input byte not yet covered by a produced symbol
    bool after_match = false;      // the next visit is the probe that follows a match
    // the literal run in progress, as the reference saw it when the run began (for hazards)
    uint32_t run0 = 0, origin_r0 = 0, odd_r0 = 0;
    // ---- emit state ----
    uint32_t j0 = 3;
    uint32_t lit_out = 0xFFFFFFFFu, lit_src = 0;
    bool overflow = false, done = false;

    auto flush_batch = [&](uint32_t first_index, uint32_t cnt) {
#ifdef TSQ_STATS
        const unsigned long long tf0_ = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t rec = ring[(first_index + lane) & (kOrbRing - 1u)];
#ifdef TSQ_STATS
        if (b == 0 && first_index + lane < 8192u && lane < cnt) g_dbg_syms[first_index + lane] = rec;
#endif
        const EmitResult r = emit_batch(rec, cnt, j0, out, src, avail, lane);
        j0 = uniform(r.end);
        const uint32_t lo = uniform(r.lit_out);
        if (lo != 0xFFFFFFFFu) { lit_out = lo; lit_src = uniform(r.lit_src); }
        if (j0 + 1200u > kSlotSize) overflow = true;
#ifdef TSQ_STATS
        st_[11] += __builtin_amdgcn_s_memtime() - tf0_; st_[12]++;
#endif
    };
    // scalar append of one symbol (hazard path)
    auto push = [&](uint32_t record, uint32_t origin_if_pair_closes) {
        if (lane == 0) ring[nsym & (kOrbRing - 1u)] = record;
        nsym++;
        if ((nsym & 1u) == 0u) origin = origin_if_pair_closes;
        if ((nsym & 63u) == 0u) flush_batch(nsym - 64u, 64u);
    };

    bool first = true;
    while (!done && !overflow) {
        // ------------------------------------------------------------------ load + classify the window
        const uint32_t base = first ? 0u : v;
        first = false;
        const uint32_t p = base + lane;
        const uint4 w16 = ld128z(src, p, avail);
        const uint32_t w = w16.x;
        const uint32_t h = hash4(w);
        bucket_owner[h] = (uint8_t)lane;
        const uint32_t t = table[h];
        const uint32_t cand0 = candidate_of(t, p);
        uint32_t k0 = prefix16(w16, ld128z(src, cand0, avail));
        if (EXT) {
            // common prefix up to 64 (tsq_encode.cpp:276-290) for the lanes whose first 16 bytes agree
            uint32_t more = 16;
            while (__ballot(k0 == more) != 0ull && more < 64u) {
                if (k0 == more) k0 += prefix16(ld128z(src, (uint64_t)p + more, avail), ld128z(src, (uint64_t)cand0 + more, avail));
                more += 16;
            }
        }
        uint64_t shared = __ballot(bucket_owner[h] != (uint8_t)lane);
        uint64_t flagged = 0;          // lanes that have an EARLIER lane of this window with the same hash
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
        const uint32_t span = (((hard | flagged) >> lane) & 1ull) ? 0u : span_nat;
        uint64_t vall = 0;             // every lane visited in this window (for the table commit)
        TSQ_ACC(0); TSQ_CNT(4, 1);

        // ---- build the symbols of one hazard-free segment from its visited mask ----
        auto build = [&](uint64_t V, uint32_t next_lane) {
            if (V == 0ull) return;
            TSQ_CNT(5, 1);
            {   // a literal run that ended exactly at the segment boundary: its pending bytes close here
                const uint32_t first_pos = base + lsb64(V);
                if (((V & certain_m) >> lsb64(V)) & 1ull) {
                    if (lit_from < first_pos) { push(rec_literal(lit_from, first_pos - lit_from), first_pos); lit_from = first_pos; }
                }
            }
            const uint32_t nsym_entry = nsym, origin_entry = origin;
            const uint64_t M = V & certain_m, N = V & ~certain_m;
            const uint32_t Ls = lsb64(V), Le = msb64(V);
            const uint32_t len_first = ((N >> Ls) & 1ull) ? ones_from(N, Ls) : 0u;      // N lanes contiguous from Ls
            const uint64_t startN = N & ~(N << 1);
            const bool isM = (M >> lane) & 1ull, isN = (N >> lane) & 1ull;
            const bool in_first = lane >= Ls && lane < Ls + len_first;
            const uint64_t sb = startN & below(lane + 1u);
            const uint32_t rs_lane = sb ? msb64(sb) : 0u;
            const uint32_t rs_pos = in_first ? lit_from : base + rs_lane;                // where this lane's literal run is chunked from
            const uint32_t off = p - rs_pos;
            const bool next_isM = lane < 63u && ((M >> (lane + 1u)) & 1ull);
            const bool ownerN = isN && ((off & 15u) == 15u || next_isM);                 // last byte of a 16-chunk, or of a closed run
            const bool sym = isM || ownerN;
            const uint64_t SS = __ballot(sym);
            const uint64_t before = SS & below(lane);
            const uint32_t idx = nsym + (uint32_t)__builtin_popcountll(before);
            const uint32_t sym_start = isM ? p : p - (off & 15u);
            const uint32_t prev_start_v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((before ? msb64(before) : 0u) << 2), (int)sym_start);
            const uint32_t pair_origin = (idx & 1u) ? (before ? prev_start_v : origin_entry) : sym_start;
#ifdef TSQ_STATS
            if (b == 0 && lane == 0 && st_[10] < 400) {
                uint32_t* d = g_dbg_syms + 4096 + st_[10] * 10; st_[10]++;
                d[0] = base; d[1] = (uint32_t)V; d[2] = (uint32_t)(V >> 32); d[3] = (uint32_t)SS; d[4] = (uint32_t)(SS >> 32);
                d[5] = nsym_entry; d[6] = lit_from; d[7] = (uint32_t)after_match | (next_lane << 8); d[8] = (uint32_t)certain_m; d[9] = (uint32_t)(certain_m >> 32);
            }
#endif
            if (sym) ring[idx & (kOrbRing - 1u)] = isM ? rec_match(pair_origin - cand0, nib) : rec_literal(sym_start, (off & 15u) + 1u);

            // scalar state after the segment
            const uint32_t cnt = (uint32_t)__builtin_popcountll(SS);
            if (cnt) {
                const uint32_t ls = msb64(SS);
                nsym += cnt;
                const uint32_t last_start = rdlane(sym_start, ls);
                const uint32_t last_end = base + ls + (((M >> ls) & 1ull) ? rdlane(span_nat, ls) : 1u);
                origin = (nsym & 1u) ? last_start : last_end;
            }
            if ((M >> Le) & 1ull) {
                after_match = true;
                lit_from = base + next_lane;
            } else {
                const uint32_t off_e = rdlane(off, Le);
                const bool own_e = (SS >> Le) & 1ull;                                    // Le is an N lane: a symbol there means it owns one
                lit_from = own_e ? base + Le + 1u : base + Le - (off_e & 15u);
                if (Le < Ls + len_first) {
                    if (after_match) { run0 = base + Ls; origin_r0 = origin_entry; odd_r0 = nsym_entry & 1u; }
                } else {
                    const uint32_t rsl = rdlane(rs_lane, Le);
                    const uint64_t bef = SS & below(rsl);
                    const uint32_t ns0 = nsym_entry + (uint32_t)__builtin_popcountll(bef);
                    run0 = base + rsl;
                    odd_r0 = ns0 & 1u;
                    origin_r0 = odd_r0 ? rdlane(sym_start, msb64(bef)) : run0;
                }
                after_match = false;
            }
            if ((nsym_entry ^ nsym) & ~63u) flush_batch((nsym & ~63u) - 64u, 64u);
        };

        // ---- exact scalar handling of one visit (hazard lanes) ----
        auto visit_serial = [&](uint32_t L) {
            TSQ_CNT(7, 1);
            const uint32_t i = base + L;
            uint32_t cand = rdlane(cand0, L);
            uint32_t k = rdlane(k0, L);
            bool e4 = k >= 4u;
            if ((flagged >> L) & 1ull) {
                const uint64_t grp = __ballot(h == rdlane(h, L)) & vall & below(L);
                if (grp) {                                  // an earlier lane with the same hash was visited: it is the candidate
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
                // rep_last_i as the reference has it at this visit: forced flushes of the run counted, nothing else
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
            // one match attempt (tsq_encode.cpp:125-159)
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

        // ------------------------------------------------------------------ orbit over the window
        uint32_t L = v - base;
        uint64_t V = 0;
        while (!done && !overflow) {
            bool window_end = false;
            for (;;) {
                if (L >= 64u) { window_end = true; break; }
                const uint32_t s = rdlane(span, L);
                if (s != 0u) { V |= 1ull << L; L += s; continue; }
                TSQ_CNT(6, 1);
                if (!((hard >> L) & 1ull)) {
                    // flagged only: harmless unless an earlier lane with the same hash was visited
                    const uint64_t grp = __ballot(h == rdlane(h, L)) & (vall | V) & below(L);
                    if (grp == 0ull) { V |= 1ull << L; L += rdlane(span_nat, L); continue; }
                }
                break;
            }
            TSQ_ACC(1);
            build(V, L);
            TSQ_ACC(3);
            vall |= V;
            V = 0;
            if (window_end) { v = base + L; break; }
            visit_serial(L);
            TSQ_ACC(8);
            L = v - base;
        }
        TSQ_ACC(1);
        if (done || overflow) break;

        // ------------------------------------------------------------------ commit the window
        if (((vall & ~flagged) >> lane) & 1ull) table[h] = (uint16_t)p;
        uint64_t late = vall & flagged;
        while (late) {
            if (lane == lsb64(late)) table[h] = (uint16_t)p;
            late &= late - 1ull;
        }
        TSQ_ACC(2);
    }

    if (overflow) { if (lane == 0) { atomicMax(status, kErrOverflow); sizes[b] = 3; } return; }

    // ---- tail (tsq_encode.cpp:176-188) ----
    const uint32_t rest = nsym & 63u;
    if (rest) flush_batch(nsym - rest, rest);
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
#ifdef TSQ_STATS
    if (b == 0 && lane == 0) { st_[9] = nsym; for (int q = 0; q < 16; ++q) g_enc_stats[q] = st_[q]; }
#endif
}

}  // namespace tsq
