// tsq_enc_fast.cuh -- wave-window block encoder for gfx950 (A/B variant 2: superseded, not in the product library).
//
// The reference parse (tsq_encode.cpp:48-342) is a greedy walk in which every decision depends on
// the hash table, and the table depends on which positions the walk visited -- the walk itself is
// inherently serial and bit-exactness forbids a different parse.  What is
// NOT serial is everything around it.  One wavefront per block, three roles for its 64 lanes:
//
//   window   64 consecutive positions at a time: lane L owns position base+L and gathers
//            w = in[p..p+16)   h = hash(w)   t = table[h]   cand   c = in[cand..cand+16)
//            k = common prefix (<= 16)  ->  per-lane candidate + length registers.
//   walk     wave-uniform (scalar) control flow over the window.  A visit is one v_readlane plus a
//            few scalar ops; literal runs are skipped with a ballot + find-first-bit instead of
//            position by position.  The walk keeps almost no state: it only appends 32-bit
//            symbol records to a register (lane = symbol index mod 64).
//   emit     every 64 symbols the lanes lay the stream out in parallel: a wave prefix sum gives each
//            symbol its output offset (payload + one control byte per 8 + one size byte per 2
//            symbols), control bytes come from a ballot, size bytes from a lane shuffle, literal
//            payloads are copied 16 bytes per lane.  This replaces the reference's
//            read-modify-write of control/size bytes in output memory (tsq_encode.cpp:94-95).
//
// Table coherence inside a window: the gathered candidates reflect the table at window start, so
// a lane whose hash equals that of an EARLIER lane of the same window ("flagged") may need the
// earlier lane as its candidate instead, if the walk visited it.  Flagged lanes are found with one
// LDS write + read per lane (a byte-per-bucket image of the table's index space) and are resolved
// at visit time from the visited mask; everything else runs on the precomputed registers.
#pragma once

#include "../tsq_common.cuh"
#include "../tsq_enc_util.cuh"
#include "../tsq_enc_builder.cuh"

namespace tsq {

constexpr uint32_t kEncLds = kHashEntries;          // u8 per bucket: which lane wrote it in this window

template <bool EXT>
__global__ __launch_bounds__(64) void enc_fast_kernel(const uint8_t* __restrict__ in, uint64_t n_total, uint64_t readable,
                                                      uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                      uint16_t* __restrict__ tables, int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t bucket_owner_lds[];   // kEncLds bytes
    volatile uint8_t* bucket_owner = bucket_owner_lds;   // volatile: the read-back must not be forwarded from the store
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint64_t start = (uint64_t)b << kBlockBits;
    const uint64_t avail = readable - start;
    const uint32_t n = n_total - start < kBlockSize ? (uint32_t)(n_total - start) : kBlockSize;
    const uint8_t* src = in + start;
    uint8_t* out = slots + (size_t)b * kSlotSize;
    uint16_t* table = tables + (size_t)b * kHashEntries;

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

    // ---- walk state (all wave-uniform) ----
    uint32_t v = 1;                // next position to visit; position 0 is never probed (tsq_encode.cpp:72)
    uint32_t pending = 0;          // first input byte not yet emitted as a symbol (last_i)
    uint32_t origin = 0;           // input position at the start of the current pair (rep_last_i)
    uint32_t nsym = 0;
    bool in_chain = false;         // the next visit is the probe that follows a match (tsq_encode.cpp:162-170)
    uint32_t rec = 0;              // per lane: symbol record nsym % 64
    // ---- emit state ----
    uint32_t j0 = 3;               // output position of the next batch
    uint32_t lit_out = 0xFFFFFFFFu, lit_src = 0;
    bool overflow = false;

    auto flush_batch = [&](uint32_t cnt) {
        const EmitResult r = emit_batch(rec, cnt, j0, out, src, avail, lane);
        j0 = uniform(r.end);                                       // a call's results are not provably uniform
        const uint32_t lo = uniform(r.lit_out);
        if (lo != 0xFFFFFFFFu) { lit_out = lo; lit_src = uniform(r.lit_src); }
        if (j0 + 1200u > kSlotSize) overflow = true;
    };
    // append one symbol; every 64 symbols the batch goes out
    auto push = [&](uint32_t record, uint32_t origin_if_pair_closes) {
        rec = lane == (nsym & 63u) ? record : rec;          // v_cmp + v_cndmask: the record lands in lane nsym % 64
        nsym++;
        if ((nsym & 1u) == 0u) origin = origin_if_pair_closes;
        if ((nsym & 63u) == 0u) flush_batch(64u);
    };
    // literal symbols for [pending, to): chunks of at most 16 bytes (tsq_encode.cpp:85-97,105-117)
    auto literals = [&](uint32_t to) {
        while (pending != to) {
            const uint32_t len = to - pending > 16u ? 16u : to - pending;
            const uint32_t at = pending;
            pending += len;
            push(rec_literal(at, len), pending);
        }
    };

    bool first = true;
    bool done = false;
    while (!done && !overflow) {
        // ------------------------------------------------------------------ load the window
        const uint32_t base = first ? 0u : v;
        first = false;
        const uint32_t wend = base + 64u;
        const uint32_t p = base + lane;
        const uint4 w16 = ld128z(src, p, avail);
        const uint32_t w = w16.x;
        const uint32_t h = hash4(w);
        bucket_owner[h] = (uint8_t)lane;
        const uint32_t t = table[h];
        const uint32_t cand0 = candidate_of(t, p);
        const uint4 c16 = ld128z(src, cand0, avail);
        const uint32_t k0 = prefix16(w16, c16);
        // lanes whose bucket was also written by another lane of this window
        uint64_t shared = __ballot(bucket_owner[h] != (uint8_t)lane);
        uint64_t flagged = 0;      // lanes that have an EARLIER lane with the same hash
        while (shared) {
            const uint32_t l = (uint32_t)__builtin_ctzll(shared);
            const uint64_t grp = __ballot(h == rdlane(h, l));
            flagged |= grp & (grp - 1ull);                          // all but the lowest lane of the group
            shared &= ~grp;
        }
        const uint64_t has4 = __ballot(k0 >= 4u);
        uint64_t visited = 0;
        TSQ_ACC(0); TSQ_CNT(4, 1); TSQ_CNT(5, flagged ? 1 : 0);

        // ------------------------------------------------------------------ walk the window
        while (v < wend) {
            TSQ_CNT(6, 1);
            uint32_t i;                    // the visit this iteration resolves
            uint32_t cand, k;
            bool eq4;
            if (!in_chain) {
                // ---- scan: positions v, v+1, ... are visited until a match condition holds
                // (tsq_encode.cpp:70-100).  Within one stretch the pair origin is constant, so the
                // condition is evaluated for all lanes at once.
                const uint32_t flush_at = pending + 32u;                                  // i - pending > 31
                const uint64_t cond = __ballot(k0 >= 4u && offset_ok(origin - cand0)) | flagged;
                const uint64_t ahead = cond & ~below(v - base);
                uint32_t stop = ahead ? base + (uint32_t)__builtin_ctzll(ahead) : wend;
                if (stop > n) stop = n;                                                  // position n ends the scan
                if (stop > flush_at) stop = flush_at;
                if (stop >= wend) {                                                      // nothing before the window ends
                    visited |= ~below(v - base);
                    v = wend;
                    break;
                }
                i = stop;
                visited |= below(i - base + 1u) & ~below(v - base);
            } else {
                i = v;                                                                   // the probe after a match
                visited |= 1ull << (i - base);
            }
            {
                const uint32_t L = i - base;
                cand = rdlane(cand0, L);
                k = rdlane(k0, L);
                eq4 = (has4 >> L) & 1ull;
                if ((flagged >> L) & 1ull) {
                    // an earlier lane with the same hash: if the walk visited it, it is the candidate
                    const uint64_t grp = __ballot(h == rdlane(h, L)) & visited & below(L);
                    if (grp) {
                        const uint32_t q = 63u - (uint32_t)__builtin_clzll(grp);
                        cand = base + q;
                        eq4 = rdlane(w, L) == rdlane(w, q);
                        k = 0xFFu;                                                       // not computed yet
                    }
                }
            }
            const bool ok = eq4 && offset_ok(origin - cand);       // offset taken before any flush (tsq_encode.cpp:80-82)
            if (!in_chain) {
                if (i == pending + 32u) literals(i);                                     // tsq_encode.cpp:82-98
                if (i < n && !ok) { v = i + 1u; continue; }                              // tsq_encode.cpp:100
                literals(i);                                                             // tsq_encode.cpp:103-118
                if (!(i < n)) { done = true; break; }                                    // tsq_encode.cpp:120
            } else if (!(i < n - 5u && ok)) {                                            // tsq_encode.cpp:170
                in_chain = false;
                if (!(i < n)) { done = true; break; }                                    // tsq_encode.cpp:173
                pending = i; v = i + 1u;
                continue;
            }

            // ---- one match attempt at i against cand (tsq_encode.cpp:125-159)
            TSQ_CNT(7, 1);
            if (k == 0xFFu) k = uniform(prefix16(ld128z(src, i, avail), ld128z(src, cand, avail)));
            if (EXT && k == 16u) {
                uint32_t nb;
                do { nb = uniform(prefix8(ldu64z(src, (uint64_t)i + k, avail), ldu64z(src, (uint64_t)cand + k, avail))); k += nb; }
                while (nb == 8u && k < 64u);
            }
            const uint32_t room = origin - cand;
            if (k > room) k = room - 1u;
            if (k < 4u || !offset_ok(room)) {                    // the chain breaks without a symbol
                in_chain = false;
                pending = i; v = i + 1u;
                continue;
            }
            const uint32_t m = length_nibble(k);
            const uint32_t ni = i + nibble_span(m);
            push(rec_match(room, m), ni);
            in_chain = true;
            v = ni;
            pending = ni;
            if (overflow) break;
        }
        TSQ_ACC(1);
        if (done || overflow) break;

        // ------------------------------------------------------------------ commit the window
        // visited lanes record their positions; among equal hashes the highest visited lane must
        // win, so the (few) flagged lanes store afterwards in ascending order
        if (((visited & ~flagged) >> lane) & 1ull) table[h] = (uint16_t)p;
        uint64_t late = visited & flagged;
        while (late) {
            const uint32_t l = (uint32_t)__builtin_ctzll(late);
            if (lane == l) table[h] = (uint16_t)p;
            late &= late - 1ull;
        }
        TSQ_ACC(2);
    }

    if (overflow) { if (lane == 0) { atomicMax(status, kErrOverflow); sizes[b] = 3; } return; }

    // ---- tail (tsq_encode.cpp:176-188) ----
    const uint32_t rest = nsym & 63u;
    if (rest) flush_batch(rest);
    // bytes the reference allocates for the NEXT symbol but never fills: they hold the spill of the
    // last 16-byte literal store, or the zero the buffer was filled with
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
