// ab/tsq_enc_stage5.cuh -- round 2's five-wave staged block encoder, frozen for A/B measurements (kernel variant 5 of the A/B library).
// Not part of the product: compiled only with -DTSQ_AB_VARIANTS (make ab).  The text below is the round-2 state of
// tsq_enc_builder.cuh + tsq_enc_stage.cuh inside namespace tsq::r02; the production encoder (eleven wavefronts) is tsq_enc_stage.cuh.
#pragma once

#include "../tsq_common.cuh"
#include "../tsq_enc_util.cuh"

namespace tsq {
namespace r02 {
// tsq_enc_builder.cuh -- the stream BUILDER wave of the staged encoder and its batch layout.
//
// The parser hands over, through a single-producer/single-consumer queue in LDS, one item per hazard-free
// segment of a tile (visited mask + which visited lanes are matches + per-lane candidate|nibble words + the parse
// state at segment entry) or per explicitly pushed symbol.  The builder derives the symbols in parallel -- matches =
// visited match lanes, literals = 16-byte chunks of visited literal runs, symbol indices by popcount, pair origin of
// an odd symbol = start of the previous one -- and every 64 symbols lays the stream out (emit_batch): control bytes
// from a ballot, size bytes from a shuffle, payload offsets from a prefix sum (tsq_encode.cpp:57-59,94-95,103-118,
// 152-159).  The never-filled trailing control/size bytes get the reference's stale values (tsq_encode.cpp:176-188).



// Lay out and store the `cnt` (<= 64) symbols held one per lane in `rec`, starting at output
// position j0 which is the start of a group of 8 (tsq_encode.cpp:57-59,94-95: control byte, then per
// pair a size byte and the two payloads).  Returns the output position after the last payload.
// lit_out / lit_src report the last literal chunk of the batch (for the never-filled trailing
// bytes); lit_out == 0xFFFFFFFF when the batch holds no literal.  Not inlined: it runs once per
// 64 symbols and must not bloat the walk loop; the caller re-uniforms the results.
struct EmitResult { uint32_t end, lit_out, lit_src; };
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

template <class Cfg>
__device__ __forceinline__ void stream_builder(const uint8_t* src, uint64_t avail, uint8_t* out, lds_u8_t* lds, uint32_t lane,
                                             uint32_t b, uint32_t* sizes, int32_t* status)
{
        volatile lds_u32_t* queue = (volatile lds_u32_t*)(lds + Cfg::off_queue);
    volatile lds_u32_t* ring = (volatile lds_u32_t*)(lds + Cfg::off_ring);
    lds_u32_t* ctl = (lds_u32_t*)(lds + Cfg::off_ctl);
    uint32_t tail = 0, nsym = 0, j0 = 3, lit_out = 0xFFFFFFFFu, lit_src = 0;
    bool overflow = false;

    auto flush_batch = [&](uint32_t first_index, uint32_t cnt) {
        if (overflow) return;
        const uint32_t rec = ring[(first_index + lane) & (Cfg::RING - 1u)];
        const EmitResult r = emit_batch(rec, cnt, j0, out, src, avail, lane);
        j0 = uniform(r.end);
        const uint32_t lo = uniform(r.lit_out);
        if (lo != 0xFFFFFFFFu) { lit_out = lo; lit_src = uniform(r.lit_src); }
        if (j0 + 1200u > kSlotSize) overflow = true;
    };

#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
#ifdef TSQ_STATS
    const unsigned long long begin_ = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
        if (__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == tail) {
#ifdef TSQ_STATS
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            while (__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == tail) __builtin_amdgcn_s_sleep(1);
#ifdef TSQ_STATS
            st_[17] += __builtin_amdgcn_s_memtime() - w0_;
#endif
        }
        volatile lds_u32_t* it = queue + (tail % Cfg::Q) * Cfg::ITEM_WORDS;
        const uint32_t kind = uniform(it[0]);
        const uint32_t nsym_entry = uniform(it[4]);
        uint32_t nsym_after = nsym_entry;
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
            if (sym) ring[idx & (Cfg::RING - 1u)] = isM ? rec_match(pair_origin - cand0, nib) : rec_literal(sym_start, (off & 15u) + 1u);
            nsym_after = idx0 + (uint32_t)__builtin_popcountll(SS);
        }
        // the item is consumed: release the slot before the (long) flush
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tail++;
        __hip_atomic_store(&ctl[1], tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (kind == kItemEnd) { nsym = nsym_entry; break; }
        if ((nsym ^ nsym_after) & ~63u) flush_batch((nsym_after & ~63u) - 64u, 64u);
        nsym = nsym_after;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[17] = st_[17]; g_enc_stats[18] = __builtin_amdgcn_s_memtime() - begin_; }
#endif

    if (overflow) { if (lane == 0) { atomicMax(status, kErrOverflow); sizes[b] = 3; } return; }
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
}


// tsq_enc_stage.cuh -- five-wave staged block encoder for gfx950 (kernel variant 0).
//
// A single wavefront issues about one instruction every five cycles, and the greedy parse of a block is
// serial (tsq_encode.cpp:72-187: the position table is a function of the parse).  So the block's work
// is cut into stages, one wavefront each, that stream 64-position tiles through records in LDS:
//
//   wave 0  SCAN     input words, hashes and the same-hash ("twin") masks of every tile -- everything
//                    that does not depend on the parse; runs ahead as far as the record ring allows.
//   wave 1  MATCH    the position table: commits the visited positions of tile t-3 (handed back by the
//                    parser), gathers the candidates of tile t, their bytes, the common prefixes, and
//                    classifies the lanes (certain match / certain literal / hazard).
//   wave 2  ORBIT    the visited set from every possible entry lane of the tile, by pointer doubling.
//   wave 3  PARSER   the serial part: picks the orbit of the actual entry lane, checks the twins it
//                    visited, resolves hazards with exact scalar code, keeps the pair state.
//   wave 4  BUILDER  symbol records and stream layout (tsq_enc_builder.cuh: stream_builder).
//
// Table lag.  MATCH gathers tile t from a table that holds exactly the visits of tiles <= t-3 (it does
// the commits itself, in program order), so parser and MATCH overlap over two tiles.  What the table
// cannot know -- a visited position of tiles t-2, t-1 or an earlier lane of t with the same hash -- is a
// twin: SCAN finds all of them exactly (byte-per-bucket owner image in LDS, folded to 16 bits, with
// exact hash comparison by ballot), and the parser takes the most recent VISITED twin as the
// candidate, which is what the reference's table would hold (tsq_encode.cpp:76-79), or keeps the
// gathered candidate when no twin was visited.



typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;

struct StageCfg {
    static constexpr uint32_t Q = 32;
    static constexpr uint32_t ITEM_WORDS = 80;
    static constexpr uint32_t RING = 128;
    static constexpr uint32_t R = 8;                              // tile records in flight
    static constexpr uint32_t OWN_MASK = 0x7FFFu;                 // owner image: hash folded to 15 bits
    static constexpr uint32_t WIN = 73728;                        // input window ring: the last 64 KiB of input and then some (a multiple of 64)
    static constexpr uint32_t F_MASK = 0xFFFu;                    // MATCH's filter of tile t-3's visited lanes: hash folded to 12 bits
    static constexpr uint32_t W16 = 16;                           // word offset of the uint4-per-lane input words
    static constexpr uint32_t ARR = 16 + 256;                     // word offset of the u32-per-lane arrays
    static constexpr uint32_t REC_WORDS = ARR + 12 * 64;
    static constexpr uint32_t off_owner = 0;                                   // u8[65536]
    static constexpr uint32_t off_queue = OWN_MASK + 1u;                       // u32[Q * ITEM_WORDS]
    static constexpr uint32_t off_ring = off_queue + Q * ITEM_WORDS * 4;       // u32[RING]
    static constexpr uint32_t off_rec = off_ring + RING * 4;                   // u32[R * REC_WORDS]
    static constexpr uint32_t off_ctl = off_rec + R * REC_WORDS * 4;           // u32[64]
    static constexpr uint32_t off_f = off_ctl + 256;                           // u8[F_MASK + 1]
    static constexpr uint32_t off_win = off_f + F_MASK + 1u;                   // u8[WIN + 32]: the ring, its first 32 bytes mirrored at the end
    static constexpr uint32_t total = off_win + WIN + 32;
    static constexpr uint32_t total_lean = off_win;                            // without the window: two blocks fit one CU
};
// ctl words: 0 queue head, 1 queue tail, 2 tiles scanned, 3 tiles matched, 4 tiles with orbits, 5 tiles parsed,
//            6 stop, 16 + 2*(t&7): visited mask of tile t (lo, hi)
// record: header words 0,1 = lanes that have an earlier twin inside the tile
//         per-lane arrays: 0 hash  1,2 twins in this tile (earlier lanes)  3,4 twins in tile t-1  5,6 twins in tile t-2
//                          7 spanword  8 candidate | nibble << 24  9 orbit halt  10,11 orbit mask
// spanword: natural span (bits 0..7) | hard (8) | has a twin (9) | certain match (10) | near twin (11) | common prefix (16..23)
enum : uint32_t { kAH = 0, kATin = 1, kATp1 = 3, kATp2 = 5, kASpan = 7, kALane = 8, kANx = 9, kAOrb = 10 };

// Instrumented builds time only the spin loops (and only when they actually spin): s_memtime costs a few
// hundred cycles, so finer timing distorts the pipeline it measures.  busy = total - waited.

// Consuming a record: the counter is read first, the record's words after it.  The LDS executes the DS operations of a
// wavefront in program order (see stage_publish), so only the compiler has to be kept from hoisting record loads above
// the counter load: the barrier below is the acquire half of the handshake at compiler level.
__device__ __forceinline__ bool stage_ready(lds_u32_t* ctl, uint32_t word, uint32_t need)
{
    const uint32_t seen = uniform(__hip_atomic_load(&ctl[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    asm volatile("" ::: "memory");
    return seen >= need;
}
__device__ __forceinline__ bool stage_spin(lds_u32_t* ctl, uint32_t word, uint32_t need)
{
    for (;;) {
        if (stage_ready(ctl, word, need)) return true;
        if (uniform(__hip_atomic_load(&ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}
#define r02_stage_wait(ctl, word, need, slot) (stage_ready(ctl, word, need) || stage_spin(ctl, word, need))
// Publishing a record: the DS operations of one wavefront are executed by the LDS in program order, so a
// counter stored after the record's words becomes visible after them; only the compiler has to be kept
// from reordering (an s_waitcnt here would park the wave for a full LDS round trip per tile).
__device__ __forceinline__ void stage_publish(lds_u32_t* ctl, uint32_t word, uint32_t value, uint32_t lane)
{
    TSQ_LDS_RELEASE();
    if (lane == 0) __hip_atomic_store(&ctl[word], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------------------------------------- SCAN
template <bool WINDOW>
__device__ __forceinline__ void stage_scan(const uint8_t* src, uint64_t avail, uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    volatile lds_u8_t* owner = lds + StageCfg::off_owner;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    const uint32_t n_tiles = (n >> 6) + 3u;            // visits reach at most n + 63
    uint32_t h_m1 = 0xFFFFFFFFu, h_m2 = 0xFFFFFFFFu, h_m3 = 0xFFFFFFFFu;   // hashes of tiles t-1 .. t-3 (per lane)
    uint32_t id = 1;                                    // (t % 3) + 1: which of the three live tiles an owner tag names
    uint32_t wbase = 0;                                 // (t * 64) % WIN
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    uint4 w_next = ld128z(src, lane, avail);
    for (uint32_t t = 0; t < n_tiles; ++t) {
        // the slot of tile t-R is free once the parser has finished tile t-R+2 (it reads the words of two tiles back)
        if (t + 3u > StageCfg::R && !r02_stage_wait(ctl, 5, t + 3u - StageCfg::R, 0)) break;
        const uint32_t p = (t << 6) + lane;
        const uint4 w16 = w_next;
        w_next = ld128z(src, (uint64_t)p + 64u, avail);    // the next tile's words: this wave's only global access, a full iteration ahead
        const uint32_t h = hash4(w16.x);
        const uint32_t hf = h & StageCfg::OWN_MASK;
        const uint32_t tag = (id << 6) | lane;          // never 0: the image starts zeroed
        // retire the entries of tile t-3 (same id as tile t) unless a later tile has taken the bucket over
        if (t >= 3u && ((uint32_t)owner[h_m3 & StageCfg::OWN_MASK] >> 6) == id) owner[h_m3 & StageCfg::OWN_MASK] = 0;
        const uint32_t before = owner[hf];              // non-zero: a lane of tile t-1 or t-2 may have this hash
        owner[hf] = (uint8_t)tag;
        // twins inside the tile: for each lane the mask of EARLIER lanes with the same hash
        uint64_t twin_in = 0, twins_here = 0;
        {
            uint64_t shared = __ballot(owner[hf] != (uint8_t)tag);
            while (shared) {
                const uint32_t hl = rdlane(h, lsb64(shared));
                const uint64_t grp = __ballot(h == hl);
                if (h == hl) twin_in = grp & below(lane);
                twins_here |= grp & (grp - 1ull);
                shared &= ~grp;
            }
        }
        // twins in the two previous tiles (tile t-3 and older are MATCH's business: by then the parser has decided them)
        uint64_t twin_p1 = 0, twin_p2 = 0;
        {
            uint64_t maybe = __ballot(before != 0u);
            TSQ_CNT(20, __builtin_popcountll(maybe));
            while (maybe) {
                TSQ_CNT(21, 1);
                const uint32_t hl = rdlane(h, lsb64(maybe));
                const uint64_t g1 = __ballot(h_m1 == hl), g2 = __ballot(h_m2 == hl);
                const uint64_t grp_cur = __ballot(h == hl);
                if (h == hl) { twin_p1 = g1; twin_p2 = g2; }
                maybe &= ~grp_cur;
            }
        }
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        {
            u32x4_t v; v.x = w16.x; v.y = w16.y; v.z = w16.z; v.w = w16.w;
            *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u) = v;
            // the tile's 64 input bytes join the window ring MATCH takes candidate bytes from
            if (WINDOW && (lane & 15u) == 0u) {
                *(volatile lds_u32x4_t*)(lds + StageCfg::off_win + wbase + lane) = v;
                if (wbase == 0u && lane < 32u) *(volatile lds_u32x4_t*)(lds + StageCfg::off_win + StageCfg::WIN + lane) = v;
            }
        }
        if (lane == 0) { rec[0] = (uint32_t)twins_here; rec[1] = (uint32_t)(twins_here >> 32); }
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane;
        arr[kAH * 64] = h;
        arr[kATin * 64] = (uint32_t)twin_in;  arr[(kATin + 1) * 64] = (uint32_t)(twin_in >> 32);
        arr[kATp1 * 64] = (uint32_t)twin_p1;  arr[(kATp1 + 1) * 64] = (uint32_t)(twin_p1 >> 32);
        arr[kATp2 * 64] = (uint32_t)twin_p2;  arr[(kATp2 + 1) * 64] = (uint32_t)(twin_p2 >> 32);
        stage_publish(ctl, 2, t + 1u, lane);
        h_m3 = h_m2; h_m2 = h_m1; h_m1 = h;
        id = id == 3u ? 1u : id + 1u;
        wbase = wbase + 64u == StageCfg::WIN ? 0u : wbase + 64u;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[0] = st_[0]; g_enc_stats[1] = TSQ_TOTAL(); g_enc_stats[32] = st_[20]; g_enc_stats[33] = st_[21]; g_enc_stats[34] = st_[20]; }
#endif
}

// --------------------------------------------------------------------------------------------- MATCH
template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_match(const uint8_t* src, uint64_t avail, uint32_t n, uint16_t* table, lds_u8_t* lds, uint32_t lane)
{
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    volatile lds_u8_t* filt = lds + StageCfg::off_f;
    constexpr uint32_t kDMin = EXT ? 128u : 64u;
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;
    const uint32_t n_tiles = (n >> 6) + 3u;
    uint32_t h_m1 = 0, h_m2 = 0, h_m3 = 0;
    uint64_t tw_m1 = 0, tw_m2 = 0, tw_m3 = 0;          // "has an earlier twin inside its tile" masks of those tiles
    uint32_t tin1_lo = 0, tin1_hi = 0, tin2_lo = 0, tin2_hi = 0, tin3_lo = 0, tin3_hi = 0;   // earlier-twin masks of the lanes of those tiles
    uint32_t wbase = 0;                                // (t * 64) % WIN
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = 0; t < n_tiles; ++t) {
        MREG_BEGIN(10);
        if (!r02_stage_wait(ctl, 2, t + 1u, 2)) break;
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane;
        const uint32_t h = arr[kAH * 64];
        const u32x4_t wv = *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u);
        const uint4 w16 = make_uint4(wv.x, wv.y, wv.z, wv.w);
        const uint64_t twin_in = (uint64_t)arr[kATin * 64] | ((uint64_t)arr[(kATin + 1) * 64] << 32);
        const uint64_t twin_p1 = (uint64_t)arr[kATp1 * 64] | ((uint64_t)arr[(kATp1 + 1) * 64] << 32);
        const uint32_t tp2_any = arr[kATp2 * 64] | arr[(kATp2 + 1) * 64];
        const uint64_t twins_here = (uint64_t)uniform(rec[0]) | ((uint64_t)uniform(rec[1]) << 32);
        MREG_END(10);
        MREG_BEGIN(12);
        // ---- the table holds the visits of tiles <= t-4 (commit(t-4) was issued at the end of the previous iteration):
        //      gather from it right away, without waiting for the parser ...
        const uint32_t tv_old = table[h];
        MREG_END(12);
        MREG_BEGIN(11);
        // ... and bring the entries up to "visits of tiles <= t-3" once the parser has finished tile t-3: a lane with a
        // visited twin there takes the most recent one (what the committed table would hold), the others keep theirs.
        // The visited lanes of tile t-3 post themselves in a small filter (hash folded to 12 bits, the highest lane of a
        // hash last, like the commit); a lane of tile t that finds its own hash there has found its most recent visited
        // twin; one that finds another hash (a fold collision, rare) is settled with ballots.
        uint32_t tv = tv_old;
        if (t >= 3u) {
            if (!r02_stage_wait(ctl, 5, t - 2u, 3)) break;
            const uint32_t slot = 16u + 2u * ((t - 3u) & 7u);
            const uint64_t vis = (uint64_t)uniform(__hip_atomic_load(&ctl[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) |
                                 ((uint64_t)uniform(__hip_atomic_load(&ctl[slot + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) << 32);
            const uint32_t p3 = ((t - 3u) << 6) + lane;
            const bool mine = (vis >> lane) & 1ull;
            const uint32_t f3 = h_m3 & StageCfg::F_MASK;
            // among equal hashes the highest visited lane must win: lanes with an earlier twin store afterwards ...
            if (((vis & ~tw_m3) >> lane) & 1ull) { table[h_m3] = (uint16_t)p3; filt[f3] = (uint8_t)(0x80u | lane); }
            // ... one store per hash group: the highest visited lane of a group stores, its earlier twins are dropped unseen
            // (a block of equal bytes is ONE group of 64 lanes)
            uint64_t late = vis & tw_m3;
            while (late) {
                const uint32_t top = msb64(late);
                if (lane == top) { table[h_m3] = (uint16_t)p3; filt[f3] = (uint8_t)(0x80u | lane); }
                late &= ~((uint64_t)rdlane(tin3_lo, top) | ((uint64_t)rdlane(tin3_hi, top) << 32) | (1ull << top));
            }
            const uint32_t seen = filt[h & StageCfg::F_MASK];
            const uint32_t q = seen & 63u;
            const uint32_t hq = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q << 2), (int)h_m3);
            if (seen != 0u && hq == h) tv = (((t - 3u) << 6) + q) & 0xFFFFu;
            uint64_t unsure = __ballot(seen != 0u && hq != h);
            while (unsure) {                                  // another hash owns the filter slot: this hash's twins, exactly
                const uint32_t hl = rdlane(h, lsb64(unsure));
                const uint64_t g3 = __ballot(h_m3 == hl) & vis;
                if (h == hl && g3 != 0ull) tv = (((t - 3u) << 6) + msb64(g3)) & 0xFFFFu;
                unsure &= ~__ballot(h == hl);
            }
            if (mine) filt[f3] = 0;                           // the filter only ever holds one tile
        }
        MREG_END(11);
        // ---- candidates of tile t
        const uint32_t p = (t << 6) + lane;
        const uint32_t cand0 = candidate_of(tv, p);
        MREG_BEGIN(13);
        // their 16 bytes come from the window ring in LDS (SCAN has written everything below (t+1)*64); the few lanes whose
        // candidate ends beyond that (closer than 19 bytes to the tile's end) gather from global memory
        uint4 cb;
        if (WINDOW) {
            int32_t wi = (int32_t)(wbase + lane) - (int32_t)(p - cand0);
            wi += wi < 0 ? (int32_t)StageCfg::WIN : 0;
            volatile lds_u32_t* wp = (volatile lds_u32_t*)(lds + StageCfg::off_win + ((uint32_t)wi & ~3u));
            const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
            const uint32_t sh = (uint32_t)wi & 3u;
            cb = make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                            __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh));
            if (cand0 + 19u > ((t + 1u) << 6) || p - cand0 > 65536u) cb = ld128z(src, cand0, avail);
        } else cb = ld128z(src, cand0, avail);          // the lean layout (two blocks per CU) has no window: gather from L2
        uint32_t k0 = prefix16(w16, cb);
        MREG_END(13);
        MREG_BEGIN(14);
        if (EXT) {
            // matches longer than 16 (tsq_encode.cpp:280-290): the next 16 bytes of both sides, up to 64.  They come from the window ring
            // too when SCAN has already put the bytes up to p + 80 there (it usually runs two or three tiles ahead of MATCH); from global memory otherwise.
            uint32_t more = 16;
            if (__ballot(k0 == more) != 0ull) {
                const uint32_t scanned = uniform(__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const bool in_window = WINDOW && scanned >= t + 2u;        // everything below (t + 2) * 64 >= p + 48 + 16 is in the ring
                auto win16 = [&](int32_t wi) -> uint4 {
                    wi += wi < 0 ? (int32_t)StageCfg::WIN : 0;
                    wi -= wi >= (int32_t)StageCfg::WIN ? (int32_t)StageCfg::WIN : 0;
                    volatile lds_u32_t* wp = (volatile lds_u32_t*)(lds + StageCfg::off_win + ((uint32_t)wi & ~3u));
                    const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
                    const uint32_t sh = (uint32_t)wi & 3u;
                    return make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                                      __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh));
                };
                while (__ballot(k0 == more) != 0ull && more < 64u) {
                    TSQ_CNT(20, in_window ? 1 : 0); TSQ_CNT(21, 1);
                    if (k0 == more) {
                        if (in_window && p - cand0 <= 65536u)
                            k0 += prefix16(win16((int32_t)(wbase + lane + more)), win16((int32_t)(wbase + lane + more) - (int32_t)(p - cand0)));
                        else
                            k0 += prefix16(ld128z(src, (uint64_t)p + more, avail), ld128z(src, (uint64_t)cand0 + more, avail));
                    }
                    more += 16;
                }
            }
        }
        const uint32_t dist = p - cand0;
        const bool eq4 = k0 >= 4u;
        const bool far_enough = dist >= kDMin && dist <= 0xFFFEu;
        const bool tail = p >= tail_from;
        // a twin at most 3 positions back (runs of equal bytes): if it is visited it becomes the candidate and,
        // being closer than 4, can never match.  Such lanes are classed "no match" optimistically; the parser
        // verifies after the orbit that a near twin was indeed visited (else the lane goes through the exact path).
        bool neart = false;
        if (__ballot((twin_in | twin_p1) != 0ull) != 0ull) {
            const uint64_t near_in = twin_in & ~below(lane >= 3u ? lane - 3u : 0u);
            const uint64_t near_prev = lane < 3u ? twin_p1 & ~below(61u + lane) : 0ull;
            neart = (near_in | near_prev) != 0ull && !tail;
        }
        const bool certain = eq4 && far_enough && !tail && !neart;
        const uint32_t nib = length_nibble(k0 < 4u ? 4u : k0);
        const uint32_t span_nat = certain ? nibble_span(nib) : 1u;
        // offset = origin - cand <= p - cand: a candidate closer than 4 bytes can never pass (offset-4) < 0xFFFB
        // (tsq_encode.cpp:100), whatever the pair origin: such a lane is a plain "no match", not a hazard
        const bool hard_l = (eq4 && !far_enough && dist >= 4u && !neart) || tail;
        const bool twin_l = (twin_in | twin_p1) != 0ull || tp2_any != 0u;
        arr[kASpan * 64] = span_nat | (hard_l ? 0x100u : 0u) | (twin_l ? 0x200u : 0u) | (certain ? 0x400u : 0u) | (neart ? 0x800u : 0u) | (k0 << 16);
        arr[kALane * 64] = cand0 | (nib << 24);
        stage_publish(ctl, 3, t + 1u, lane);
        MREG_END(14);
        h_m3 = h_m2; h_m2 = h_m1; h_m1 = h;
        tw_m3 = tw_m2; tw_m2 = tw_m1; tw_m1 = twins_here;
        tin3_lo = tin2_lo; tin3_hi = tin2_hi; tin2_lo = tin1_lo; tin2_hi = tin1_hi; tin1_lo = (uint32_t)twin_in; tin1_hi = (uint32_t)(twin_in >> 32);
        wbase = wbase + 64u == StageCfg::WIN ? 0u : wbase + 64u;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[2] = st_[2]; g_enc_stats[3] = st_[3]; g_enc_stats[4] = TSQ_TOTAL(); g_enc_stats[13] = st_[13]; g_enc_stats[14] = st_[14]; g_enc_stats[35] = st_[20]; g_enc_stats[36] = st_[21]; }
#endif
}

// --------------------------------------------------------------------------------------------- ORBIT
template <bool EXT>
__device__ __forceinline__ void stage_orbit(uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    constexpr uint32_t kDMin = EXT ? 128u : 64u;
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;
    const uint32_t n_tiles = (n >> 6) + 3u;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = 0; t < n_tiles; ++t) {
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane;
        // ---- late classification of lanes whose only twins are in tile t-2.  By now the parser has (almost always)
        //      finished that tile, so which of those twins were visited is known: the most recent visited one IS the
        //      candidate (tsq_encode.cpp:76-79), 65..191 bytes back.  Such a lane becomes an ordinary certain lane and
        //      never reaches the parser's scalar path.  (Lanes that also have twins in tile t-1 or t stay hazards.)
        //      All of this needs only SCAN's part of the record, so it runs while MATCH is still gathering tile t.
        uint32_t fix_sw = 0, fix_lw = 0;
        bool fix = false, clear_tp2 = false;
        if (t >= 2u) {
            if (!r02_stage_wait(ctl, 2, t + 1u, 6)) break;
            const uint32_t tp2_lo = arr[kATp2 * 64], tp2_hi = arr[(kATp2 + 1) * 64];
            const uint32_t nearer = arr[kATin * 64] | arr[(kATin + 1) * 64] | arr[kATp1 * 64] | arr[(kATp1 + 1) * 64];
            const uint32_t p = (t << 6) + lane;
            const bool only2 = (tp2_lo | tp2_hi) != 0u && nearer == 0u && p < tail_from;
            if (__ballot(only2) != 0ull) {
                if (!r02_stage_wait(ctl, 5, t - 1u, 5)) break;                   // the parser has finished tile t-2
                const uint32_t slot = 16u + 2u * ((t - 2u) & 7u);
                const uint32_t v2_lo = uniform(__hip_atomic_load(&ctl[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const uint32_t v2_hi = uniform(__hip_atomic_load(&ctl[slot + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const uint32_t hit_lo = tp2_lo & v2_lo, hit_hi = tp2_hi & v2_hi;
                clear_tp2 = only2;                       // visited or not, tile t-2 is settled for this lane
                if (only2 && (hit_lo | hit_hi) != 0u) {
                    const uint32_t q = hit_hi ? 63u - (uint32_t)__builtin_clz(hit_hi) : 31u - (uint32_t)__builtin_clz(hit_lo);
                    const uint32_t cand = (t << 6) - 128u + q;
                    const u32x4_t a = *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u);
                    const u32x4_t b = *(volatile lds_u32x4_t*)(recs + ((t - 2u) % StageCfg::R) * StageCfg::REC_WORDS + StageCfg::W16 + q * 4u);
                    const uint32_t k = prefix16(make_uint4(a.x, a.y, a.z, a.w), make_uint4(b.x, b.y, b.z, b.w));
                    // exact when the outcome cannot depend on the pair origin (as for the table's candidates) and, with
                    // extensions, when the first 16 bytes decide the length; otherwise the lane stays a hazard
                    if (p - cand >= kDMin && !(EXT && k >= 16u)) {
                        const bool eq4 = k >= 4u;
                        const uint32_t nib = length_nibble(eq4 ? k : 4u);
                        fix_sw = (eq4 ? nibble_span(nib) | 0x400u : 1u) | (k << 16);
                        fix_lw = cand | (nib << 24);
                        fix = true;
                    } else clear_tp2 = false;
                }
            }
        }
        if (!r02_stage_wait(ctl, 3, t + 1u, 6)) break;
        uint32_t sw = arr[kASpan * 64];
        if (fix) { sw = fix_sw; arr[kASpan * 64] = fix_sw; arr[kALane * 64] = fix_lw; }
        if (clear_tp2) { arr[kATp2 * 64] = 0; arr[(kATp2 + 1) * 64] = 0; }
        // the whole orbit of every lane, by pointer doubling: `nx` = where the orbit started at this lane halts
        // (lane, or position past the tile: 7 bits | halted: bit 7), `orb` = the lanes it visits before that.
        // A hop halts when it lands on a hard lane or past the tile.  Twin lanes do not halt: the parser takes
        // the orbit optimistically and checks the visited twins afterwards.
        const uint32_t self = lane | ((sw & 0x100u) ? 0x80u : 0u);              // arriving at this lane: halts?
        const uint32_t c = lane + (sw & 0xFFu);                                 // < 128
        const uint32_t there = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((c & 63u) << 2), (int)self);
        uint32_t nx = c >= 64u ? (c | 0x80u) : there;
        uint64_t orb = 1ull << lane;
#pragma unroll
        for (int round = 0; round < 6; ++round) {                               // (testing "all halted" each round costs more than it saves)
            const int at = (int)((nx & 63u) << 2);
            const uint32_t nx2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)nx);
            const uint32_t olo = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)(uint32_t)orb);
            const uint32_t ohi = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)(uint32_t)(orb >> 32));
            if ((nx & 0x80u) == 0u) { nx = nx2; orb |= (uint64_t)olo | ((uint64_t)ohi << 32); }
        }
        arr[kANx * 64] = nx;
        arr[kAOrb * 64] = (uint32_t)orb;
        arr[(kAOrb + 1) * 64] = (uint32_t)(orb >> 32);
        stage_publish(ctl, 4, t + 1u, lane);
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[6] = st_[6] + st_[5]; g_enc_stats[7] = TSQ_TOTAL(); }
#endif
}

// ---- uniform (SGPR) flag arithmetic for the parser wave.  Flags are 0/1 integers and every select is an explicit
//      s_cmp + s_cselect pair: left to itself the compiler keeps uniform booleans as 64-bit lane masks, selects through
//      `s_and_b64 exec` triples and converts them to integers through a VGPR (v_cndmask + v_readfirstlane, ~30 cycles).
__device__ __forceinline__ uint32_t s_sel(uint32_t c, uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(d) : "s"(c), "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint64_t s_sel64(uint32_t c, uint64_t a, uint64_t b)
{ uint64_t d; asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b64 %0, %2, %3" : "=s"(d) : "s"(c), "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_nz64(uint64_t x)
{ uint32_t d; asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(x) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_nz(uint32_t x)
{ uint32_t d; asm("s_min_u32 %0, %1, 1" : "=s"(d) : "s"(x) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_lt(uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_ge(uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_ge_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_eq(uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_eq_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_offset_ok(uint32_t offset)          // tsq_encode.cpp:100
{ uint32_t d; asm("s_add_u32 %0, %1, -4\n\ts_cmp_lt_u32 %0, 0xfffb\n\ts_cselect_b32 %0, 1, 0" : "=&s"(d) : "s"(offset) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_msb64(uint64_t x)                   // x != 0
{ uint32_t d; asm("s_flbit_i32_b64 %0, %1\n\ts_xor_b32 %0, %0, 63" : "=s"(d) : "s"(x) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_lsb64(uint64_t x)                   // x != 0
{ uint32_t d; asm("s_ff1_i32_b64 %0, %1" : "=s"(d) : "s"(x)); return d; }

// -------------------------------------------------------------------------------------------- PARSER
// All parse state lives in uniform 32-bit integers (flags as 0/1, not bool: a bool that crosses a branch
// becomes a 64-bit lane mask and costs VALU round trips), and the common paths are straight-line selects:
// for a single wavefront a uniform branch costs more than the few instructions it skips.
template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_parser(const uint8_t* src, uint64_t avail, uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    volatile lds_u32_t* queue = (volatile lds_u32_t*)(lds + StageCfg::off_queue);
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);

    uint32_t head = 0, tail_seen = 0;  // tail_seen: last value read of the builder's progress (re-read only when the queue looks full)
    uint32_t v = 1, nsym = 0, origin = 0, lit_from = 0;
    uint32_t am = 0;                   // 1 right after a match (tsq_encode.cpp:160-187), 0 inside a literal run
    uint32_t run0 = 0, origin_r0 = 0, odd_r0 = 0;   // where the current literal run started, the pair origin and symbol parity then
    uint32_t done = 0;
    uint64_t vall_p1 = 0, vall_p2 = 0;   // visited lanes of the two previous tiles
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif

    auto slot_begin = [&]() -> volatile lds_u32_t* {
        if (head - tail_seen >= StageCfg::Q) {
#ifdef TSQ_STATS
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            while (head - tail_seen >= StageCfg::Q) {
                tail_seen = uniform(__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (head - tail_seen >= StageCfg::Q) __builtin_amdgcn_s_sleep(2);
            }
#ifdef TSQ_STATS
            st_[9] += __builtin_amdgcn_s_memtime() - w0_;
#endif
        }
        return queue + (head % StageCfg::Q) * StageCfg::ITEM_WORDS;
    };
    auto slot_publish = [&]() {
        TSQ_LDS_RELEASE();
        head++;
        if (lane == 0) __hip_atomic_store(&ctl[0], head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto push = [&](uint32_t record, uint32_t origin_if_pair_closes) {
        volatile lds_u32_t* it = slot_begin();
        if (lane == 0) { it[0] = kItemSym; it[4] = nsym; it[9] = record; }
        slot_publish();
        nsym++;
        if ((nsym & 1u) == 0u) origin = origin_if_pair_closes;
    };
    // the 16 input bytes at a position of tiles t-2 .. t, from the tile records (one LDS address for the whole wave)
    auto words_at = [&](uint32_t pos) -> uint4 {
        const u32x4_t q = *(volatile lds_u32x4_t*)(recs + ((pos >> 6) % StageCfg::R) * StageCfg::REC_WORDS + StageCfg::W16 + (pos & 63u) * 4u);
        return make_uint4(q.x, q.y, q.z, q.w);
    };

    uint32_t wbase = 0;                // (t * 64) % WIN
    TSQ_BEGIN();
    for (uint32_t t = 0; done == 0u; ++t, wbase = wbase + 64u == StageCfg::WIN ? 0u : wbase + 64u) {
        const uint32_t base = t << 6;
        uint64_t vall = 0;
        if (v < base + 64u) {
            // ---- the tile's record
            REG_BEGIN(0); REG_END(0);
            REG_BEGIN(1);
            // (the serial stage polls without sleeping: a wake-up from s_sleep costs it up to 64 cycles per hand-off)
            if (!stage_ready(ctl, 4, t + 1u)) {
#ifdef TSQ_STATS
                const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
                while (!stage_ready(ctl, 4, t + 1u)) {}
#ifdef TSQ_STATS
                st_[8] += __builtin_amdgcn_s_memtime() - w0_;
#endif
            }
            TSQ_CNT(15, 1);
            volatile lds_u32_t* arr = recs + (t % StageCfg::R) * StageCfg::REC_WORDS + StageCfg::ARR + lane;
            const uint32_t spanword = arr[kASpan * 64];
            const uint32_t lane_word = arr[kALane * 64];
            const uint32_t nx = arr[kANx * 64];
            const uint32_t orb_lo = arr[kAOrb * 64], orb_hi = arr[(kAOrb + 1) * 64];
            const uint32_t tin_lo = arr[kATin * 64], tin_hi = arr[(kATin + 1) * 64];
            const uint32_t tp1_lo = arr[kATp1 * 64], tp1_hi = arr[(kATp1 + 1) * 64];
            const uint32_t tp2_lo = arr[kATp2 * 64], tp2_hi = arr[(kATp2 + 1) * 64];
            const uint64_t hard = __ballot((spanword & 0x100u) != 0u);
            const uint64_t certain_m = __ballot((spanword & 0x400u) != 0u);
            const uint64_t near_m = __ballot((spanword & 0x800u) != 0u);
            // a twin visited in the two previous tiles: fixed for the whole tile (kept per lane, it joins the in-tile test)
            const uint32_t prev_hit = (tp1_lo & (uint32_t)vall_p1) | (tp1_hi & (uint32_t)(vall_p1 >> 32)) |
                                      (tp2_lo & (uint32_t)vall_p2) | (tp2_hi & (uint32_t)(vall_p2 >> 32));
            const uint32_t span_nat = spanword & 0xFFu;
            const uint32_t k0 = (spanword >> 16) & 0xFFu;
            const uint32_t cand0 = lane_word & 0xFFFFFFu;

            // The tile's symbols go to the builder as ONE item: visited lanes, which of them are matches, and the
            // per-lane candidate|nibble words -- the builder derives literal chunks, symbol indices and pair origins
            // from the masks.  Hazard lanes resolved below patch the masks; only the rare outcomes the masks cannot
            // express (a literal closed in front of a match that then fails, the block tail) are pushed explicitly,
            // after flushing what is pending.
            uint64_t Vt = 0, Mt = 0;
            uint32_t lw = lane_word;
            uint32_t e_nsym = nsym, e_origin = origin, e_lit_from = lit_from;
            auto flush_pending = [&]() {
                if (Vt != 0ull) {
                    volatile lds_u32_t* it = slot_begin();
                    // the nine header words, one per lane, in one store
                    uint32_t hv = kItemSeg;
                    asm volatile("v_writelane_b32 %0, %1, 1" : "+v"(hv) : "s"(base));
                    asm volatile("v_writelane_b32 %0, %1, 2" : "+v"(hv) : "s"((uint32_t)Vt));
                    asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(hv) : "s"((uint32_t)(Vt >> 32)));
                    asm volatile("v_writelane_b32 %0, %1, 4" : "+v"(hv) : "s"(e_nsym));
                    asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(hv) : "s"(e_origin));
                    asm volatile("v_writelane_b32 %0, %1, 6" : "+v"(hv) : "s"(e_lit_from));
                    asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(hv) : "s"((uint32_t)Mt));
                    asm volatile("v_writelane_b32 %0, %1, 8" : "+v"(hv) : "s"((uint32_t)(Mt >> 32)));
                    if (lane < 9u) it[lane] = hv;
                    it[16 + lane] = lw;
                    slot_publish();
                }
                Vt = 0; Mt = 0;
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
                        if (am) { run0 = q; origin_r0 = origin; odd_r0 = nsym & 1u; am = 0; }
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
                        am = 1;
                        L += sp;
                    }
                }
            };
            uint32_t L = v - base;
            REG_END(1);
            // Costs that shape this loop (one wavefront, measured: tools/micro/issue_rate.hip): ALU instruction 4.5 cycles,
            // branch ~20 cycles taken OR NOT, a VALU result read by the SALU (readlane, ballot) +20 cycles.  So: selects
            // instead of branches, rare cases folded into one test, lane reads batched.
#ifdef TSQ_STATS
            unsigned long long back_ = 0;
#endif
            while (done == 0u) {
#ifdef TSQ_STATS
                if (18 == TSQ_REGION && back_) { st_[11] += __builtin_amdgcn_s_memtime() - back_; st_[12] += 1; }
#endif
                REG_BEGIN(2);
                const uint32_t L0 = L;                                           // < 64
                {
                    const uint32_t fresh = s_nz64(Vt) ^ 1u;
                    e_nsym = s_sel(fresh, nsym, e_nsym); e_origin = s_sel(fresh, origin, e_origin); e_lit_from = s_sel(fresh, lit_from, e_lit_from);
                }
                // the orbit from L0: halts on a hard lane or past the tile (nothing at all if L0 itself is hard)
                const uint32_t o_lo = rdlane(orb_lo, L0), o_hi = rdlane(orb_hi, L0), o_nx = rdlane(nx, L0);
                const uint32_t entry_ok = ((uint32_t)(hard >> L0) & 1u) ^ 1u;
                uint64_t V = s_sel64(entry_ok, (uint64_t)o_lo | ((uint64_t)o_hi << 32), 0ull);
                L = s_sel(entry_ok, o_nx & 0x7Fu, L0);
                // the orbit treated twin lanes as ordinary lanes.  That is wrong for a visited lane that has a VISITED
                // twin before it (earlier in this tile, or in the two previous tiles): its gathered candidate is not
                // current.  The first such lane ends the segment; everything before it is exact.
                const uint64_t seen = vall | V;
                const uint32_t in_lo = tin_lo & (uint32_t)seen, in_hi = tin_hi & (uint32_t)(seen >> 32);
                uint64_t bad = __ballot((in_lo | in_hi | prev_hit) != 0u) & V;
                if (__builtin_expect(s_nz64(V & near_m), 0)) {
                    // near-twin lanes (classed "no match" on the assumption that a twin at most 3 back is visited) are
                    // the other way round: they are right exactly when such a twin was visited
                    uint32_t a_lo = in_lo, a_hi = in_hi;
                    asm volatile("; near twins" : "+v"(a_lo), "+v"(a_hi));             // keeps this block's arithmetic out of the tile prologue
                    const uint32_t pv_lo = tp1_lo & (uint32_t)vall_p1, pv_hi = tp1_hi & (uint32_t)(vall_p1 >> 32);
                    const bool has_in = (a_lo | a_hi) != 0u, has_prev = (pv_lo | pv_hi) != 0u;
                    const uint32_t nearest = a_hi ? 63u - (uint32_t)__builtin_clz(a_hi) : 31u - (uint32_t)__builtin_clz(a_lo | 1u);
                    const uint32_t nearest_prev = pv_hi ? 63u - (uint32_t)__builtin_clz(pv_hi) : 31u - (uint32_t)__builtin_clz(pv_lo | 1u);
                    const bool near_visited = (has_in && lane - nearest < 4u) || (has_prev && lane + 64u - nearest_prev < 4u);
                    bad = ((bad & ~near_m) | (near_m & ~__ballot(near_visited))) & V;
                }
                {
                    const uint32_t trunc = s_nz64(bad);
                    const uint32_t Lb = s_lsb64(bad | (1ull << 63));
                    V = s_sel64(trunc, V & ((1ull << Lb) - 1ull), V);
                    L = s_sel(trunc, Lb, L);
                    TSQ_CNT(23, trunc);
                }
                TSQ_CNT(24, 1);
                REG_END(2);
                REG_BEGIN(3);
                // ---- the segment's effect on the parse state, O(1) from its masks.  `dsym` symbols close (matches and
                //      the literal runs in front of them); the state afterwards hangs on the last match.
                const uint64_t M = V & certain_m, N = V ^ M;
                {
                    const uint32_t nonempty = s_nz64(V), has_m = s_nz64(M);
                    const uint32_t Le = s_msb64(V | 1ull), Lm = s_msb64(M | 1ull);
                    const uint32_t first_isN = (uint32_t)(N >> L0) & 1u;                // L0 is the lowest lane of V
                    uint64_t r = N & (N >> 1); r &= r >> 2; r &= r >> 4; r &= r >> 8;     // a literal run of 16 or more inside
                    const uint32_t run_end = base + s_sel(has_m, s_lsb64(M | (1ull << 63)), Le + 1u);
                    const uint32_t chunk = s_ge(run_end - lit_from, 16u) & first_isN;     // the entry run completes a 16-byte chunk
                    if (__builtin_expect(s_nz64(r) | chunk, 0)) { TSQ_CNT(28, 1); replay_segment(V); }
                    else {
                        const uint32_t e_pos = base + L0, lm_pos = base + Lm, endm = lm_pos + rdlane(span_nat, Lm);
                        const uint32_t last_m = s_eq(Le, Lm) & has_m;
                        const uint32_t pre = s_lt(lit_from, e_pos) & (first_isN ^ 1u);      // a pending literal closes in front of an entry match
                        const uint32_t dsym = (uint32_t)__builtin_popcountll(M) + (uint32_t)__builtin_popcountll(M & (N << 1)) + pre;
                        const uint32_t nsym_n = nsym + s_sel(has_m, dsym, 0u);
                        const uint32_t origin_n = s_sel(has_m, s_sel(nsym_n & 1u, lm_pos, endm), origin);
                        const uint32_t new_run = s_sel(has_m, last_m ^ 1u, am & nonempty);   // a literal run starts inside / at the entry of the segment
                        run0 = s_sel(new_run, s_sel(has_m, endm, e_pos), run0);
                        origin_r0 = s_sel(new_run, origin_n, origin_r0);
                        odd_r0 = s_sel(new_run, nsym_n & 1u, odd_r0);
                        lit_from = s_sel(has_m, endm, lit_from);
                        am = s_sel(nonempty, last_m, am);
                        nsym = nsym_n;
                        origin = origin_n;
                    }
                }
                REG_END(3);
                REG_BEGIN(17);
                Vt |= V; Mt |= M;
                vall |= V;
                if (L >= 64u) { v = base + L; REG_END(17); break; }
                REG_END(17);
                REG_BEGIN(4);
                TSQ_CNT(26, 1); TSQ_CNT(27, ((hard >> L) & 1ull) ? 1 : 0);
                {
                    // ---- exact scalar resolution of one hazard lane (hard, or with a visited twin)
                    const uint32_t i = base + L;
                    const uint64_t bit = 1ull << L;
                    uint32_t cand = rdlane(cand0, L);
                    uint32_t k = rdlane(k0, L);
                    uint32_t twin_cand = 0;
                    {
                        // visited twins of lane L: in this tile (before L) and in the two previous tiles;
                        // the most recent one is the candidate
                        const uint64_t in_tile = ((uint64_t)rdlane(tin_lo, L) | ((uint64_t)rdlane(tin_hi, L) << 32)) & vall;
                        const uint64_t in_p1 = ((uint64_t)rdlane(tp1_lo, L) | ((uint64_t)rdlane(tp1_hi, L) << 32)) & vall_p1;
                        const uint64_t in_p2 = ((uint64_t)rdlane(tp2_lo, L) | ((uint64_t)rdlane(tp2_hi, L) << 32)) & vall_p2;
                        if (in_tile | in_p1 | in_p2) {
                            const uint64_t pick = in_tile ? in_tile : in_p1 ? in_p1 : in_p2;
                            const uint32_t back = in_tile ? 0u : in_p1 ? 64u : 128u;
                            cand = base - back + msb64(pick);
                            k = uniform(prefix16(words_at(i), words_at(cand)));
                            twin_cand = 1;
                            TSQ_CNT(17, in_tile ? 1 : 0); TSQ_CNT(18, (!in_tile && in_p1) ? 1 : 0); TSQ_CNT(19, (!in_tile && !in_p1) ? 1 : 0);
                        }
                    }
                    vall |= bit;
                    const uint32_t e4 = s_ge(k, 4u);
                    const uint32_t f = (i - 1u - run0) >> 5;
                    const uint32_t o_ref = s_sel(f, s_sel(odd_r0, run0 + 32u * f - 16u, run0 + 32u * f), origin_r0);
                    // the first test (tsq_encode.cpp:80,100 in a literal run; :170 right after a match)
                    const uint32_t pass = e4 & s_sel(am, s_lt(i, n - 5u) & s_offset_ok(origin - cand), s_offset_ok(o_ref - cand));
                    if (__builtin_expect(!(i < n), 0)) {
                        // the end of the block (tsq_encode.cpp:120,173)
                        if (am == 0u || pass) {
                            flush_pending();
                            if (am == 0u && i > lit_from) { push(rec_literal(lit_from, i - lit_from), i); lit_from = i; }
                        }
                        am = 0;
                        done = 1;
                    } else if (pass == 0u) {
                        // no match here: the lane is a literal byte, either the first of a new run or one more of the current one
                        // (where a full 16-byte chunk may close: the builder sees that in the masks)
                        Vt |= bit;
                        v = i + 1u;
                        const uint32_t full = s_eq(v - lit_from, 16u) & (am ^ 1u);
                        run0 = s_sel(am, i, run0); origin_r0 = s_sel(am, origin, origin_r0); odd_r0 = s_sel(am, nsym & 1u, odd_r0);
                        nsym += full;
                        origin = s_sel(full & ((nsym & 1u) ^ 1u), v, origin);
                        lit_from = s_sel(am, i, s_sel(full, v, lit_from));
                        am = 0;
                    } else {
                        const uint32_t pend = s_lt(lit_from, i) & (am ^ 1u);   // a pending literal closes in front of the match (tsq_encode.cpp:103-118)
                        if (EXT && twin_cand) {
                            // (the bytes behind the first 16 come from the input window ring when SCAN has put everything up to i + 64 there)
                            const bool in_window = WINDOW && i - cand <= 65536u &&
                                                   uniform(__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= t + 2u;
                            auto win16 = [&](int32_t wi) -> uint4 {
                                wi += wi < 0 ? (int32_t)StageCfg::WIN : 0;
                                wi -= wi >= (int32_t)StageCfg::WIN ? (int32_t)StageCfg::WIN : 0;
                                volatile lds_u32_t* wp = (volatile lds_u32_t*)(lds + StageCfg::off_win + ((uint32_t)wi & ~3u));
                                const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
                                const uint32_t sh = (uint32_t)wi & 3u;
                                return make_uint4(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh),
                                                  __builtin_amdgcn_alignbyte(d3, d2, sh), __builtin_amdgcn_alignbyte(d4, d3, sh));
                            };
                            while (k >= 16u && k < 64u && (k & 15u) == 0u) {
                                const int32_t wi = (int32_t)(wbase + L + k);
                                const uint32_t add = in_window ? uniform(prefix16(win16(wi), win16(wi - (int32_t)(i - cand))))
                                                               : uniform(prefix16(ld128z(src, (uint64_t)i + k, avail), ld128z(src, (uint64_t)cand + k, avail)));
                                k += add;
                                if (add < 16u) break;
                            }
                        }
                        // the pair origin the match sees: after the pending literal, if there is one
                        const uint32_t nsym1 = nsym + pend;
                        const uint32_t origin1 = s_sel(pend & ((nsym1 & 1u) ^ 1u), i, origin);
                        const uint32_t room = origin1 - cand;
                        k = s_sel(s_lt(room, k), room - 1u, k);
                        if (__builtin_expect(s_lt(k, 4u) | (s_offset_ok(room) ^ 1u), 0)) {
                            // the literal was closed and the match then fails: the masks cannot say that
                            if (pend) {
                                flush_pending();
                                push(rec_literal(lit_from, i - lit_from), i);
                                e_nsym = nsym; e_origin = origin; e_lit_from = i;
                            }
                            Vt |= bit;
                            am = 0; run0 = i; origin_r0 = origin; odd_r0 = nsym & 1u; lit_from = i; v = i + 1u;
                        } else {
                            const uint32_t m = length_nibble(k);
                            const uint32_t ni = i + nibble_span(m);
                            nsym = nsym1 + 1u;
                            origin = s_sel(nsym & 1u, origin1, ni);
                            TSQ_CNT(21, 1);
                            Vt |= bit; Mt |= bit;
                            lw = lane == L ? (cand | (m << 24)) : lw;
                            am = 1;
                            lit_from = ni;
                            v = ni;
                        }
                    }
                }
                L = v - base;
                REG_END(4);
#ifdef TSQ_STATS
                if (18 == TSQ_REGION) back_ = __builtin_amdgcn_s_memtime();
#endif
                if (L >= 64u) break;
            }
            REG_BEGIN(5);
            flush_pending();
            REG_END(5);
        }
        // ---- hand the tile's visited mask to MATCH (it commits the table) and move on
        REG_BEGIN(6);
        {
            const uint32_t slot = 16u + 2u * (t & 7u);
            if (lane < 2u) __hip_atomic_store(&ctl[slot + lane], lane ? (uint32_t)(vall >> 32) : (uint32_t)vall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        stage_publish(ctl, 5, t + 1u, lane);
        vall_p2 = vall_p1; vall_p1 = vall;
        REG_END(6);
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[8] = st_[8]; g_enc_stats[9] = st_[9]; g_enc_stats[10] = TSQ_TOTAL(); g_enc_stats[11] = st_[11]; g_enc_stats[12] = st_[12]; g_enc_stats[15] = st_[15]; g_enc_stats[16] = nsym; for (int q = 22; q < 29; ++q) g_enc_stats[q] = st_[q]; g_enc_stats[29] = st_[17]; g_enc_stats[30] = st_[18]; g_enc_stats[31] = st_[19]; g_enc_stats[20] = st_[20]; g_enc_stats[21] = st_[21]; }
#endif
    if (lane == 0) __hip_atomic_store(&ctl[6], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    {
        volatile lds_u32_t* it = slot_begin();
        if (lane == 0) { it[0] = kItemEnd; it[4] = nsym; }
        slot_publish();
    }
}

template <bool EXT, bool WINDOW>
__global__ __launch_bounds__(320) void enc_stage_kernel(const uint8_t* __restrict__ in, uint64_t n_total, uint64_t readable, uint64_t stride,
                                                        uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                        uint16_t* __restrict__ tables, int32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t stage_lds[];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t role = uniform(threadIdx.x >> 6);
    // block b of the launch lies at in + b * stride (stride = 4 MiB: one contiguous buffer; larger: a shard's blocks, each
    // followed by its own look-ahead bytes); its length follows from the virtual total n_total = (blocks - 1) * 4 MiB + last
    const uint64_t start = (uint64_t)b * stride;
    const uint64_t avail = readable - start;
    const uint64_t vstart = (uint64_t)b << kBlockBits;
    const uint32_t n = n_total - vstart < kBlockSize ? (uint32_t)(n_total - vstart) : kBlockSize;
    const uint8_t* src = in + start;
    uint8_t* out = slots + (size_t)b * kSlotSize;
    uint16_t* table = tables + (size_t)b * kHashEntries;

    {   // tsqInit (tsq_context.cpp:77-80), all five waves
        uint4* t4 = reinterpret_cast<uint4*>(table);
        for (uint32_t k = threadIdx.x; k < kHashEntries * 2 / 16; k += 320) t4[k] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < 64) reinterpret_cast<uint32_t*>(stage_lds + StageCfg::off_ctl)[threadIdx.x] = 0;
        uint4* o4 = reinterpret_cast<uint4*>(stage_lds + StageCfg::off_owner);          // owner image: no valid entries
        for (uint32_t k = threadIdx.x; k < (StageCfg::OWN_MASK + 1u) / 16; k += 320) o4[k] = make_uint4(0, 0, 0, 0);
        uint4* f4 = reinterpret_cast<uint4*>(stage_lds + StageCfg::off_f);              // MATCH's filter: empty
        for (uint32_t k = threadIdx.x; k < (StageCfg::F_MASK + 1u) / 16; k += 320) f4[k] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) { out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); }
    }
    __syncthreads();
    lds_u8_t* lds3 = (lds_u8_t*)stage_lds;
    // Five waves on four SIMDs: waves 0 and 4 share one.  A wave64 VALU instruction occupies its SIMD for four cycles, so the
    // two that share should not both be VALU-heavy: the parser is almost pure SALU, the builder almost pure VALU.
    if (role == 0) stage_parser<EXT, WINDOW>(src, avail, n, lds3, lane);
    else if (role == 1) stage_scan<WINDOW>(src, avail, n, lds3, lane);
    else if (role == 2) stage_match<EXT, WINDOW>(src, avail, n, table, lds3, lane);
    else if (role == 3) stage_orbit<EXT>(n, lds3, lane);
    else stream_builder<StageCfg>(src, avail, out, lds3, lane, b, sizes, status);
}


}  // namespace r02
}  // namespace tsq
