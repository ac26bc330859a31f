// tsq_enc_stage.cuh -- staged block encoder for gfx950: fourteen working wavefronts per block, twelve in the lean layout (kernel variant 0).
//
// A single wavefront issues about one instruction every five cycles, and the greedy parse of a block is
// serial (tsq_encode.cpp:72-187: the position table is a function of the parse).  So the block's work
// is cut into stages that stream 64-position tiles through records in LDS; the serial stage is kept as
// small as it can be and everything else is spread over as many wavefronts as it takes to keep up with it:
//
//   HASH     input words (requested four tiles ahead), hashes, the "owner image" (one byte per bucket of a multiplicative fold of
//            the hash: which lane of the last four tiles wrote it last) -- a filter for equal hashes.
//   IN       the twins inside the tile: for every lane the earlier lanes with its hash (one round of ballots per group).
//   TWINS    the exact same-hash ("twin") masks of every lane against the LM tiles before: inherited from the bucket
//            owner's own masks when the owner has the lane's hash (no search), settled with ballots on a fold collision.
//   NEAR     the common prefix of every lane with its nearest twin, ahead of time (WALK's hazard lanes mostly need exactly that).
//   MATCH x2 (even / odd tiles) the candidates: gather from the position table, patch with the visited twins
//            of tile t-LM (mask test), candidate bytes from the input window ring in LDS, common prefix,
//            lane classes (certain match / certain literal / hazard).
//   ORBIT x2 (even / odd tiles) the visited set from every possible entry lane of the tile, by pointer doubling;
//            first the late classification of lanes whose only twins are LF or more tiles back.
//   WALK     THE serial stage: picks the orbit of the actual entry lane, checks the twins it visited, finds a
//            hazard lane's candidate; decides it on the spot when no pair origin can matter, else asks ACCOUNT.
//   ACCOUNT  the symbol state (count, pair origin, pending literal) in O(1) per tile from WALK's masks, the exact
//            scalar decision of the hazards WALK could not decide, the items for the builder.
//   COMMIT   the position table's writer: the visited positions of a tile, as soon as WALK has published them.
//   BUILDER x2 (items alternately) symbol records from the items (tsq_enc_builder.cuh: stream_builder).
//   EMIT     stream layout, 64 symbols at a time (tsq_enc_builder.cuh: stream_emitter).
//
// Table lag.  MATCH gathers tile t from a table that holds the visits of tiles <= t-LM-1 (COMMIT publishes how far it
// is) and patches in the visits of tile t-LM from TWINS' masks, so WALK and MATCH overlap over LM-1 tiles.  What the
// table cannot know -- a visited position of tiles t-LM+1 .. t-1 or an earlier lane of t with the same hash -- is a
// twin: TWINS and IN find all of them exactly, and WALK takes the most recent VISITED twin as the candidate, which is
// what the reference's table would hold (tsq_encode.cpp:76-79), or keeps the gathered candidate when no twin
// was visited.  The output therefore does not depend on how the wavefronts interleave (make jitter: a stress
// build that delays every hand-off pseudo-randomly; tests/test_gpu_parity.py::test_encoder_handoffs_under_jitter).
#pragma once
#ifndef TSQ_LATE_FIX
#define TSQ_LATE_FIX 1
#endif
// Table lag LM: MATCH gathers tile t from a table that holds the visits of tiles <= t-LM-1 and patches in the visits of tile t-LM.
// Late-fix lag LF: ORBIT settles the lanes whose only twins lie in tiles t-LM+1 .. t-LF once WALK has finished tile t-LF.
// WALK looks after the twins of tiles t-LM+1 .. t.  LM = 3, LF = 2 was round 3's pipeline: the loops WALK(t-3) -> MATCH(t) -> ORBIT(t) ->
// WALK(t) and WALK(t-2) -> ORBIT(t) -> WALK(t) paced it together with the serial stage itself.  With LM = 4, LF = 3 (standard layout)
// neither loop is ever waited for (tools/bottleneck.sh: MATCH, ORBIT, COMMIT at 0.01 - 0.04), at the price of 0.18 more hazard lanes
// per tile for WALK; what paces the pipeline then is the busiest wavefront -- WALK itself.  The lean layout (two blocks per CU hide each
// other's waits) gains little from it: 4 GiB of text 34.4 GB/s either way without extensions, 32.7 against 31.4 with them.
#ifndef TSQ_LM
#define TSQ_LM 4
#endif
#ifndef TSQ_LF
#define TSQ_LF 3
#endif
#ifndef TSQ_LM_LEAN
#define TSQ_LM_LEAN 4
#endif
#ifndef TSQ_LF_LEAN
#define TSQ_LF_LEAN 3
#endif

#include "tsq_common.cuh"
#include "tsq_experiment.h"
#include "tsq_enc_util.cuh"
#include "tsq_enc_builder.cuh"

namespace tsq {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2_t lds_u32x2_t;

// WINDOW = the standard layout (one block per CU: the input window ring in LDS, ten tile records); the lean layout (two blocks per CU)
// has no window and eight records.
template <bool WINDOW>
struct StageCfgT {
    static constexpr uint32_t THREADS = 1024;                     // sixteen wavefronts are launched, twelve work (see the kernel)
    static constexpr uint32_t THREADS_LEAN = 768;                 // the lean layout launches its twelve working ones only (two workgroups of more do not fit a CU)
    static constexpr bool SPLIT_IN = WINDOW;                      // the in-tile twin search on a wavefront of its own (IN)
    static constexpr bool DUAL_BUILDER = WINDOW;                  // two BUILDER wavefronts take the items alternately
    static constexpr uint32_t EQ = 64, EV_WORDS = 4;              // events between WALK and ACCOUNT
    static constexpr uint32_t Q = WINDOW ? 16 : 8;                // items between ACCOUNT and BUILDER (the queue is never full: profiles/r03_encoder_spin_counts.txt)
    static constexpr uint32_t ITEM_WORDS = 80;
    static constexpr uint32_t RING = 256;                          // symbol records between BUILDER and EMIT (four batches)
    static constexpr uint32_t LM = WINDOW ? TSQ_LM : TSQ_LM_LEAN, LF = WINDOW ? TSQ_LF : TSQ_LF_LEAN;
    static_assert((LM == 3u || LM == 4u) && LF >= 2u && LF < LM, "lags");
#ifndef TSQ_RECORDS
#define TSQ_RECORDS 11
#endif
    // tile records in flight (HASH runs at most R - LM tiles ahead of WALK).  Measured in the standard layout: 9 records 44.5 ms, 10 41.9,
    // 11 41.1, 12 41.5, 13 41.1, 14 41.4 (the even counts, where the even and the odd tiles' wavefronts keep to their own slots, are the
    // slower ones); the eleventh record has the room the input window's margin gave up.  The lean layout: 9, 10 or 11 make no difference.
    static constexpr uint32_t R = LM == 3u ? (WINDOW ? 10 : 8) : (WINDOW ? TSQ_RECORDS : 10);
#ifndef TSQ_OWNBITS
#define TSQ_OWNBITS 15
#endif
    static constexpr uint32_t OWN_MASK = (WINDOW ? TSQ_OWNBITS == 15 : LM == 3u) ? 0x7FFFu : 0x3FFFu;   // owner image: hash folded to 15 bits (14 in the lean layout when it keeps ten records)
    // input window ring: the last 64 KiB of input and what HASH may be ahead of MATCH and WALK: while they work on tile t, WALK has not
    // finished it, so HASH is at tile t + R - LM at most -- 64 KiB + 7 tiles + a tile's own 64 bytes = 66 047; a multiple of 64
#ifndef TSQ_WIN
#define TSQ_WIN 66560
#endif
    static constexpr uint32_t WIN = TSQ_WIN;
    static_assert(WIN % 64u == 0u && WIN > 65536u + (TSQ_RECORDS - 3u) * 64u + 63u, "window margin");
    static constexpr uint32_t W16 = 16;                           // word offset of the uint4-per-lane input words
    static constexpr uint32_t ARR = 16 + 256;                     // word offset of the per-lane words: four groups of four words per lane
    static constexpr uint32_t REC_WORDS = ARR + 16 * 64;
    // bucket of a hash in the owner image: the top bits of a 32-bit multiplicative hash.  (Masking the 17-bit hash drops two bits that
    // carry entropy on text: 1.6 lanes per tile met another hash in their bucket, three times what 256 live entries make likely, and
    // every one costs TWINS a round of ballots: 44.0 -> 43.5 ms.)
    static __device__ __forceinline__ uint32_t fold(uint32_t h) { return (h * 0x9E3779B1u) >> (OWN_MASK == 0x7FFFu ? 17 : 18); }
    static constexpr uint32_t off_owner = 0;                                   // u8[OWN_MASK + 1]
    static constexpr uint32_t off_queue = OWN_MASK + 1u;                       // u32[Q * ITEM_WORDS]
    static constexpr uint32_t off_ring = off_queue + Q * ITEM_WORDS * 4;       // u32[RING]
    static constexpr uint32_t off_rec = off_ring + RING * 4;                   // u32[R * REC_WORDS]
    static constexpr uint32_t off_ctl = off_rec + R * REC_WORDS * 4;           // u32[64]
    static constexpr uint32_t off_evq = off_ctl + 256;                         // u32[EQ * EV_WORDS]
    static constexpr uint32_t off_win = off_evq + EQ * EV_WORDS * 4;           // u8[WIN + 32]: the ring, its first 32 bytes mirrored at the end
    static constexpr uint32_t total = WINDOW ? off_win + WIN + 32 : off_win;   // (without the window two blocks fit one CU)
    static_assert(total <= (WINDOW ? 160u : 80u) * 1024u, "LDS budget");
};
// ctl words: 0 queue head, 1 queue tail, 2 tiles with twin masks, 3 / 35 even / odd tiles matched, 10 / 15 even / odd tiles with orbits,
//            4 events produced (WALK -> ACCOUNT) and 5 tiles walked: an aligned pair, ACCOUNT reads both with one 8-byte load, 6 stop, 7..9 BUILDER/EMIT (tsq_enc_builder.cuh), 10..14 WALK/ACCOUNT events, 33 tiles committed, 34 tiles hashed,
// record: header words 2,3 = the lanes the parse visited (WALK)
//         per-lane words, in four groups of four: group g of lane l is the 16-byte LDS word at ARR + g * 256 + l * 4, so that a stage
//         reads or writes a whole group (or half of one) with ONE LDS instruction -- the LDS pipe is what all the wavefronts share,
//         and a 4-byte access per lane costs it as much as an 8-byte one and half of a 16-byte one:
//           A: spanword | candidate | nibble << 24 | orbit mask lo, hi                 (MATCH, ORBIT -> WALK, ACCOUNT)
//           B: owner word (HASH -> TWINS), then the nearest twin's word (NEAR) | hash | twins in this tile (earlier lanes) lo, hi
//              (a two-word access at an odd word is a misaligned LDS access: ~40 cycles of the LDS pipe instead of 5 -- SQ_LDS_UNALIGNED_STALL)
//           C: twins in tile t-1 lo, hi | twins in tile t-2 lo, hi                     (TWINS -> NEAR, MATCH, ORBIT, WALK)
//           D: twins in tile t-3 lo, hi | twins in tile t-4 lo, hi
// spanword: natural span (bits 0..7) | hard (8) | has a twin (9) | certain match (10) | near twin (11) | twins of the late-fix tiles settled (12) |
//           common prefix (16..23) | orbit halt (24..31: ORBIT)
enum : uint32_t { kGA = 0, kGB = 256, kGC = 512, kGD = 768 };
__device__ __forceinline__ u32x4_t lds_ld4(volatile lds_u32_t* p) { return *(volatile lds_u32x4_t*)p; }
__device__ __forceinline__ u32x2_t lds_ld2(volatile lds_u32_t* p) { return *(volatile lds_u32x2_t*)p; }
__device__ __forceinline__ void lds_st4(volatile lds_u32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { u32x4_t v; v.x = a; v.y = b; v.z = c; v.w = d; *(volatile lds_u32x4_t*)p = v; }
__device__ __forceinline__ void lds_st2(volatile lds_u32_t* p, uint32_t a, uint32_t b) { u32x2_t v; v.x = a; v.y = b; *(volatile lds_u32x2_t*)p = v; }
// events between WALK and ACCOUNT, and their ctl words (4 events produced, 11 consumed, 12 queries answered, 13 the answer,
// 14 the tile ACCOUNT works on: the tiles before it are accounted)
enum : uint32_t { kEvSeg = 1, kEvHaz = 2, kEvEnd = 3 };
enum : uint32_t { kCtlEvHead = 4, kCtlOrbitEven = 10, kCtlEvTail = 11, kCtlReplies = 12, kCtlReplyValue = 13, kCtlAccounted = 14, kCtlOrbitOdd = 15, kCtlCommitted = 33, kCtlHashed = 34, kCtlMatchedOdd = 35, kCtlNear = 36, kCtlIn = 37, kCtlOrbitQ = 16 /* .. 19: tiles with orbits by t % 4 (stage_match_orbit) */ };

// Instrumented builds time only the spin loops (and only when they actually spin): s_memtime costs a few
// hundred cycles, so finer timing distorts the pipeline it measures.  busy = total - waited.
#ifdef TSQ_STATS
#define TSQ_BEGIN() const unsigned long long begin_ = __builtin_amdgcn_s_memtime()
#define TSQ_WAITED(slot, expr) do { const unsigned long long w0_ = __builtin_amdgcn_s_memtime(); expr; st_[slot] += __builtin_amdgcn_s_memtime() - w0_; } while (0)
#define TSQ_TOTAL() (__builtin_amdgcn_s_memtime() - begin_)
// one timed region per build (-DTSQ_REGION=k): two s_memtime per pass through the region, calibrated by region 0 (empty)
#ifndef TSQ_REGION
#define TSQ_REGION -1
#endif
#define REG_BEGIN(k) unsigned long long r0_##k = 0; if (k == TSQ_REGION) r0_##k = __builtin_amdgcn_s_memtime()
#define REG_END(k) do { if (k == TSQ_REGION) { st_[11] += __builtin_amdgcn_s_memtime() - r0_##k; st_[12] += 1; } } while (0)
// the same for the MATCH wave (regions 10..14); the end of a region drains the memory counters so that latency lands in it
#define MREG_BEGIN(k) unsigned long long m0_##k = 0; if (k == TSQ_REGION) m0_##k = __builtin_amdgcn_s_memtime()
#define MREG_END(k) do { if (k == TSQ_REGION) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st_[13] += __builtin_amdgcn_s_memtime() - m0_##k; st_[14] += 1; } } while (0)
#else
#define MREG_BEGIN(k) do {} while (0)
#define MREG_END(k) do {} while (0)
#ifdef TSQ_MARKS      // (-S builds: the parser's regions show in the assembly)
#define REG_BEGIN(k) asm volatile("; REG_BEGIN " #k ::: "memory")
#define REG_END(k) asm volatile("; REG_END " #k ::: "memory")
#else
#define REG_BEGIN(k) do {} while (0)
#define REG_END(k) do {} while (0)
#endif
#define TSQ_BEGIN() do {} while (0)
#endif

// Hand-off stress (-DTSQ_JITTER, `make jitter`; tests/test_gpu_parity.py::test_encoder_handoffs_under_jitter): every publication
// and every tile of every stage is delayed by a pseudo-random number of cycles that differs from block to block, so that one launch
// over many copies of a block runs the pipeline under many different interleavings; the streams must not change.
#ifdef TSQ_JITTER
__device__ __forceinline__ void stage_jitter(uint32_t salt)
{
    uint32_t x = (blockIdx.x + 1u) * 2654435761u ^ (salt + (threadIdx.x >> 6) * 977u) * 40503u;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    x = uniform(x);
    if ((x & 3u) == 0u) for (uint32_t q = (x >> 2) & 15u; q != 0u; --q) __builtin_amdgcn_s_sleep(3);
}
#define TSQ_JIT(salt) stage_jitter(salt)
#else
#define TSQ_JIT(salt) do {} while (0)
#endif

// Consuming a record: the counter is read first, the record's words after it.  The LDS executes the DS operations of a
// wavefront in program order (see stage_publish), so only the compiler has to be kept from hoisting record loads above
// the counter load: the barrier below is the acquire half of the handshake at compiler level.
__device__ __forceinline__ bool stage_ready(lds_u32_t* ctl, uint32_t word, uint32_t need)
{
    const uint32_t seen = uniform(__hip_atomic_load(&ctl[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    asm volatile("" ::: "memory");
    return seen >= need;
}
__device__ __forceinline__ bool stage_spin_tight(lds_u32_t* ctl, uint32_t word, uint32_t need)
{
    for (;;) {
        if (stage_ready(ctl, word, need)) return true;
        if (uniform(__hip_atomic_load(&ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) return false;
        TSQ_SPIN(ctl);
    }
}
__device__ __forceinline__ bool stage_spin(lds_u32_t* ctl, uint32_t word, uint32_t need)
{
    for (;;) {
        if (stage_ready(ctl, word, need)) return true;
        if (uniform(__hip_atomic_load(&ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) return false;
        TSQ_SPIN(ctl);
        __builtin_amdgcn_s_sleep(1);
    }
}
// The same with the last value seen kept by the caller: a stage that runs behind its producer finds most of its waits
// satisfied by what it read a few tiles ago and skips the LDS round trip.
__device__ __forceinline__ bool stage_spin_seen(lds_u32_t* ctl, uint32_t word, uint32_t need, uint32_t& seen)
{
    for (;;) {
        seen = uniform(__hip_atomic_load(&ctl[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        asm volatile("" ::: "memory");
        if (seen >= need) return true;
        if (uniform(__hip_atomic_load(&ctl[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) != 0u) return false;
        TSQ_SPIN(ctl);
        __builtin_amdgcn_s_sleep(1);
    }
}
#ifdef TSQ_STATS
#define stage_wait_seen(ctl, word, need, seen, slot) ((seen) >= (need) || [&]() { bool ok_; TSQ_WAITED(slot, ok_ = stage_spin_seen(ctl, word, need, seen)); return ok_; }())
#else
#define stage_wait_seen(ctl, word, need, seen, slot) ((seen) >= (need) || stage_spin_seen(ctl, word, need, seen))
#endif
#ifdef TSQ_STATS
#define stage_wait_tight(ctl, word, need, slot) (stage_ready(ctl, word, need) || [&]() { bool ok_; TSQ_WAITED(slot, ok_ = stage_spin_tight(ctl, word, need)); return ok_; }())
#else
#define stage_wait_tight(ctl, word, need, slot) (stage_ready(ctl, word, need) || stage_spin_tight(ctl, word, need))
#endif
#ifdef TSQ_STATS
#define stage_wait(ctl, word, need, slot) (stage_ready(ctl, word, need) || [&]() { bool ok_; TSQ_WAITED(slot, ok_ = stage_spin(ctl, word, need)); return ok_; }())
#else
#define stage_wait(ctl, word, need, slot) (stage_ready(ctl, word, need) || stage_spin(ctl, word, need))
#endif
// Publishing a record: the DS operations of one wavefront are executed by the LDS in program order, so a
// counter stored after the record's words becomes visible after them; only the compiler has to be kept
// from reordering (an s_waitcnt here would park the wave for a full LDS round trip per tile).
#define TSQ_LDS_RELEASE() asm volatile("" ::: "memory")
__device__ __forceinline__ void stage_publish(lds_u32_t* ctl, uint32_t word, uint32_t value, uint32_t lane)
{
    TSQ_JIT(value * 64u + word);
    TSQ_LDS_RELEASE();
    // (every lane stores the same word: cheaper than masking the wavefront down to one lane -- measured both ways, 40.9 -> 40.7 ms)
    (void)lane;
    __hip_atomic_store(&ctl[word], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------------------------------------- SCAN
// SCAN is two wavefronts: HASH loads the tile's input words, hashes them and runs the owner image (which lanes share a
// bucket: a filter); TWINS turns that into the exact twin masks.
template <bool WINDOW>
__device__ __forceinline__ void stage_hash(const uint8_t* src, uint64_t avail, uint32_t n, lds_u8_t* lds, uint32_t lane, const uint16_t* table = nullptr)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u8_t* owner = lds + StageCfg::off_owner;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    const uint32_t n_tiles = (n >> 6) + 3u;            // visits reach at most n + 63
    uint32_t wbase = 0;                                 // (t * 64) % WIN
    uint32_t parsed_seen = 0, accounted_seen = 0, committed_seen = 0, twinned_seen = 0;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    // The input words are requested D tiles ahead: every other tile starts a new 128-byte line that comes from HBM (two microseconds,
    // more than a tile period), and this wavefront feeds every other stage.
#ifndef TSQ_HASH_AHEAD
#define TSQ_HASH_AHEAD 4
#endif
    constexpr uint32_t D = TSQ_HASH_AHEAD;                     // tiles the input words are requested ahead
    uint4 w_q[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) w_q[d] = ld128z(src, (uint64_t)lane + 64u * d, avail);
    auto one_tile = [&](uint32_t t, const uint4 w16) -> bool {
        // the slot of tile t-R is free once WALK has finished tile t-R+LM-1 (it reads the words of LM-1 tiles back), ACCOUNT and
        // COMMIT are past tile t-R, and TWINS has finished tile t-R+LM (tile q's masks are inherited from the records of tiles q-1 .. q-LM)
        if (t + StageCfg::LM > StageCfg::R && !stage_wait_seen(ctl, 5, t + StageCfg::LM - StageCfg::R, parsed_seen, 0)) return false;
        if (t + 1u > StageCfg::R && !stage_wait_seen(ctl, kCtlAccounted, t + 1u - StageCfg::R, accounted_seen, 0)) return false;
        if (t + 1u > StageCfg::R && !stage_wait_seen(ctl, kCtlCommitted, t + 1u - StageCfg::R, committed_seen, 0)) return false;
        if (t + StageCfg::LM + 1u > StageCfg::R && !stage_wait_seen(ctl, 2, t + StageCfg::LM + 1u - StageCfg::R, twinned_seen, 0)) return false;
        TSQ_TRACE(0, t);
        const uint32_t h = hash4(w16.x);
        const uint32_t hf = StageCfg::fold(h);
        // The owner image: per folded hash, the last lane that had it and the low two bits of its tile number.  Nothing is ever
        // retired: an entry is taken for what it says -- a lane one to four tiles back -- and TWINS checks it against that lane's own
        // hash: a lane that really owns the bucket has this folded hash; the zero the image starts with and entries older than four
        // tiles name a lane that (but for a coincidence, which is settled exactly like any fold collision) has not.
        const uint32_t tag = ((t & 3u) << 6) | lane;
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        const uint32_t before = owner[hf];
        owner[hf] = (uint8_t)tag;
        const uint32_t after = owner[hf];
        {
            u32x4_t v; v.x = w16.x; v.y = w16.y; v.z = w16.z; v.w = w16.w;
            *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u) = v;
            // the tile's 64 input bytes join the window ring MATCH takes candidate bytes from
            if (WINDOW && (lane & 15u) == 0u) {
                *(volatile lds_u32x4_t*)(lds + StageCfg::off_win + wbase + lane) = v;
                if (wbase == 0u && lane < 32u) *(volatile lds_u32x4_t*)(lds + StageCfg::off_win + StageCfg::WIN + lane) = v;
            }
        }
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane * 4u;
        // group B: owner before | another lane of the tile took the bucket; hash; (twins in the tile: IN / TWINS)
        lds_st4(arr + kGB, before | (after != tag ? 0x10000u : 0u), h, 0u, 0u);
        TSQ_TRACE(1, t);
        TSQ_DELAY(0);
        stage_publish(ctl, kCtlHashed, t + 1u, lane);
        wbase = wbase + 64u == StageCfg::WIN ? 0u : wbase + 64u;
        return true;
    };
    // (unrolled by hand so that the queue of requested words stays in fixed registers)
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += D) {
        bool go = true;
#pragma unroll
        for (uint32_t d = 0; d < D; ++d) {
            const uint32_t t = t0 + d;
            if (go && t < n_tiles) {
                go = one_tile(t, w_q[d]);
                w_q[d] = ld128z(src, (uint64_t)(t + D) * 64u + lane, avail);      // this wave's only global access
            }
        }
        if (!go) break;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[0] = st_[0]; g_enc_stats[1] = TSQ_TOTAL(); }
#endif
}

// The twins inside the tile: for each lane the mask of EARLIER lanes with the same hash (one round per group of equal hashes among the
// lanes whose bucket another lane of the tile took).
__device__ __forceinline__ uint64_t in_tile_twins(uint32_t h, uint32_t own, uint64_t below_me)
{
    uint64_t twin_in = 0;
    uint64_t shared = __ballot((own & 0x10000u) != 0u);
    while (shared) {
        const uint32_t hl = rdlane(h, lsb64(shared));
        const uint64_t grp = __ballot(h == hl);
        if (h == hl) twin_in = grp & below_me;
        shared &= ~grp;
    }
    return twin_in;
}

// IN: the twins inside the tile -- for each lane the mask of EARLIER lanes with the same hash.  Independent from tile to tile, so it
// has a wavefront of its own in the standard layout (it was TWINS' data-dependent loop: one round per group of equal hashes, 2.3 rounds
// per tile on text).  The lean layout keeps it inside TWINS: two workgroups share a CU only with at most twelve wavefronts each.
template <bool WINDOW>
__device__ __forceinline__ void stage_in(uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    const uint32_t n_tiles = (n >> 6) + 3u;
    uint32_t hashed_seen = 0, slot = 0;
    const uint64_t below_me = below(lane);
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = 0; t < n_tiles; ++t, slot = slot + 1u == StageCfg::R ? 0u : slot + 1u) {
        if (!stage_wait_seen(ctl, kCtlHashed, t + 1u, hashed_seen, 0)) break;
        volatile lds_u32_t* arr = recs + slot * StageCfg::REC_WORDS + StageCfg::ARR + lane * 4u;
        const u32x4_t gb = lds_ld4(arr + kGB);
        const uint32_t h = gb.y, own = gb.x;
        const uint64_t twin_in = in_tile_twins(h, own, below_me);
        lds_st2(arr + kGB + 2u, (uint32_t)twin_in, (uint32_t)(twin_in >> 32));
        TSQ_DELAY(10);
        stage_publish(ctl, kCtlIn, t + 1u, lane);
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[34] = st_[22]; g_enc_stats[57] = st_[0]; g_enc_stats[58] = TSQ_TOTAL(); }
#endif
}

template <bool WINDOW>
__device__ __forceinline__ void stage_twins(uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    using StageCfg = StageCfgT<WINDOW>;
    constexpr uint32_t LM = StageCfg::LM;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    const uint32_t n_tiles = (n >> 6) + 3u;
    uint32_t h_m1 = 0xFFFFFFFFu, h_m2 = 0xFFFFFFFFu, h_m3 = 0xFFFFFFFFu, h_m4 = 0xFFFFFFFFu;   // hashes of tiles t-1 .. t-4 (per lane)
    uint32_t hashed_seen = 0, slot = 0;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = 0; t < n_tiles; ++t, slot = slot + 1u == StageCfg::R ? 0u : slot + 1u) {
        if (!stage_wait_seen(ctl, StageCfg::SPLIT_IN ? kCtlIn : kCtlHashed, t + 1u, hashed_seen, 0)) break;   // (IN has finished the tile, so HASH has)
        volatile lds_u32_t* rec = recs + slot * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane * 4u;
        const u32x4_t gb = lds_ld4(arr + kGB);
        const uint32_t h = gb.y, own = gb.x;
        const uint32_t before = own & 0xFFu;
        if (!StageCfg::SPLIT_IN) {
            const uint64_t twin_in = in_tile_twins(h, own, below(lane));
            lds_st2(arr + kGB + 2u, (uint32_t)twin_in, (uint32_t)(twin_in >> 32));
        }
        // ---- twins in the LM previous tiles (t-1 .. t-LM+1: the parser's business; t-LM: MATCH folds its visited lanes into the
        // candidates; older tiles are in the table).  If the bucket's owner -- the MOST RECENT lane with this folded hash, d tiles back --
        // has this lane's hash, this lane's twins are the owner and the owner's own twins (already exact, by induction): no search.
        // If it has another hash (a fold collision), a twin may hide behind it: settled with ballots below.
        uint64_t twin_p1 = 0, twin_p2 = 0, twin_p3 = 0, twin_p4 = 0;
        bool unsure = false;
        const uint32_t d0 = (t - (before >> 6)) & 3u;
        const uint32_t d = d0 ? d0 : 4u;                                          // how many tiles back the owner says it is
        const bool live = d <= t && d <= LM;
        if (__ballot(live) != 0ull) {
            const uint32_t q = before & 63u;
            const uint32_t dd = live ? d : 1u;
            const uint32_t qslot = slot >= dd ? slot - dd : slot + StageCfg::R - dd;
            volatile lds_u32_t* qa = recs + qslot * StageCfg::REC_WORDS + StageCfg::ARR + q * 4u;
            const u32x4_t qb = lds_ld4(qa + kGB);
            const u32x4_t qc = lds_ld4(qa + kGC);
            const uint32_t hq = qb.y;
            const uint64_t q_in = (uint64_t)qb.z | ((uint64_t)qb.w << 32);
            const uint64_t q_p1 = (uint64_t)qc.x | ((uint64_t)qc.y << 32);
            const uint64_t q_p2 = (uint64_t)qc.z | ((uint64_t)qc.w << 32);
            uint64_t q_p3 = 0;
            if (LM == 4u) { const u32x2_t qd = lds_ld2(qa + kGD); q_p3 = (uint64_t)qd.x | ((uint64_t)qd.y << 32); }
            const bool owns = live && StageCfg::fold(hq) == StageCfg::fold(h);      // the named lane has this folded hash: the entry is what it says
            const bool same = owns && hq == h;
            unsure = owns && hq != h;
            const uint64_t chain = q_in | (1ull << q);
            if (same) {
                // twins k tiles back: none for k < d, the owner and its in-tile twins for k = d, the owner's twins k - d tiles before ITS tile beyond
                twin_p1 = d == 1u ? chain : 0ull;
                twin_p2 = d == 2u ? chain : d == 1u ? q_p1 : 0ull;
                twin_p3 = d == 3u ? chain : d == 2u ? q_p1 : d == 1u ? q_p2 : 0ull;
                if (LM == 4u) twin_p4 = d == 4u ? chain : d == 3u ? q_p1 : d == 2u ? q_p2 : q_p3;
            }
        }
        {
            uint64_t maybe = __ballot(unsure);
            TSQ_CNT(20, __builtin_popcountll(maybe));
            while (maybe) {
                TSQ_CNT(21, 1);
                const uint32_t hl = rdlane(h, lsb64(maybe));
                const uint64_t g1 = __ballot(h_m1 == hl), g2 = __ballot(h_m2 == hl), g3 = __ballot(h_m3 == hl);
                const uint64_t g4 = LM == 4u ? __ballot(h_m4 == hl) : 0ull;
                const uint64_t grp_cur = __ballot(h == hl);
                if (h == hl) { twin_p1 = g1; twin_p2 = g2; twin_p3 = g3; twin_p4 = g4; }
                maybe &= ~grp_cur;
            }
        }
        lds_st4(arr + kGC, (uint32_t)twin_p1, (uint32_t)(twin_p1 >> 32), (uint32_t)twin_p2, (uint32_t)(twin_p2 >> 32));
        lds_st4(arr + kGD, (uint32_t)twin_p3, (uint32_t)(twin_p3 >> 32), (uint32_t)twin_p4, (uint32_t)(twin_p4 >> 32));
        TSQ_TRACE(2, t);
        TSQ_DELAY(1);
        stage_publish(ctl, 2, t + 1u, lane);
        h_m4 = h_m3; h_m3 = h_m2; h_m2 = h_m1; h_m1 = h;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[48] = st_[0]; g_enc_stats[49] = TSQ_TOTAL(); g_enc_stats[32] = st_[20]; g_enc_stats[33] = st_[21]; }
#endif
}

// ---------------------------------------------------------------------------------------------- NEAR
// Off the critical loop, tiles ahead of WALK: for every lane that has a twin in the window (its own tile and the LM-1
// before), the common prefix with its NEAREST twin.  When WALK meets a hazard lane, the candidate is the most recent VISITED twin;
// five times out of six that is the nearest twin, and then the prefix is here already (no LDS round trip and no byte compare on
// the serial stage).  Word per lane (it takes the place of HASH's owner word, which TWINS has consumed by now):
//   bit 15 valid | tiles back (0..3) << 12 | twin's lane << 6 | common prefix (0..16)
template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_near(uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    const uint32_t n_tiles = (n >> 6) + 3u;
    uint32_t scanned_seen = 0, slot = 0;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = 0; t < n_tiles; ++t, slot = slot + 1u == StageCfg::R ? 0u : slot + 1u) {
        if (!stage_wait_seen(ctl, 2, t + 1u, scanned_seen, 0)) break;
        volatile lds_u32_t* rec = recs + slot * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane * 4u;
        const u32x2_t gi = lds_ld2(arr + kGB + 2u);
        const u32x4_t gc = lds_ld4(arr + kGC);
        u32x2_t gd; gd.x = 0; gd.y = 0;
        if (StageCfg::LM == 4u) gd = lds_ld2(arr + kGD);
        const uint32_t tin_lo = gi.x, tin_hi = gi.y, tp1_lo = gc.x, tp1_hi = gc.y, tp2_lo = gc.z, tp2_hi = gc.w, tp3_lo = gd.x, tp3_hi = gd.y;
        const bool in0 = (tin_lo | tin_hi) != 0u, in1 = (tp1_lo | tp1_hi) != 0u, in2 = (tp2_lo | tp2_hi) != 0u, in3 = (tp3_lo | tp3_hi) != 0u;
        uint32_t word = 0;
        if (__ballot(in0 || in1 || in2 || in3) != 0ull) {
            const uint32_t m_lo = in0 ? tin_lo : in1 ? tp1_lo : in2 ? tp2_lo : tp3_lo, m_hi = in0 ? tin_hi : in1 ? tp1_hi : in2 ? tp2_hi : tp3_hi;
            const uint32_t q = m_hi ? 63u - (uint32_t)__builtin_clz(m_hi) : 31u - (uint32_t)__builtin_clz(m_lo | 1u);
            const uint32_t back = in0 ? 0u : in1 ? 1u : in2 ? 2u : 3u;
            const uint32_t qslot = slot >= back ? slot - back : slot + StageCfg::R - back;
            const u32x4_t a = *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u);
            const u32x4_t b = *(volatile lds_u32x4_t*)(recs + qslot * StageCfg::REC_WORDS + StageCfg::W16 + q * 4u);
            const uint32_t k = prefix16(make_uint4(a.x, a.y, a.z, a.w), make_uint4(b.x, b.y, b.z, b.w));
            // (with extensions a prefix of 16 may go on: WALK takes the long way round for those)
            if ((in0 || in1 || in2 || in3) && !(EXT && k >= 16u)) word = 0x8000u | (back << 12) | (q << 6) | k;
        }
        arr[kGB] = word;
        TSQ_DELAY(2);
        stage_publish(ctl, kCtlNear, t + 1u, lane);
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[55] = st_[0]; g_enc_stats[56] = TSQ_TOTAL(); }
#endif
}

// --------------------------------------------------------------------------------------------- MATCH
template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_match(const uint8_t* src, uint64_t avail, uint32_t n, uint16_t* table, lds_u8_t* lds, uint32_t lane, uint32_t parity)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    constexpr uint32_t kDMin = EXT ? 128u : 64u;
    constexpr uint32_t LM = StageCfg::LM;
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;
    const uint32_t n_tiles = (n >> 6) + 3u;
    // two MATCH wavefronts take the even and the odd tiles (nothing is carried from tile to tile)
    uint32_t wbase = parity << 6;                      // (t * 64) % WIN
    uint32_t scanned_seen = 0, committed_seen = 0;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = parity; t < n_tiles; t += 2u) {
        MREG_BEGIN(10);
        if (!stage_wait_seen(ctl, 2, t + 1u, scanned_seen, 2)) break;
        TSQ_TRACE(3, t);
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane * 4u;
        const u32x4_t gb = lds_ld4(arr + kGB);
        const u32x4_t wv = *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u);
        const u32x4_t gc = lds_ld4(arr + kGC);
        const u32x4_t gd = lds_ld4(arr + kGD);
        const uint32_t h = gb.y;
        const uint4 w16 = make_uint4(wv.x, wv.y, wv.z, wv.w);
        const uint64_t twin_in = (uint64_t)gb.z | ((uint64_t)gb.w << 32);
        const uint64_t twin_p1 = (uint64_t)gc.x | ((uint64_t)gc.y << 32);
        // twins in the other tiles WALK looks after (t-2 .. t-LM+1), and the mask MATCH patches with (tile t-LM)
        const uint32_t tpw_any = LM == 4u ? (gc.z | gc.w | gd.x | gd.y) : (gc.z | gc.w);
        const uint32_t tpm_lo = LM == 4u ? gd.z : gd.x, tpm_hi = LM == 4u ? gd.w : gd.y;
        MREG_END(10);
        MREG_BEGIN(12);
        // ---- the table holds the visits of tiles <= t-LM-1 (the COMMIT wave has published them: its stores are complete and the
        //      wavefronts of a workgroup share the vector L1): gather from it right away, without waiting for the parser ...
#ifndef TSQ_X_NOCOMMITWAIT     // (timing only, wrong streams: the gather does not wait for COMMIT -- what does the loop WALK -> COMMIT -> MATCH -> ORBIT -> WALK cost?)
        if (t >= LM + 1u && !stage_wait_seen(ctl, kCtlCommitted, t - LM, committed_seen, 3)) break;
#endif
        TSQ_TRACE(10, t);
#ifdef TSQ_X_FAKE_TABLE   // timing only (wrong streams): every gather hits 512 bytes of the table
        const uint32_t tv_old = table[h & 0xFFu];
#else
        const uint32_t tv_old = table[h];                // (a plain load: an atomic one is waited for on the spot, and the gather's latency must stay hidden)
#endif
        MREG_END(12);
        MREG_BEGIN(11);
        // ... and bring the entries up to "visits of tiles <= t-LM" once the parser has finished tile t-LM: a lane with a
        // visited twin there (SCAN's exact mask) takes the most recent one -- what the committed table would hold --, the others
        // keep theirs.
        uint32_t tv = tv_old;
#ifdef TSQ_X_NOPATCH      // timing only (streams differ where a twin in tile t-LM was visited): MATCH does not wait for WALK's verdict on tile t-LM
        if (false) {
#else
        if (t >= LM) {
#endif
            volatile lds_u32_t* vis = recs + ((t - LM) % StageCfg::R) * StageCfg::REC_WORDS + 2u;
            uint32_t vis_lo, vis_hi;
            {   // (counter and visited mask requested together: one LDS round trip less on the lag loop)
                const uint32_t seen_v = ((volatile lds_u32_t*)ctl)[5];
                uint32_t a = vis[0], b = vis[1];
                asm volatile("" ::: "memory");
                if (uniform(seen_v) < t - LM + 1u) {
                    if (!stage_wait_tight(ctl, 5, t - LM + 1u, 3)) break;
                    a = vis[0]; b = vis[1];
                }
                vis_lo = uniform(a); vis_hi = uniform(b);
            }
            TSQ_TRACE(4, t);
            const uint32_t hit_lo = tpm_lo & vis_lo, hit_hi = tpm_hi & vis_hi;
            const uint32_t q = hit_hi ? 63u - (uint32_t)__builtin_clz(hit_hi) : 31u - (uint32_t)__builtin_clz(hit_lo | 1u);
            if ((hit_lo | hit_hi) != 0u) tv = (((t - LM) << 6) + q) & 0xFFFFu;
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
        }
#ifdef TSQ_X_FAKE_CAND    // timing only (wrong streams): candidate bytes from the lane's own position (the line HASH loaded) -- what do the candidate gathers cost?
        else cb = ld128z(src, p, avail);
#else
        else cb = ld128z(src, cand0, avail);          // the lean layout (two blocks per CU) has no window: gather from L2
#endif
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
        const bool twin_l = (twin_in | twin_p1) != 0ull || tpw_any != 0u;
        lds_st2(arr + kGA, span_nat | (hard_l ? 0x100u : 0u) | (twin_l ? 0x200u : 0u) | (certain ? 0x400u : 0u) | (neart ? 0x800u : 0u) | (k0 << 16),
                cand0 | (nib << 24));
#ifdef TSQ_STATS
        if (t >= 3u) { st_[24] += (uint32_t)((uint32_t)__builtin_amdgcn_s_memtime() - uniform(ctl[40u + ((t - 3u) & 7u)])); st_[25] += 1; }
#endif
        TSQ_TRACE(5, t);
        TSQ_DELAY(3);
        stage_publish(ctl, parity ? kCtlMatchedOdd : 3u, t + 1u, lane);
        MREG_END(14);
        wbase = wbase + 128u >= StageCfg::WIN ? wbase + 128u - StageCfg::WIN : wbase + 128u;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0 && parity == 0u) { g_enc_stats[2] = st_[2]; g_enc_stats[3] = st_[3]; g_enc_stats[4] = TSQ_TOTAL(); g_enc_stats[13] = st_[13]; g_enc_stats[14] = st_[14]; g_enc_stats[35] = st_[20]; g_enc_stats[36] = st_[21]; g_enc_stats[51] = st_[24]; g_enc_stats[52] = st_[25]; }
#endif
}

// -------------------------------------------------------------------------------------------- COMMIT
// The position table's writer.  Tile by tile, as soon as WALK has published a tile's visited lanes: their positions go to the
// table (tsq_encode.cpp:79; among equal hashes the highest visited lane last) and into MATCH's filter for that tile (two filters,
// used in turn).  `posted` tells MATCH the filter is ready; `committed` -- after the stores have completed -- that the table is.
template <bool WINDOW>
__device__ __forceinline__ void stage_commit(uint32_t n, uint16_t* table, lds_u8_t* lds, uint32_t lane)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    const uint32_t n_tiles = (n >> 6) + 3u;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = 0; t < n_tiles; ++t) {
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane * 4u;
        // (requesting the tile's hashes and in-tile twins in front of this wait and polling without sleeping -- COMMIT is a member of the
        //  lag loop WALK -> COMMIT -> MATCH -> ORBIT -> WALK -- was measured: 39.4 ms against 39.3, nothing)
        if (!stage_wait(ctl, 5, t + 1u, 3)) break;
        TSQ_TRACE(12, t);
        const u32x4_t gb = lds_ld4(arr + kGB);
        const uint32_t h = gb.y, tin_lo = gb.z, tin_hi = gb.w;
        const uint64_t tw = __ballot((tin_lo | tin_hi) != 0u);                             // lanes with an earlier twin inside the tile
        lds_u32_t* visw = (lds_u32_t*)(rec + 2u);
        const uint64_t vis = (uint64_t)uniform(__hip_atomic_load(&visw[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) |
                             ((uint64_t)uniform(__hip_atomic_load(&visw[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) << 32);
        const uint32_t p = (t << 6) + lane;
        // among equal hashes the highest visited lane must win: lanes with an earlier twin store afterwards ...
#ifdef TSQ_X_FAKE_TABLE
#define TSQ_TBL(hh) ((hh) & 0xFFu)
#else
#define TSQ_TBL(hh) (hh)
#endif
        if (((vis & ~tw) >> lane) & 1ull) table[TSQ_TBL(h)] = (uint16_t)p;
        // ... one store per hash group: the highest visited lane of a group stores, its earlier twins are dropped unseen
        // (a block of equal bytes is ONE group of 64 lanes)
        uint64_t late = vis & tw;
        while (late) {
            const uint32_t top = msb64(late);
            if (lane == top) table[TSQ_TBL(h)] = (uint16_t)p;
            late &= ~((uint64_t)rdlane(tin_lo, top) | ((uint64_t)rdlane(tin_hi, top) << 32) | (1ull << top));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the table stores are complete
        TSQ_TRACE(6, t);
        TSQ_DELAY(4);
        stage_publish(ctl, kCtlCommitted, t + 1u, lane);
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[45] = st_[3] + st_[4]; g_enc_stats[46] = TSQ_TOTAL(); }
#endif
}

// --------------------------------------------------------------------------------------------- ORBIT
template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_orbit(uint32_t n, lds_u8_t* lds, uint32_t lane, uint32_t parity)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    constexpr uint32_t kDMin = EXT ? 128u : 64u;
    constexpr uint32_t LM = StageCfg::LM, LF = StageCfg::LF;
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;
    const uint32_t n_tiles = (n >> 6) + 3u;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    // two ORBIT wavefronts take the even and the odd tiles (nothing is carried from tile to tile)
    uint32_t scanned_seen = 0, parsed_seen = 0, near_seen = 0;
    for (uint32_t t = parity; t < n_tiles; t += 2u) {
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane * 4u;
        // ---- late classification of lanes whose only twins are in tiles t-LF .. t-LM+1.  By now the parser has (almost always)
        //      finished those tiles, so which of those twins were visited is known: the most recent visited one IS the
        //      candidate (tsq_encode.cpp:76-79), 65 bytes or more back.  Such a lane becomes an ordinary certain lane and
        //      never reaches the parser's scalar path.  (Lanes that also have nearer twins stay hazards.)
        //      All of this needs only SCAN's part of the record, so it runs while MATCH is still gathering tile t.
        uint32_t fix_sw = 0, fix_lw = 0;
        bool fix = false, clear_far = false;
        if (TSQ_LATE_FIX && t >= LF) {
            if (!stage_wait_seen(ctl, 2, t + 1u, scanned_seen, 6)) break;
            const u32x4_t gc = lds_ld4(arr + kGC);
            const u32x2_t gi = lds_ld2(arr + kGB + 2u);
            u32x2_t gd; gd.x = 0; gd.y = 0;
            if (LM == 4u) gd = lds_ld2(arr + kGD);
            // masks of the tiles WALK looks after, nearest first: tp[1] .. tp[LM-1]
            const uint32_t tp_lo[4] = {0u, gc.x, gc.z, gd.x}, tp_hi[4] = {0u, gc.y, gc.w, gd.y};
            uint32_t nearer = gi.x | gi.y, far = 0;
#pragma unroll
            for (uint32_t k = 1; k < LM; ++k) { if (k < LF) nearer |= tp_lo[k] | tp_hi[k]; else far |= tp_lo[k] | tp_hi[k]; }
            const uint32_t p = (t << 6) + lane;
            const bool only_far = far != 0u && nearer == 0u && p < tail_from;
            if (__ballot(only_far) != 0ull) {
                if (!stage_wait_seen(ctl, 5, t - LF + 1u, parsed_seen, 5)) break;              // the parser has finished tile t-LF
                uint32_t hit_lo = 0, hit_hi = 0, back = 0;
#pragma unroll
                for (uint32_t k = LM - 1u; k >= LF; --k) {                                     // (the nearest tile last: it wins)
                    lds_u32_t* vis = (lds_u32_t*)(recs + ((t - k) % StageCfg::R) * StageCfg::REC_WORDS + 2u);
                    const uint32_t v_lo = uniform(__hip_atomic_load(&vis[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const uint32_t v_hi = uniform(__hip_atomic_load(&vis[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const uint32_t h_lo = tp_lo[k] & v_lo, h_hi = tp_hi[k] & v_hi;
                    if ((h_lo | h_hi) != 0u) { hit_lo = h_lo; hit_hi = h_hi; back = k; }
                }
                clear_far = only_far;                    // visited or not, those tiles are settled for this lane
                if (only_far && (hit_lo | hit_hi) != 0u) {
                    const uint32_t q = hit_hi ? 63u - (uint32_t)__builtin_clz(hit_hi) : 31u - (uint32_t)__builtin_clz(hit_lo);
                    const uint32_t cand = (t << 6) - (back << 6) + q;
                    const uint32_t tq = t - back;
                    const u32x4_t a = *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u);
                    const u32x4_t b = *(volatile lds_u32x4_t*)(recs + (tq % StageCfg::R) * StageCfg::REC_WORDS + StageCfg::W16 + q * 4u);
                    const uint32_t k = prefix16(make_uint4(a.x, a.y, a.z, a.w), make_uint4(b.x, b.y, b.z, b.w));
                    // exact when the outcome cannot depend on the pair origin (as for the table's candidates) and, with
                    // extensions, when the first 16 bytes decide the length; otherwise the lane stays a hazard
                    if (p - cand >= kDMin && !(EXT && k >= 16u)) {
                        const bool eq4 = k >= 4u;
                        const uint32_t nib = length_nibble(eq4 ? k : 4u);
                        fix_sw = (eq4 ? nibble_span(nib) | 0x400u : 1u) | (k << 16);
                        fix_lw = cand | (nib << 24);
                        fix = true;
                    } else clear_far = false;
                }
            }
        }
        // (counter and record words requested together, as in WALK: one LDS round trip less on the lag loop)
        uint32_t sw, lw;
        {
            const uint32_t seen_v = ((volatile lds_u32_t*)ctl)[parity ? kCtlMatchedOdd : 3u];
            u32x2_t ga = lds_ld2(arr + kGA);
            asm volatile("" ::: "memory");
            if (uniform(seen_v) < t + 1u) {
                if (!stage_wait_tight(ctl, parity ? kCtlMatchedOdd : 3u, t + 1u, 6)) break;
                ga = lds_ld2(arr + kGA);
            }
            sw = ga.x; lw = ga.y;
        }
        TSQ_TRACE(11, t);
        if (clear_far) fix_sw |= 0x1000u;
        if (fix) { sw = fix_sw; lw = fix_lw; }
        else if (clear_far) sw |= 0x1000u;                          // (SCAN's masks stay as they are: later tiles inherit from them)
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
        // (WALK also reads NEAR's word of the tile: NEAR runs tiles ahead, this wait is satisfied by what was read long ago)
        if (!stage_wait_seen(ctl, kCtlNear, t + 1u, near_seen, 6)) break;
        // what WALK takes for an entry at this lane, ready to use: the lanes visited and where the walk halts (7 bits); a hard lane
        // halts on itself with nothing visited (the serial stage then needs no select for it)
        const bool hard_l = (sw & 0x100u) != 0u;
        const uint32_t orb_lo = hard_l ? 0u : (uint32_t)orb, orb_hi = hard_l ? 0u : (uint32_t)(orb >> 32);
        const uint32_t halt = hard_l ? lane : (nx & 0x7Fu);
        lds_st4(arr + kGA, sw | (halt << 24), lw, orb_lo, orb_hi);
#ifdef TSQ_STATS
        if (t >= 3u) { st_[24] += (uint32_t)((uint32_t)__builtin_amdgcn_s_memtime() - uniform(ctl[40u + ((t - 3u) & 7u)])); st_[25] += 1; }
#endif
        TSQ_TRACE(7, t);
        TSQ_DELAY(5);
        stage_publish(ctl, parity ? kCtlOrbitOdd : kCtlOrbitEven, t + 1u, lane);
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0 && parity == 0u) { g_enc_stats[6] = st_[6] + st_[5]; g_enc_stats[7] = TSQ_TOTAL(); g_enc_stats[53] = st_[24]; g_enc_stats[54] = st_[25]; }
#endif
}

// ------------------------------------------------------------------------------------- MATCH + ORBIT on one wavefront per tile
// Four wavefronts take the tiles t % 4 == which and do both stages' work for them (the same statements as stage_match and stage_orbit:
// what is said there is not repeated): no record store, counter, poll and reload between the two, and four tile periods per wavefront
// instead of two.  Measured against the two pairs above (encode kernel, ms, 239 / 256 blocks and the lean layout's 1 024): text without
// extensions 39.3 against 39.0 and 115.1 against 118.3; text with extensions 41.8 / 42.3 and 121.8 / 125.4; zeros with extensions
// **40.3 / 52.9 and 115.0 / 143.3** (every lane's match goes through the three extension rounds: two MATCH wavefronts are the bottleneck
// there); the 50 % mix 42.6 / 43.4 and 135.2 / 139.0; random bytes 41.3 / 41.6 and 168.7 / 167.6.  FusedMO picks it for every kernel but
// the one the headline runs (no extensions, window in LDS), where the pairs are 0.7 % faster.  The late classification runs behind
// MATCH's here (in front of the gather's use it costs text 0.5 ms).  (The two forms share no helper functions on purpose: the same
// statements moved into inlined helpers came out 0.45 ms slower for the pairs and 2.5 ms slower in the lean layout -- scheduling.)
#if defined(TSQ_FUSED_OFF)
template <bool EXT, bool WINDOW> struct FusedMO { static constexpr bool value = false; };
#elif defined(TSQ_FUSED_ALL)
template <bool EXT, bool WINDOW> struct FusedMO { static constexpr bool value = true; };
#else
template <bool EXT, bool WINDOW> struct FusedMO { static constexpr bool value = EXT || !WINDOW; };
#endif
// ORBIT's counter of tile t: one ctl word per even / odd tile, or one per t % 4
template <bool EXT, bool WINDOW>
__device__ __forceinline__ uint32_t orbit_word(uint32_t t)
{ return FusedMO<EXT, WINDOW>::value ? kCtlOrbitQ + (t & 3u) : ((t & 1u) ? kCtlOrbitOdd : kCtlOrbitEven); }
template <bool EXT, bool WINDOW>
__device__ __forceinline__ uint32_t orbit_word_after(uint32_t t)          // of tile t + 1
{ return FusedMO<EXT, WINDOW>::value ? kCtlOrbitQ + ((t + 1u) & 3u) : ((t & 1u) ? kCtlOrbitEven : kCtlOrbitOdd); }

template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_match_orbit(const uint8_t* src, uint64_t avail, uint32_t n, uint16_t* table, lds_u8_t* lds, uint32_t lane, uint32_t which)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    constexpr uint32_t kDMin = EXT ? 128u : 64u;
    constexpr uint32_t LM = StageCfg::LM, LF = StageCfg::LF;
    const uint32_t tail_from = n >= 5u ? n - 5u : 0u;
    const uint32_t n_tiles = (n >> 6) + 3u;
    uint32_t wbase = which << 6;                       // (t * 64) % WIN
    uint32_t scanned_seen = 0, committed_seen = 0, parsed_seen = 0, near_seen = 0;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif
    TSQ_BEGIN();
    for (uint32_t t = which; t < n_tiles; t += 4u) {
        if (!stage_wait_seen(ctl, 2, t + 1u, scanned_seen, 2)) break;
        TSQ_TRACE(3, t);
        volatile lds_u32_t* rec = recs + (t % StageCfg::R) * StageCfg::REC_WORDS;
        volatile lds_u32_t* arr = rec + StageCfg::ARR + lane * 4u;
        const u32x4_t gb = lds_ld4(arr + kGB);
        const u32x4_t wv = *(volatile lds_u32x4_t*)(rec + StageCfg::W16 + lane * 4u);
        const u32x4_t gc = lds_ld4(arr + kGC);
        const u32x4_t gd = lds_ld4(arr + kGD);
        const uint32_t h = gb.y;
        const uint4 w16 = make_uint4(wv.x, wv.y, wv.z, wv.w);
        const uint64_t twin_in = (uint64_t)gb.z | ((uint64_t)gb.w << 32);
        const uint64_t twin_p1 = (uint64_t)gc.x | ((uint64_t)gc.y << 32);
        const uint32_t tpw_any = LM == 4u ? (gc.z | gc.w | gd.x | gd.y) : (gc.z | gc.w);
        const uint32_t tpm_lo = LM == 4u ? gd.z : gd.x, tpm_hi = LM == 4u ? gd.w : gd.y;
        // ---- MATCH: the gather behind COMMIT(t - LM - 1), the patch behind WALK(t - LM) (see stage_match)
        if (t >= LM + 1u && !stage_wait_seen(ctl, kCtlCommitted, t - LM, committed_seen, 3)) break;
        TSQ_TRACE(10, t);
        const uint32_t tv_old = table[h];
        uint32_t tv = tv_old;
        if (t >= LM) {
            volatile lds_u32_t* vis = recs + ((t - LM) % StageCfg::R) * StageCfg::REC_WORDS + 2u;
            uint32_t vis_lo, vis_hi;
            {
                const uint32_t seen_v = ((volatile lds_u32_t*)ctl)[5];
                uint32_t a = vis[0], b = vis[1];
                asm volatile("" ::: "memory");
                if (uniform(seen_v) < t - LM + 1u) {
                    if (!stage_wait_tight(ctl, 5, t - LM + 1u, 3)) break;
                    a = vis[0]; b = vis[1];
                }
                vis_lo = uniform(a); vis_hi = uniform(b);
            }
            TSQ_TRACE(4, t);
            const uint32_t hit_lo = tpm_lo & vis_lo, hit_hi = tpm_hi & vis_hi;
            const uint32_t q = hit_hi ? 63u - (uint32_t)__builtin_clz(hit_hi) : 31u - (uint32_t)__builtin_clz(hit_lo | 1u);
            if ((hit_lo | hit_hi) != 0u) tv = (((t - LM) << 6) + q) & 0xFFFFu;
        }
        // ---- ORBIT: the late classification of the lanes whose only twins are in tiles t-LF .. t-LM+1 (see stage_orbit), from the
        //      masks this wavefront already holds
        uint32_t fix_sw = 0, fix_lw = 0;
        bool fix = false, clear_far = false;
        const uint32_t p = (t << 6) + lane;
        auto late_fix = [&]() -> bool {
        if (TSQ_LATE_FIX && t >= LF) {
            const uint32_t tp_lo[4] = {0u, gc.x, gc.z, gd.x}, tp_hi[4] = {0u, gc.y, gc.w, gd.y};
            uint32_t nearer = gb.z | gb.w, far = 0;
#pragma unroll
            for (uint32_t k = 1; k < LM; ++k) { if (k < LF) nearer |= tp_lo[k] | tp_hi[k]; else far |= tp_lo[k] | tp_hi[k]; }
            const bool only_far = far != 0u && nearer == 0u && p < tail_from;
            if (__ballot(only_far) != 0ull) {
                if (!stage_wait_seen(ctl, 5, t - LF + 1u, parsed_seen, 5)) return false;
                uint32_t hit_lo = 0, hit_hi = 0, back = 0;
#pragma unroll
                for (uint32_t k = LM - 1u; k >= LF; --k) {
                    lds_u32_t* vis = (lds_u32_t*)(recs + ((t - k) % StageCfg::R) * StageCfg::REC_WORDS + 2u);
                    const uint32_t v_lo = uniform(__hip_atomic_load(&vis[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const uint32_t v_hi = uniform(__hip_atomic_load(&vis[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const uint32_t h_lo = tp_lo[k] & v_lo, h_hi = tp_hi[k] & v_hi;
                    if ((h_lo | h_hi) != 0u) { hit_lo = h_lo; hit_hi = h_hi; back = k; }
                }
                clear_far = only_far;
                if (only_far && (hit_lo | hit_hi) != 0u) {
                    const uint32_t q = hit_hi ? 63u - (uint32_t)__builtin_clz(hit_hi) : 31u - (uint32_t)__builtin_clz(hit_lo);
                    const uint32_t cand = (t << 6) - (back << 6) + q;
                    const uint32_t tq = t - back;
                    const u32x4_t b = *(volatile lds_u32x4_t*)(recs + (tq % StageCfg::R) * StageCfg::REC_WORDS + StageCfg::W16 + q * 4u);
                    const uint32_t k = prefix16(w16, make_uint4(b.x, b.y, b.z, b.w));
                    if (p - cand >= kDMin && !(EXT && k >= 16u)) {
                        const bool eq4f = k >= 4u;
                        const uint32_t nibf = length_nibble(eq4f ? k : 4u);
                        fix_sw = (eq4f ? nibble_span(nibf) | 0x400u : 1u) | (k << 16);
                        fix_lw = cand | (nibf << 24);
                        fix = true;
                    } else clear_far = false;
                }
            }
        }
        return true;
        };
        const uint32_t cand0 = candidate_of(tv, p);
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
        }
        else cb = ld128z(src, cand0, avail);
        uint32_t k0 = prefix16(w16, cb);
        if (EXT) {
            uint32_t more = 16;
            if (__ballot(k0 == more) != 0ull) {
                const uint32_t scanned = uniform(__hip_atomic_load(&ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const bool in_window = WINDOW && scanned >= t + 2u;
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
        bool neart = false;
        if (__ballot((twin_in | twin_p1) != 0ull) != 0ull) {
            const uint64_t near_in = twin_in & ~below(lane >= 3u ? lane - 3u : 0u);
            const uint64_t near_prev = lane < 3u ? twin_p1 & ~below(61u + lane) : 0ull;
            neart = (near_in | near_prev) != 0ull && !tail;
        }
        const bool certain = eq4 && far_enough && !tail && !neart;
        const uint32_t nib = length_nibble(k0 < 4u ? 4u : k0);
        const uint32_t span_nat = certain ? nibble_span(nib) : 1u;
        const bool hard_m = (eq4 && !far_enough && dist >= 4u && !neart) || tail;
        const bool twin_l = (twin_in | twin_p1) != 0ull || tpw_any != 0u;
        uint32_t sw = span_nat | (hard_m ? 0x100u : 0u) | (twin_l ? 0x200u : 0u) | (certain ? 0x400u : 0u) | (neart ? 0x800u : 0u) | (k0 << 16);
        uint32_t lw = cand0 | (nib << 24);
        TSQ_TRACE(5, t);
        TSQ_TRACE(11, t);
        if (!late_fix()) break;
        if (fix) { sw = fix_sw; lw = fix_lw; }
        if (clear_far) sw |= 0x1000u;
        // ---- the whole orbit of every lane, by pointer doubling (see stage_orbit)
        const uint32_t self = lane | ((sw & 0x100u) ? 0x80u : 0u);
        const uint32_t c = lane + (sw & 0xFFu);
        const uint32_t there = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((c & 63u) << 2), (int)self);
        uint32_t nx = c >= 64u ? (c | 0x80u) : there;
        uint64_t orb = 1ull << lane;
#pragma unroll
        for (int round = 0; round < 6; ++round) {
            const int at = (int)((nx & 63u) << 2);
            const uint32_t nx2 = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)nx);
            const uint32_t olo = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)(uint32_t)orb);
            const uint32_t ohi = (uint32_t)__builtin_amdgcn_ds_bpermute(at, (int)(uint32_t)(orb >> 32));
            if ((nx & 0x80u) == 0u) { nx = nx2; orb |= (uint64_t)olo | ((uint64_t)ohi << 32); }
        }
        if (!stage_wait_seen(ctl, kCtlNear, t + 1u, near_seen, 6)) break;
        const bool hard_l = (sw & 0x100u) != 0u;
        const uint32_t orb_lo = hard_l ? 0u : (uint32_t)orb, orb_hi = hard_l ? 0u : (uint32_t)(orb >> 32);
        const uint32_t halt = hard_l ? lane : (nx & 0x7Fu);
        lds_st4(arr + kGA, sw | (halt << 24), lw, orb_lo, orb_hi);
        TSQ_TRACE(7, t);
        TSQ_DELAY(5);
        stage_publish(ctl, kCtlOrbitQ + (t & 3u), t + 1u, lane);
        wbase = wbase + 256u >= StageCfg::WIN ? wbase + 256u - StageCfg::WIN : wbase + 256u;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0 && which == 0u) { g_enc_stats[2] = st_[2]; g_enc_stats[3] = st_[3]; g_enc_stats[4] = TSQ_TOTAL(); g_enc_stats[6] = st_[6] + st_[5]; g_enc_stats[7] = TSQ_TOTAL(); }
#endif
}

// ---- uniform (SGPR) flag arithmetic for the parser wave.  Flags are 0/1 integers and every select is an explicit
//      s_cmp + s_cselect pair: left to itself the compiler keeps uniform booleans as 64-bit lane masks, selects through
//      `s_and_b64 exec` triples and converts them to integers through a VGPR (v_cndmask + v_readfirstlane, ~30 cycles).
__device__ __forceinline__ uint32_t s_sel(uint32_t c, uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(d) : "s"(c), "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint64_t s_sel64(uint32_t c, uint64_t a, uint64_t b)
{ uint64_t d; asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b64 %0, %2, %3" : "=s"(d) : "s"(c), "s"(a), "s"(b) : "scc"); return d; }
// two selects on one condition (one s_cmp), and selects straight on "x != 0" (no 0/1 flag in between)
__device__ __forceinline__ void s_sel_64_32(uint32_t c, uint64_t a1, uint64_t b1, uint32_t a2, uint32_t b2, uint64_t& d1, uint32_t& d2)
{ asm("s_cmp_lg_u32 %2, 0\n\ts_cselect_b64 %0, %3, %4\n\ts_cselect_b32 %1, %5, %6" : "=&s"(d1), "=&s"(d2) : "s"(c), "s"(a1), "s"(b1), "s"(a2), "s"(b2) : "scc"); }
__device__ __forceinline__ void s_selnz64_64_32(uint64_t x, uint64_t a1, uint64_t b1, uint32_t a2, uint32_t b2, uint64_t& d1, uint32_t& d2)
{ asm("s_cmp_lg_u64 %2, 0\n\ts_cselect_b64 %0, %3, %4\n\ts_cselect_b32 %1, %5, %6" : "=&s"(d1), "=&s"(d2) : "s"(x), "s"(a1), "s"(b1), "s"(a2), "s"(b2) : "scc"); }
__device__ __forceinline__ uint32_t s_selnz64(uint64_t x, uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, %2, %3" : "=s"(d) : "s"(x), "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_nz64(uint64_t x)
{ uint32_t d; asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(x) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_nz(uint32_t x)
{ uint32_t d; asm("s_min_u32 %0, %1, 1" : "=s"(d) : "s"(x) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_lt(uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_lt_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_ge(uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_ge_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_ne(uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_lg_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_eq(uint32_t a, uint32_t b)
{ uint32_t d; asm("s_cmp_eq_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(d) : "s"(a), "s"(b) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_offset_ok(uint32_t offset)          // tsq_encode.cpp:100
{ uint32_t d; asm("s_add_u32 %0, %1, -4\n\ts_cmp_lt_u32 %0, 0xfffb\n\ts_cselect_b32 %0, 1, 0" : "=&s"(d) : "s"(offset) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_msb64(uint64_t x)                   // x != 0
{ uint32_t d; asm("s_flbit_i32_b64 %0, %1\n\ts_xor_b32 %0, %0, 63" : "=s"(d) : "s"(x) : "scc"); return d; }
__device__ __forceinline__ uint32_t s_lsb64(uint64_t x)                   // x != 0
{ uint32_t d; asm("s_ff1_i32_b64 %0, %1" : "=s"(d) : "s"(x)); return d; }

// ------------------------------------------------------------------------------------ WALK + ACCOUNT
// The serial part of the parse, on two wavefronts.
//
//   WALK     decides which positions the parse VISITS -- the only thing MATCH (the position table) and the next tile need.  Per tile:
//            the orbit of the actual entry lane (three v_readlane), the check for visited twins, and for a hazard lane its candidate
//            (the most recent visited twin, or the gathered one) and common prefix.  A hazard whose candidate lies far enough back
//            that the pair origin cannot matter is decided on the spot and its class patched into the tile record; the others
//            (candidate right behind the last match, block tail) are QUERIES: ACCOUNT, which holds the symbol state, answers with
//            the next position.
//   ACCOUNT  the symbol accounting of tsq_encode.cpp:93-95,113-115,157-159 -- symbol count, pair origin, pending literal, the
//            reference-time origin inside literal runs -- in O(1) per segment from its masks and the tile record, the exact scalar
//            decision of the queries (tsq_encode.cpp:100,139-145), and the items for the BUILDER.  It trails WALK by an event or two.
//
// Events (4 words each, a single-producer/single-consumer ring in LDS): kEvSeg {base, V lo, V hi}: lanes of tile `base` the parse
// visited (everything since the tile's previous event; the classes of the hazard lanes among them that WALK decided are in the
// record by the time the event is pushed); kEvHaz {i, candidate, k | twin << 8}: a query; kEvEnd.

template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_walk(const uint8_t* src, uint64_t avail, uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* evq = (volatile lds_u32_t*)(lds + StageCfg::off_evq);
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);
    const uint32_t tail_from = uniform(n >= 5u ? n - 5u : 0u);
    constexpr uint32_t LM = StageCfg::LM, LF = StageCfg::LF;

    uint32_t ev_head = 0, ev_tail_seen = 0, n_query = 0;
    uint32_t v = 1, done = 0;
    uint32_t last_m = 0;                 // where the most recent match symbol starts: no pair origin lies before it
    uint64_t vall_p1 = 0, vall_p2 = 0, vall_p3 = 0;   // visited lanes of the previous tiles (LM-1 of them are looked at)
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif

    auto ev_wait_space = [&]() {
#ifdef TSQ_STATS
        const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
        while (ev_head - ev_tail_seen >= StageCfg::EQ) {
            ev_tail_seen = uniform(__hip_atomic_load(&ctl[kCtlEvTail], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            if (ev_head - ev_tail_seen >= StageCfg::EQ) { TSQ_SPIN_AT(ctl, 46u); __builtin_amdgcn_s_sleep(1); }
        }
#ifdef TSQ_STATS
        st_[9] += __builtin_amdgcn_s_memtime() - w0_;
#endif
    };
    auto ev_push = [&](uint32_t kind, uint32_t a, uint32_t b, uint32_t c) {
        if (ev_head - ev_tail_seen >= StageCfg::EQ) ev_wait_space();
        volatile lds_u32_t* e = evq + (ev_head % StageCfg::EQ) * StageCfg::EV_WORDS;
        uint32_t hv = kind;
        asm volatile("v_writelane_b32 %0, %1, 1" : "+v"(hv) : "s"(a));
        asm volatile("v_writelane_b32 %0, %1, 2" : "+v"(hv) : "s"(b));
        asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(hv) : "s"(c));
        if (lane < 4u) e[lane] = hv;
        TSQ_JIT(ev_head * 64u + 40u);
        TSQ_LDS_RELEASE();
        ev_head++;
        __hip_atomic_store(&ctl[kCtlEvHead], ev_head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // the 16 input bytes at a position of tiles t-2 .. t, from the tile records (one LDS address for the whole wave)
    auto words_at = [&](uint32_t slot, uint32_t pos) -> uint4 {
        const u32x4_t q = *(volatile lds_u32x4_t*)(recs + slot * StageCfg::REC_WORDS + StageCfg::W16 + (pos & 63u) * 4u);
        return make_uint4(q.x, q.y, q.z, q.w);
    };

    uint32_t wbase = 0;                // (t * 64) % WIN
    uint32_t rec_slot = 0;             // t % R
    // the tile record's words (loop-carried: the next tile's are requested at the end of a tile)
    uint32_t spanword, lane_word, nx, orb_lo, orb_hi, tin_lo, tin_hi, tp1_lo, tp1_hi, tp2r_lo, tp2r_hi, tp3r_lo = 0, tp3r_hi = 0, nearw, seen_v;
    auto load_record = [&](volatile lds_u32_t* arr) {
        const u32x4_t ga = lds_ld4(arr + kGA), gb = lds_ld4(arr + kGB), gc = lds_ld4(arr + kGC);      // three LDS instructions (four with LM = 4)
        spanword = ga.x; lane_word = ga.y; orb_lo = ga.z; orb_hi = ga.w;
        nearw = gb.x; tin_lo = gb.z; tin_hi = gb.w;
        tp1_lo = gc.x; tp1_hi = gc.y; tp2r_lo = gc.z; tp2r_hi = gc.w;
        if (LM == 4u) { const u32x2_t gd = lds_ld2(arr + kGD); tp3r_lo = gd.x; tp3r_hi = gd.y; }
        nx = spanword >> 24;
    };
    seen_v = ((volatile lds_u32_t*)ctl)[orbit_word<EXT, WINDOW>(0u)];
    load_record(recs + StageCfg::ARR + lane * 4u);
    asm volatile("" ::: "memory");
    TSQ_BEGIN();
    for (uint32_t t = 0; done == 0u; ++t, wbase = wbase + 64u == StageCfg::WIN ? 0u : wbase + 64u, rec_slot = rec_slot + 1u == StageCfg::R ? 0u : rec_slot + 1u) {
        const uint32_t base = t << 6;
        uint64_t vall = 0;
        // (the walk always enters the tile: a symbol spans at most 64 positions, so v <= base - 1 + 64 here -- no test, a branch costs
        //  the serial stage as much as five instructions whether it is taken or not)
        {
            REG_BEGIN(0); REG_END(0);
            REG_BEGIN(1);
            // (the serial stage polls without sleeping: a wake-up from s_sleep costs it up to 64 cycles per hand-off)
            const uint32_t orbit_at = orbit_word<EXT, WINDOW>(t);
            volatile lds_u32_t* arr = recs + rec_slot * StageCfg::REC_WORDS + StageCfg::ARR + lane * 4u;
            // The counter and the record's words were requested together at the end of the previous tile (in front of its publication's
            // bookkeeping): the LDS serves a wavefront's requests in order, so when the counter (asked for first) says the record is
            // there, the words that came back behind it are the record's; only when it was not there yet (the lag loop is late) are they
            // asked for again.
            if (uniform(seen_v) < t + 1u) {
#ifdef TSQ_STATS
                const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
                while (!stage_ready(ctl, orbit_at, t + 1u)) { TSQ_SPIN(ctl); }
#ifdef TSQ_STATS
                st_[8] += __builtin_amdgcn_s_memtime() - w0_;
#endif
                load_record(arr);
            }
            TSQ_CNT(15, 1);
            TSQ_TRACE(8, t);
            const uint32_t settled = (spanword & 0x1000u) ? 0u : 0xFFFFFFFFu;     // ORBIT has settled the lane's twins in tiles t-LF .. t-LM+1
            const uint32_t tp2_lo = LF <= 2u ? tp2r_lo & settled : tp2r_lo, tp2_hi = LF <= 2u ? tp2r_hi & settled : tp2r_hi;
            const uint32_t tp3_lo = tp3r_lo & settled, tp3_hi = tp3r_hi & settled;
            uint64_t handed = 0;                 // visited lanes already handed to ACCOUNT (segments pushed in front of queries, and the query lanes)
            const uint64_t near_m = __ballot((spanword & 0x800u) != 0u);
            const uint64_t certain_m = __ballot((spanword & 0x400u) != 0u);
            // a twin visited in the previous tiles: fixed for the whole tile (kept per lane, it joins the in-tile test)
            uint32_t prev_hit = (tp1_lo & (uint32_t)vall_p1) | (tp1_hi & (uint32_t)(vall_p1 >> 32)) |
                                (tp2_lo & (uint32_t)vall_p2) | (tp2_hi & (uint32_t)(vall_p2 >> 32));
            if (LM == 4u) prev_hit |= (tp3_lo & (uint32_t)vall_p3) | (tp3_hi & (uint32_t)(vall_p3 >> 32));
            const uint32_t k0 = (spanword >> 16) & 0xFFu;
            const uint32_t cand0 = lane_word & 0xFFFFFFu;

            uint32_t L = v - base;
            // One segment: the orbit from lane L and the first lane on it whose candidate is stale; sets L to that lane (a hazard lane) or
            // to where the walk leaves the tile.  The first segment of a tile is straight-line code: two tiles out of three end with it,
            // and every loop exit costs the serial stage a branch or two.
            auto segment = [&]() {
                REG_BEGIN(2);
                const uint32_t L0 = L;                                           // < 64
                // the orbit from L0: halts on a hard lane or past the tile (nothing at all if L0 itself is hard)
                // (ORBIT's words are ready to use: a hard entry lane has an empty orbit that halts on the lane itself)
                const uint32_t o_lo = rdlane(orb_lo, L0), o_hi = rdlane(orb_hi, L0);
                uint64_t V = (uint64_t)o_lo | ((uint64_t)o_hi << 32);
                L = rdlane(nx, L0);
                // the orbit treated twin lanes as ordinary lanes.  That is wrong for a visited lane that has a VISITED
                // twin before it (earlier in this tile, or in the two previous tiles): its gathered candidate is not
                // current.  The first such lane ends the segment; everything before it is exact.
                const uint64_t seen = vall | V;
                const uint32_t in_lo = tin_lo & (uint32_t)seen, in_hi = tin_hi & (uint32_t)(seen >> 32);
#ifdef TSQ_X_NOHAZ      // timing only (wrong streams): no lane with a visited twin stops the walk -- what do the hazard lanes cost the pipeline?
                uint64_t bad = 0;
#else
                uint64_t bad = __ballot((in_lo | in_hi | prev_hit) != 0u) & V;
#endif
                if (__builtin_expect((V & near_m) != 0ull, 0)) {
                    // near-twin lanes (classed "no match" on the assumption that a twin at most 3 back is visited) are
                    // the other way round: they are right exactly when such a twin was visited
                    uint32_t a_lo = in_lo, a_hi = in_hi, b_lo = tp1_lo, b_hi = tp1_hi;
                    asm volatile("; near twins" : "+v"(a_lo), "+v"(a_hi), "+v"(b_lo), "+v"(b_hi));   // keeps this block's arithmetic out of the tile prologue
                    const uint32_t pv_lo = b_lo & (uint32_t)vall_p1, pv_hi = b_hi & (uint32_t)(vall_p1 >> 32);
                    const bool has_in = (a_lo | a_hi) != 0u, has_prev = (pv_lo | pv_hi) != 0u;
                    const uint32_t nearest = a_hi ? 63u - (uint32_t)__builtin_clz(a_hi) : 31u - (uint32_t)__builtin_clz(a_lo | 1u);
                    const uint32_t nearest_prev = pv_hi ? 63u - (uint32_t)__builtin_clz(pv_hi) : 31u - (uint32_t)__builtin_clz(pv_lo | 1u);
                    const bool near_visited = (has_in && lane - nearest < 4u) || (has_prev && lane + 64u - nearest_prev < 4u);
                    bad = ((bad & ~near_m) | (near_m & ~__ballot(near_visited))) & V;
                }
                {
                    const uint32_t Lb = s_lsb64(bad | (1ull << 63));
                    s_selnz64_64_32(bad, V & ((1ull << Lb) - 1ull), V, Lb, L, V, L);
                    TSQ_CNT(23, bad != 0ull ? 1 : 0);
                }
                TSQ_CNT(24, 1);
                REG_END(2);
                REG_BEGIN(3);
                {
                    const uint64_t M = V & certain_m;
                    last_m = s_selnz64(M, base + s_msb64(M | 1ull), last_m);
                }
                vall |= V;
                v = base + L;
                REG_END(3);
            };
            auto hazard_lane = [&]() {
                REG_BEGIN(4);
                TSQ_CNT(26, 1);
                // ---- one hazard lane (hard, or with a visited twin): its candidate and common prefix
                const uint32_t i = base + L;
                uint32_t cand, k, twin_cand;
                // The lane's nearest twin, with the common prefix, is in NEAR's word.  If that twin was visited it IS the candidate
                // (the most recent visited position with the lane's hash): five hazard lanes out of six, one readlane and a bit test.
                const uint32_t nw = rdlane(nearw, L);
                const uint32_t nback = (nw >> 12) & 3u, nq = (nw >> 6) & 63u;
                const uint64_t nmask = s_sel64(nback, s_sel64(nback & 2u, s_sel64(nback & 1u, vall_p3, vall_p2), vall_p1), vall);
                if ((nw >> 15) & (uint32_t)(nmask >> nq) & 1u) {
                    cand = base - (nback << 6) + nq;
                    k = nw & 63u;
                    twin_cand = 1;
                    TSQ_CNT(16, 1);
                } else {
                    cand = rdlane(cand0, L);
                    k = rdlane(k0, L);
                    twin_cand = 0;
                    // visited twins of lane L: in this tile (before L) and in the two previous tiles;
                    // the most recent one is the candidate
                    const uint64_t in_tile = ((uint64_t)rdlane(tin_lo, L) | ((uint64_t)rdlane(tin_hi, L) << 32)) & vall;
                    const uint64_t in_p1 = ((uint64_t)rdlane(tp1_lo, L) | ((uint64_t)rdlane(tp1_hi, L) << 32)) & vall_p1;
                    const uint64_t in_p2 = ((uint64_t)rdlane(tp2_lo, L) | ((uint64_t)rdlane(tp2_hi, L) << 32)) & vall_p2;
                    const uint64_t in_p3 = LM == 4u ? ((uint64_t)rdlane(tp3_lo, L) | ((uint64_t)rdlane(tp3_hi, L) << 32)) & vall_p3 : 0ull;
                    if (in_tile | in_p1 | in_p2 | in_p3) {
                        const uint64_t pick = in_tile ? in_tile : in_p1 ? in_p1 : in_p2 ? in_p2 : in_p3;
                        const uint32_t back = in_tile ? 0u : in_p1 ? 64u : in_p2 ? 128u : 192u;
                        const uint32_t pick_lane = msb64(pick);
                        cand = base - back + pick_lane;
                        const uint32_t tiles_back = back >> 6;
                        const uint32_t cand_slot = rec_slot >= tiles_back ? rec_slot - tiles_back : rec_slot + StageCfg::R - tiles_back;
                        k = uniform(prefix16(words_at(rec_slot, i), words_at(cand_slot, cand)));
                        twin_cand = 1;
                        TSQ_CNT(17, in_tile ? 1 : 0); TSQ_CNT(18, (!in_tile && in_p1) ? 1 : 0); TSQ_CNT(19, (!in_tile && !in_p1) ? 1 : 0);
                        if (EXT && k >= 16u) {
                            // matches longer than 16 (tsq_encode.cpp:280-290).  (The bytes behind the first 16 come from the input window
                            // ring when SCAN has put everything up to i + 64 there.)
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
                    }
                }
                vall |= 1ull << L;
                REG_END(4);
                REG_BEGIN(5);
                // The lane decides like a certain lane when no pair origin the reference could use here lies closer to the candidate than
                // the match is long (tsq_encode.cpp:100,139-145: offset = origin - candidate must reach 4 and the match length): every
                // such origin -- the one of the scan-time test and the one after the pending literal -- is at or behind the start of the
                // most recent match symbol.  WALK then moves on at once and ACCOUNT patches the lane into the tile's masks.  Everything
                // else (a candidate right behind the last match, the block tail) waits for ACCOUNT's exact answer.
                const uint32_t need = k < 4u ? 4u : k;
#ifdef TSQ_X_FREE_QUERY     // timing only (wrong streams): every hazard lane in front of the block tail is decided on the spot -- what do the queries cost?
                const uint32_t local = s_lt(i, tail_from);
#else
                const uint32_t local = s_ge(last_m, cand + need) & s_lt(i, tail_from) & s_lt(i - cand, 0xFF00u);
#endif
                if (local) {
                    // the lane's class goes straight into the tile's record: ACCOUNT reads the record when the segment that holds the
                    // lane arrives (no event of its own: ACCOUNT is the busiest wavefront of the pipeline)
                    const uint32_t is_m = s_ge(k, 4u);
                    // (without extensions a match of k <= 16 bytes advances by k: mlen[k] = k - 1, tsq_encode.cpp:44-45,154)
                    const uint32_t m = EXT ? length_nibble(need) : need - 1u;
                    const uint32_t sp = s_sel(is_m, EXT ? nibble_span(m) : need, 1u);
                    if (lane == L) {
                        lds_st2(arr + kGA, (spanword & ~0x4FFu) | sp | (is_m << 10), cand | (m << 24));
                    }
                    v = i + sp;
                    last_m = s_sel(is_m, i, last_m);
                } else {
                    const uint64_t Vacc = vall & ~(handed | (1ull << L));
                    if (Vacc != 0ull) ev_push(kEvSeg, base, (uint32_t)Vacc, (uint32_t)(Vacc >> 32));
                    handed = vall;
                    ev_push(kEvHaz, i, cand, k | (twin_cand << 8));
                    n_query++;
                    TSQ_CNT(20, 1);
#ifdef TSQ_STATS
                    const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
                    while (!stage_ready(ctl, kCtlReplies, n_query)) { TSQ_SPIN_AT(ctl, 44u); }
#ifdef TSQ_STATS
                    st_[10] += __builtin_amdgcn_s_memtime() - w0_;
#endif
                    const uint32_t r = uniform(__hip_atomic_load(&ctl[kCtlReplyValue], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    v = r & 0x7FFFFFFFu;
                    done = r >> 31;
                    last_m = s_sel(s_ge(v - i, 4u), i, last_m);
                }
                L = v - base;
                REG_END(5);
            };
            segment();
            while (L < 64u) {
                hazard_lane();
                if (L >= 64u || done != 0u) break;
                segment();
            }
        }
        // ---- the tile's last segment goes to ACCOUNT, its visited mask to MATCH and COMMIT (they patch / commit the table), and the
        //      tile counter moves on: four stores of lane 0 in one stretch (the LDS executes them in this order)
        // ---- the tile's visited mask goes to MATCH and COMMIT (they patch / commit the table) and the tile counter moves on: the lag
        //      loop WALK(t) -> MATCH(t+3) -> ORBIT(t+3) -> WALK(t+3) waits for exactly this, so it goes out first; the tile's last
        //      segment for ACCOUNT follows
        REG_BEGIN(7);
        {
            lds_u32_t* vis = (lds_u32_t*)(recs + rec_slot * StageCfg::REC_WORDS + 2u);
            __hip_atomic_store(&vis[lane & 1u], (lane & 1u) ? (uint32_t)(vall >> 32) : (uint32_t)vall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        TSQ_DELAY(6);
        // (no event for the tile's last segment: ACCOUNT takes it from the visited mask above once this counter says the tile is walked)
        stage_publish(ctl, 5, t + 1u, lane);
        {   // the next tile's record is requested now: its words travel while the loop's bookkeeping runs
            const uint32_t next_slot = rec_slot + 1u == StageCfg::R ? 0u : rec_slot + 1u;
            seen_v = ((volatile lds_u32_t*)ctl)[orbit_word_after<EXT, WINDOW>(t)];
            load_record(recs + next_slot * StageCfg::REC_WORDS + StageCfg::ARR + lane * 4u);
            asm volatile("" ::: "memory");
        }
#ifdef TSQ_STATS
        if (lane == 0) ctl[40u + (t & 7u)] = (uint32_t)__builtin_amdgcn_s_memtime();       // (the lag loop is timed from here)
#endif
        TSQ_TRACE(9, t);
        vall_p3 = vall_p2; vall_p2 = vall_p1; vall_p1 = vall;
        REG_END(7);
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[8] = st_[8]; g_enc_stats[9] = st_[9]; g_enc_stats[10] = TSQ_TOTAL(); g_enc_stats[11] = st_[11]; g_enc_stats[12] = st_[12]; g_enc_stats[15] = st_[15]; for (int q = 23; q < 28; ++q) g_enc_stats[q] = st_[q]; g_enc_stats[29] = st_[17]; g_enc_stats[30] = st_[18]; g_enc_stats[31] = st_[19]; g_enc_stats[20] = st_[20]; g_enc_stats[40] = st_[10]; }
#endif
    __hip_atomic_store(&ctl[6], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ev_push(kEvEnd, 0u, 0u, 0u);
}

// All symbol state lives in uniform 32-bit integers (flags as 0/1, not bool: a bool that crosses a branch
// becomes a 64-bit lane mask and costs VALU round trips), and the common paths are straight-line selects:
// for a single wavefront a uniform branch costs more than the few instructions it skips.
template <bool EXT, bool WINDOW>
__device__ __forceinline__ void stage_account(uint32_t n, lds_u8_t* lds, uint32_t lane)
{
    using StageCfg = StageCfgT<WINDOW>;
    volatile lds_u32_t* queue = (volatile lds_u32_t*)(lds + StageCfg::off_queue);
    volatile lds_u32_t* evq = (volatile lds_u32_t*)(lds + StageCfg::off_evq);
    volatile lds_u32_t* recs = (volatile lds_u32_t*)(lds + StageCfg::off_rec);
    lds_u32_t* ctl = (lds_u32_t*)(lds + StageCfg::off_ctl);

    uint32_t head = 0, tail_seen = 0;  // tail_seen: last value read of the builder's progress (re-read only when the queue looks full)
    uint32_t nsym = 0, origin = 0, lit_from = 0;
    uint32_t am = 0;                   // 1 right after a match (tsq_encode.cpp:160-187), 0 inside a literal run
    uint32_t run0 = 0, origin_r0 = 0, odd_r0 = 0;   // where the current literal run started, the pair origin and symbol parity then
    uint32_t ev_tail = 0, n_reply = 0;
#ifdef TSQ_STATS
    unsigned long long st_[32] = {0};
#endif

    auto slot_begin = [&]() -> volatile lds_u32_t* {
        if (head - tail_seen >= StageCfg::Q) {
#ifdef TSQ_STATS
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            while (head - tail_seen >= StageCfg::Q) {
                tail_seen = uniform(__hip_atomic_load(&ctl[kCtlTail0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (StageCfg::DUAL_BUILDER) {          // (each BUILDER wavefront publishes the next item it will take: every item below both is consumed)
                    const uint32_t t1 = uniform(__hip_atomic_load(&ctl[kCtlTail1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    tail_seen = tail_seen < t1 ? tail_seen : t1;
                }
                if (head - tail_seen >= StageCfg::Q) { TSQ_SPIN_AT(ctl, 45u); __builtin_amdgcn_s_sleep(2); }
            }
#ifdef TSQ_STATS
            st_[9] += __builtin_amdgcn_s_memtime() - w0_;
#endif
        }
        return queue + (head % StageCfg::Q) * StageCfg::ITEM_WORDS;
    };
    auto slot_publish = [&]() {
        TSQ_JIT(head * 64u + 41u);
        TSQ_LDS_RELEASE();
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

    // the tile the events at hand belong to: its masks and per-lane words
    uint32_t base = 0xFFFFFFFFu;
    uint32_t span_nat = 0, lw = 0;
    uint64_t certain_m = 0;
    // The tile's symbols go to the builder as ONE item: visited lanes, which of them are matches, and the
    // per-lane candidate|nibble words -- the builder derives literal chunks, symbol indices and pair origins
    // from the masks.  Hazard lanes patch the masks; only the rare outcomes the masks cannot
    // express (a literal closed in front of a match that then fails, the block tail) are pushed explicitly,
    // after flushing what is pending.
    uint64_t Vt = 0, Mt = 0, qmask = 0;        // qmask: the lanes of the tile a query decided
    uint32_t e_nsym = 0, e_origin = 0, e_lit_from = 0;
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
            // (every lane stores: lanes 9 .. 15 put the kind word into the item's unused words, lanes above 15 into word 15 -- cheaper
            //  than masking the wavefront down to nine lanes: no exec save / restore, no branch around an empty mask)
            it[lane < 15u ? lane : 15u] = hv;
            it[16 + lane] = lw;
            slot_publish();
        }
        Vt = 0; Mt = 0;
        // what the next item starts from: the symbol state as it stands (nothing changes it before the next segment or query, and a
        // query that pushes a symbol of its own sets these itself)
        e_nsym = nsym; e_origin = origin; e_lit_from = lit_from;
    };
    // exact effect of a segment on the symbol state, one step per literal RUN or match (used when a run
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

    // WALK's progress reaches ACCOUNT two ways.  A tile's visited lanes: the mask in the tile's record once the "tiles walked" counter
    // has passed the tile (no event: the serial stage stores the mask for MATCH and COMMIT anyway).  Queries, each preceded by the lanes
    // visited so far in its tile (kEvSeg), through the event ring.  WALK pushes a tile's events before it publishes the tile, so: an
    // event of the tile at hand comes first; any other event means the tile at hand is complete.  The event counter and the tile
    // counter are read with ONE 8-byte load (a consistent pair), and the next event's words, the tile's mask and its class words are
    // requested in the same breath (the LDS serves a wavefront's requests in order: what comes back behind a counter that says "there"
    // is there) -- one LDS round trip per tile on this wavefront.
    uint32_t cur = 0, cur_slot = 0;           // the tile at hand and its record
    uint64_t got = 0;                         // its lanes already accounted: segments in front of queries, and the query lanes
    uint32_t opened = 0;                      // an event has opened it already
    // the tile's record: its class words (a later segment of a tile that a query opened takes the words again: WALK patches the
    // hazard lanes it decides itself into the class words, and it has done so for every lane of a segment before the segment's event
    // or the tile counter)
    auto open_tile = [&](const u32x2_t ga) {
        base = cur << 6;
        const uint32_t spanword = ga.x, fresh_lw = ga.y;
        lw = ((qmask >> lane) & 1ull) ? lw : fresh_lw;            // (a lane a query decided keeps the candidate ACCOUNT gave it)
        span_nat = spanword & 0xFFu;
        certain_m = __ballot((spanword & 0x400u) != 0u);
        TSQ_DELAY(7);
        TSQ_CNT(15, 1);
    };
    auto account_segment = [&](const uint64_t V) {
        // ---- the segment's effect on the symbol state, O(1) from its masks.  `dsym` symbols close (matches and
        //      the literal runs in front of them); the state afterwards hangs on the last match.
        // Two hand-written scalar blocks (ACCOUNT is one of the two wavefronts that pace the pipeline, and the compiler spends about
        // twice the instructions on these flag computations: every 0/1 flag through a compare of its own, phi copies at every merge):
        //   M = V & certain, N = V ^ M; L0 / Le lowest / highest visited lane, Lm highest match lane, fM lowest match lane;
        //   first_isN: the segment enters on a literal; slow: a literal run of 16 or more inside, or the entry run completes a
        //   16-byte chunk (then the segment is replayed step by step).
        uint64_t M, N, T, R;
        uint32_t L0, Le, Lm, fM, has_m, first_isN, slow, t1;
        asm volatile(
            "s_and_b64 %[M], %[V], %[C]\n\t"
            "s_cselect_b32 %[hasm], 1, 0\n\t"
            "s_xor_b64 %[N], %[V], %[M]\n\t"
            "s_ff1_i32_b64 %[L0], %[V]\n\t"
            "s_flbit_i32_b64 %[Le], %[V]\n\t"
            "s_flbit_i32_b64 %[Lm], %[M]\n\t"
            "s_ff1_i32_b64 %[fM], %[M]\n\t"
            "s_xor_b32 %[Le], %[Le], 63\n\t"
            "s_xor_b32 %[Lm], %[Lm], 63\n\t"
            "s_bitcmp1_b64 %[N], %[L0]\n\t"
            "s_cselect_b32 %[fN], 1, 0\n\t"
            "s_lshr_b64 %[T], %[N], 1\n\t"
            "s_and_b64 %[R], %[N], %[T]\n\t"
            "s_lshr_b64 %[T], %[R], 2\n\t"
            "s_and_b64 %[R], %[R], %[T]\n\t"
            "s_lshr_b64 %[T], %[R], 4\n\t"
            "s_and_b64 %[R], %[R], %[T]\n\t"
            "s_lshr_b64 %[T], %[R], 8\n\t"
            "s_and_b64 %[R], %[R], %[T]\n\t"
            "s_cselect_b32 %[slow], 1, 0\n\t"
            "s_add_i32 %[t1], %[Le], 1\n\t"
            "s_cmp_lg_u32 %[hasm], 0\n\t"
            "s_cselect_b32 %[t1], %[fM], %[t1]\n\t"
            "s_add_i32 %[t1], %[t1], %[base]\n\t"
            "s_sub_i32 %[t1], %[t1], %[lit]\n\t"
            "s_cmp_ge_u32 %[t1], 16\n\t"
            "s_cselect_b32 %[t1], %[fN], 0\n\t"
            "s_or_b32 %[slow], %[slow], %[t1]"
            : [M] "=&s"(M), [N] "=&s"(N), [T] "=&s"(T), [R] "=&s"(R), [L0] "=&s"(L0), [Le] "=&s"(Le), [Lm] "=&s"(Lm), [fM] "=&s"(fM),
              [hasm] "=&s"(has_m), [fN] "=&s"(first_isN), [slow] "=&s"(slow), [t1] "=&s"(t1)
            : [V] "s"(V), [C] "s"(certain_m), [base] "s"(base), [lit] "s"(lit_from)
            : "scc");
        if (__builtin_expect(slow, 0)) { TSQ_CNT(28, 1); replay_segment(V); }
        else {
            //   e_pos / lm_pos: where the segment enters / its last match starts, endm: where that match ends; last_m: the segment ends
            //   with it; pre: a pending literal closes in front of an entry match; dsym: symbols closed (tsq_encode.cpp:93-95,113-115,157-159)
            uint32_t sp, e_pos, lm_pos, endm, last_m, pre, d1, d2, par, t2, t3, new_run;
            asm volatile(
                "v_readlane_b32 %[sp], %[span], %[Lm]\n\t"
                "s_add_i32 %[epos], %[base], %[L0]\n\t"
                "s_add_i32 %[lmpos], %[base], %[Lm]\n\t"
                "s_cmp_eq_u32 %[Le], %[Lm]\n\t"
                "s_cselect_b32 %[lastm], %[hasm], 0\n\t"
                "s_cmp_lt_u32 %[lit], %[epos]\n\t"
                "s_cselect_b32 %[pre], 1, 0\n\t"
                "s_andn2_b32 %[pre], %[pre], %[fN]\n\t"
                "s_bcnt1_i32_b64 %[d1], %[M]\n\t"
                "s_lshl_b64 %[T], %[N], 1\n\t"
                "s_and_b64 %[T], %[T], %[M]\n\t"
                "s_bcnt1_i32_b64 %[d2], %[T]\n\t"
                "s_add_i32 %[d1], %[d1], %[d2]\n\t"
                "s_add_i32 %[d1], %[d1], %[pre]\n\t"
                "s_add_i32 %[endm], %[lmpos], %[sp]\n\t"
                "s_xor_b32 %[t2], %[lastm], 1\n\t"
                "s_cmp_lg_u32 %[hasm], 0\n\t"
                "s_cselect_b32 %[d1], %[d1], 0\n\t"
                "s_add_i32 %[nsym], %[nsym], %[d1]\n\t"
                "s_and_b32 %[par], %[nsym], 1\n\t"
                "s_cselect_b32 %[d2], %[lmpos], %[endm]\n\t"
                "s_cmp_lg_u32 %[hasm], 0\n\t"
                "s_cselect_b32 %[origin], %[d2], %[origin]\n\t"
                "s_cselect_b32 %[newrun], %[t2], %[am]\n\t"
                "s_cselect_b32 %[t3], %[endm], %[epos]\n\t"
                "s_cselect_b32 %[lit], %[endm], %[lit]\n\t"
                "s_cmp_lg_u32 %[newrun], 0\n\t"
                "s_cselect_b32 %[run0], %[t3], %[run0]\n\t"
                "s_cselect_b32 %[or0], %[origin], %[or0]\n\t"
                "s_cselect_b32 %[odd0], %[par], %[odd0]\n\t"
                "s_mov_b32 %[am], %[lastm]"
                : [sp] "=&s"(sp), [epos] "=&s"(e_pos), [lmpos] "=&s"(lm_pos), [endm] "=&s"(endm), [lastm] "=&s"(last_m), [pre] "=&s"(pre),
                  [d1] "=&s"(d1), [d2] "=&s"(d2), [par] "=&s"(par), [t2] "=&s"(t2), [t3] "=&s"(t3), [newrun] "=&s"(new_run), [T] "=&s"(T),
                  [nsym] "+s"(nsym), [origin] "+s"(origin), [lit] "+s"(lit_from), [am] "+s"(am), [run0] "+s"(run0), [or0] "+s"(origin_r0), [odd0] "+s"(odd_r0)
                : [span] "v"(span_nat), [M] "s"(M), [N] "s"(N), [L0] "s"(L0), [Le] "s"(Le), [Lm] "s"(Lm), [hasm] "s"(has_m), [fN] "s"(first_isN), [base] "s"(base)
                : "scc");
        }
        Vt |= V; Mt |= M;
    };

    TSQ_BEGIN();
    uint32_t kind, ea, eb, ec, walked, has_ev, ev_is_cur;
    u32x2_t ga;
    uint32_t vw;
    // counters, next event, the tile's visited mask and class words in one breath; waits until there is an event or a walked tile
    auto snapshot = [&]() {
        uint32_t w;
        for (;;) {
            volatile lds_u32_t* const e = evq + (ev_tail % StageCfg::EQ) * StageCfg::EV_WORDS;
            volatile lds_u32_t* const rec = recs + cur_slot * StageCfg::REC_WORDS;
            const u32x2_t hw = lds_ld2((volatile lds_u32_t*)ctl + kCtlEvHead);      // [4] events produced, [5] tiles walked
            w = e[lane & 3u];
            vw = rec[2u + (lane & 1u)];
            ga = lds_ld2(rec + StageCfg::ARR + lane * 4u + kGA);
            asm volatile("" ::: "memory");
            has_ev = s_ne(uniform(hw.x), ev_tail);
            walked = uniform(hw.y);
            // (flags as 0/1 integers from scalar compares: a C comparison that crosses a branch becomes a lane mask, a v_cndmask and a v_cmp)
            if (has_ev | s_ne(walked, cur)) break;
#ifdef TSQ_STATS
            const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
            TSQ_SPIN(ctl); __builtin_amdgcn_s_sleep(1);
#ifdef TSQ_STATS
            st_[8] += __builtin_amdgcn_s_memtime() - w0_;
#endif
        }
        kind = rdlane(w, 0); ea = rdlane(w, 1); eb = rdlane(w, 2); ec = rdlane(w, 3);
        ev_is_cur = has_ev & s_ne(kind, kEvEnd) & s_eq(ea >> 6, cur);
    };
    auto handle_event = [&]() {
        // ---- an event of the tile at hand
        ev_tail++;
        __hip_atomic_store(&ctl[kCtlEvTail], ev_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((opened ^ 1u) | (kind == kEvSeg ? 1u : 0u)) {
            if (opened == 0u) qmask = 0;
            open_tile(ga);
            opened = 1;
        }
        if (kind == kEvSeg) {
            const uint64_t V = (uint64_t)eb | ((uint64_t)ec << 32);
            account_segment(V);
            got |= V;
        } else {
            const uint32_t i = ea, cand = eb;
            uint32_t k = ec & 0xFFu;
            const uint32_t L = i - base;
            const uint64_t bit = 1ull << L;
            qmask |= bit;
            // ---- exact scalar resolution of one hazard lane (hard, or with a visited twin)
            uint32_t v = i + 1u, done = 0;
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
            __hip_atomic_store(&ctl[kCtlReplyValue], v | (done << 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            TSQ_LDS_RELEASE();
            n_reply++;
            __hip_atomic_store(&ctl[kCtlReplies], n_reply, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            got |= bit;
        }
        };
    auto finish_tile = [&]() {
        cur++;
        cur_slot = cur_slot + 1u == StageCfg::R ? 0u : cur_slot + 1u;
        got = 0; opened = 0;
        // (SCAN may reuse the records of the tiles before `cur`: everything ACCOUNT needs of them is in registers or in the queue)
        __hip_atomic_store(&ctl[kCtlAccounted], cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        TSQ_TRACE_ALL(13, cur - 1u);
    };
    // ---- The usual tile in one hand-written block (TSQ_ACCT_ASM): no event pending, the tile walked and not opened by an event, no
    // literal run that reaches a 16-byte chunk boundary, room in the item queue.  It does what the loop body below does for such a tile
    // -- requests the counters, the visited mask and the class words together, account_segment's two scalar blocks, the item's header
    // and lane words, the queue head, the tile counter -- in about a hundred instructions and four not-taken branches; the compiled
    // loop body spends about twice that (flags through lane masks, register copies where its paths merge, a branch pair per test).
    // Whatever it does not handle leaves it untouched (code != 0) and goes through the compiled path below.  ACCOUNT and WALK are the two
    // wavefronts that pace the pipeline.  s[60:83] and v[40:47] are its scratch registers.
// (instrumented builds keep it: `make stats` / `make spins` time and count the product's own path -- the waits are in snapshot(), which this
    //  block falls back to whenever it would have to wait; only TSQ_CNT's per-tile counters inside the compiled path skip the tiles it takes)
#if !defined(TSQ_NO_ACCT_ASM)
#define TSQ_ACCT_ASM 1
    const uint32_t lds0 = (uint32_t)(size_t)lds;
    const uint32_t a_ctl = lds0 + StageCfg::off_ctl;                                                   // ctl[0]
    const uint32_t a_vis = lds0 + StageCfg::off_rec + 8u + (lane & 1u) * 4u;                           // record word 2 / 3
    const uint32_t a_ga = lds0 + StageCfg::off_rec + StageCfg::ARR * 4u + lane * 16u;                  // group A of this lane
    const uint32_t a_hdr = lds0 + StageCfg::off_queue + (lane < 15u ? lane : 15u) * 4u;                // item header word
    const uint32_t a_lw = lds0 + StageCfg::off_queue + 64u + lane * 4u;                                // item lane word
#endif
    for (;;) {
#ifdef TSQ_ACCT_ASM
        {
            uint32_t code;
            TSQ_JIT(head * 64u + 41u);
            TSQ_DELAY(7);
            asm volatile(
                "s_mul_i32 s61, %[slot], %[recb]\n\t"
                "s_mov_b32 %[code], 1\n\t"
                "v_add_u32_e32 v41, s61, %[avis]\n\t"
                "v_add_u32_e32 v42, s61, %[aga]\n\t"
                "v_mov_b32_e32 v40, %[actl]\n\t"
                "ds_read_b64 v[44:45], v40 offset:16\n\t"          // ctl[4] events produced, ctl[5] tiles walked
                "ds_read_b32 v43, v41\n\t"                         // visited mask: lane & 1 selects the word
                "ds_read_b64 v[46:47], v42\n\t"                    // spanword, lane word
                "s_sub_u32 s64, %[head], %[tseen]\n\t"
                "s_lshl_b32 s82, %[cur], 6\n\t"                    // base
                "s_waitcnt lgkmcnt(2)\n\t"
                "v_readfirstlane_b32 s62, v44\n\t"
                "v_readfirstlane_b32 s63, v45\n\t"
                "s_cmp_lg_u32 s62, %[evt]\n\t"
                "s_cbranch_scc1 9f\n\t"                            // an event is pending
                "s_cmp_eq_u32 s63, %[cur]\n\t"
                "s_cbranch_scc1 9f\n\t"                            // the tile is not walked yet
                "s_cmp_ge_u32 s64, %[Q]\n\t"
                "s_cbranch_scc1 9f\n\t"                            // the item queue looks full
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_readlane_b32 s64, v43, 0\n\t"
                "v_readlane_b32 s65, v43, 1\n\t"                   // V = s[64:65]
                "v_bfe_u32 v41, v46, 10, 1\n\t"
                "v_and_b32_e32 v40, 0xff, v46\n\t"                 // natural span
                "v_cmp_ne_u32_e64 s[66:67], 0, v41\n\t"            // certain match lanes
                "s_cmp_eq_u64 s[64:65], 0\n\t"
                "s_cbranch_scc1 9f\n\t"
                // ---- account_segment, first block: M s[68:69], N s[70:71], T s[72:73], R s[74:75], L0 s76, Le s77, Lm s78, fM s79, has_m s80, first_isN s81
                "s_and_b64 s[68:69], s[64:65], s[66:67]\n\t"
                "s_cselect_b32 s80, 1, 0\n\t"
                "s_xor_b64 s[70:71], s[64:65], s[68:69]\n\t"
                "s_ff1_i32_b64 s76, s[64:65]\n\t"
                "s_flbit_i32_b64 s77, s[64:65]\n\t"
                "s_flbit_i32_b64 s78, s[68:69]\n\t"
                "s_ff1_i32_b64 s79, s[68:69]\n\t"
                "s_xor_b32 s77, s77, 63\n\t"
                "s_xor_b32 s78, s78, 63\n\t"
                "s_bitcmp1_b64 s[70:71], s76\n\t"
                "s_cselect_b32 s81, 1, 0\n\t"
                "s_lshr_b64 s[72:73], s[70:71], 1\n\t"
                "s_and_b64 s[74:75], s[70:71], s[72:73]\n\t"
                "s_lshr_b64 s[72:73], s[74:75], 2\n\t"
                "s_and_b64 s[74:75], s[74:75], s[72:73]\n\t"
                "s_lshr_b64 s[72:73], s[74:75], 4\n\t"
                "s_and_b64 s[74:75], s[74:75], s[72:73]\n\t"
                "s_lshr_b64 s[72:73], s[74:75], 8\n\t"
                "s_and_b64 s[74:75], s[74:75], s[72:73]\n\t"
                "s_cbranch_scc1 6f\n\t"                            // a literal run of 16 or more inside: below
                "s_add_i32 s60, s77, 1\n\t"
                "s_cmp_lg_u32 s80, 0\n\t"
                "s_cselect_b32 s60, s79, s60\n\t"
                "s_add_i32 s60, s60, s82\n\t"
                "s_sub_i32 s60, s60, %[lit]\n\t"
                "s_cmp_ge_u32 s60, 16\n\t"
                "s_cselect_b32 s60, s81, 0\n\t"
                "s_cmp_lg_u32 s60, 0\n\t"
                "s_cbranch_scc1 9f\n\t"                            // the entry run completes a 16-byte chunk
                // ---- the item's header, from the state as it stands at the tile's entry
                "v_mov_b32_e32 v41, 1\n\t"                         // kItemSeg
                "v_readlane_b32 s61, v40, s78\n\t"                 // span of the last match lane
                "v_writelane_b32 v41, s82, 1\n\t"
                "s_add_i32 s62, s82, s76\n\t"                      // e_pos
                "v_writelane_b32 v41, s64, 2\n\t"
                "s_add_i32 s63, s82, s78\n\t"                      // lm_pos
                "v_writelane_b32 v41, s65, 3\n\t"
                "s_cmp_eq_u32 s77, s78\n\t"
                "v_writelane_b32 v41, %[nsym], 4\n\t"
                "s_cselect_b32 s83, s80, 0\n\t"                    // last_m
                "v_writelane_b32 v41, %[origin], 5\n\t"
                "s_cmp_lt_u32 %[lit], s62\n\t"
                "v_writelane_b32 v41, %[lit], 6\n\t"
                "s_cselect_b32 s60, 1, 0\n\t"
                "v_writelane_b32 v41, s68, 7\n\t"
                "s_andn2_b32 s60, s60, s81\n\t"                    // pre
                "v_writelane_b32 v41, s69, 8\n\t"
                // ---- account_segment, second block
                "s_bcnt1_i32_b64 s66, s[68:69]\n\t"
                "s_lshl_b64 s[72:73], s[70:71], 1\n\t"
                "s_and_b64 s[72:73], s[72:73], s[68:69]\n\t"
                "s_bcnt1_i32_b64 s67, s[72:73]\n\t"
                "s_add_i32 s66, s66, s67\n\t"
                "s_add_i32 s66, s66, s60\n\t"                      // dsym
                "s_add_i32 s61, s63, s61\n\t"                      // endm
                "s_xor_b32 s67, s83, 1\n\t"
                "s_cmp_lg_u32 s80, 0\n\t"
                "s_cselect_b32 s66, s66, 0\n\t"
                "s_add_i32 %[nsym], %[nsym], s66\n\t"
                "s_and_b32 s66, %[nsym], 1\n\t"                    // parity; SCC = odd
                "s_cselect_b32 s60, s63, s61\n\t"
                "s_cmp_lg_u32 s80, 0\n\t"
                "s_cselect_b32 %[origin], s60, %[origin]\n\t"
                "s_cselect_b32 s67, s67, %[am]\n\t"                // new_run
                "s_cselect_b32 s60, s61, s62\n\t"
                "s_cselect_b32 %[lit], s61, %[lit]\n\t"
                "s_cmp_lg_u32 s67, 0\n\t"
                "s_cselect_b32 %[run0], s60, %[run0]\n\t"
                "s_cselect_b32 %[or0], %[origin], %[or0]\n\t"
                "s_cselect_b32 %[odd0], s66, %[odd0]\n\t"
                "s_mov_b32 %[am], s83\n\t"
                "s_branch 7f\n"
                // ---- a literal run of 16 or more.  Taken here when the whole segment is ONE run of literals and nothing else (incompressible
                //      data: every lane visited, no match); any other shape goes to the compiled path, which replays the segment run by run.
                "6:\n\t"
                "s_cmp_lg_u32 s80, 0\n\t"
                "s_cbranch_scc1 9f\n\t"
                "s_lshr_b64 s[72:73], s[64:65], s76\n\t"
                "s_add_u32 s60, s72, 1\n\t"
                "s_addc_u32 s61, s73, 0\n\t"
                "s_and_b64 s[60:61], s[60:61], s[72:73]\n\t"       // zero when the visited lanes are consecutive
                "s_cbranch_scc1 9f\n\t"
                "v_mov_b32_e32 v41, 1\n\t"                         // kItemSeg
                "s_bcnt1_i32_b64 s62, s[64:65]\n\t"                // the run's length
                "v_writelane_b32 v41, s82, 1\n\t"
                "s_add_i32 s63, s82, s76\n\t"                      // where it starts
                "v_writelane_b32 v41, s64, 2\n\t"
                "s_and_b32 s60, %[nsym], 1\n\t"
                "v_writelane_b32 v41, s65, 3\n\t"
                "s_cmp_lg_u32 %[am], 0\n\t"                        // right behind a match: a new literal run starts here
                "v_writelane_b32 v41, %[nsym], 4\n\t"
                "s_cselect_b32 %[run0], s63, %[run0]\n\t"
                "v_writelane_b32 v41, %[origin], 5\n\t"
                "s_cselect_b32 %[or0], %[origin], %[or0]\n\t"
                "v_writelane_b32 v41, %[lit], 6\n\t"
                "s_cselect_b32 %[odd0], s60, %[odd0]\n\t"
                "v_writelane_b32 v41, s68, 7\n\t"
                "s_mov_b32 %[am], 0\n\t"
                "v_writelane_b32 v41, s69, 8\n\t"
                "s_add_i32 s60, s63, s62\n\t"
                "s_sub_i32 s60, s60, %[lit]\n\t"
                "s_lshr_b32 s60, s60, 4\n\t"                       // 16-byte chunks that complete inside the run (tsq_encode.cpp:82-97)
                "s_add_i32 %[nsym], %[nsym], s60\n\t"
                "s_lshl_b32 s61, s60, 4\n\t"
                "s_add_i32 s61, s61, %[lit]\n\t"                   // the pending literal starts here afterwards
                "s_add_i32 s62, s61, -16\n\t"
                "s_cmp_ge_u32 s60, 2\n\t"
                "s_cselect_b32 s62, s62, %[origin]\n\t"            // odd count: the chunk before the last closed a pair (if there are two)
                "s_bitcmp0_b32 %[nsym], 0\n\t"
                "s_cselect_b32 s62, s61, s62\n\t"                  // even count: the last chunk closed a pair
                "s_cmp_lg_u32 s60, 0\n\t"
                "s_cselect_b32 %[origin], s62, %[origin]\n\t"
                "s_mov_b32 %[lit], s61\n"
                // ---- the item: header word per lane, lane words, then the queue head; the tile counter
                "7:\n\t"
                "s_and_b32 s60, %[head], %[Qm]\n\t"
                "s_mulk_i32 s60, %[itemb]\n\t"
                "v_add_u32_e32 v42, s60, %[ahdr]\n\t"
                "v_add_u32_e32 v43, s60, %[alw]\n\t"
                "s_add_i32 %[head], %[head], 1\n\t"
                "ds_write_b32 v42, v41\n\t"
                "ds_write_b32 v43, v47\n\t"
                "v_mov_b32_e32 v40, %[actl]\n\t"
                "v_mov_b32_e32 v42, %[head]\n\t"
                "s_add_i32 %[cur], %[cur], 1\n\t"
                "s_add_i32 %[slot], %[slot], 1\n\t"
                "ds_write_b32 v40, v42\n\t"                        // ctl[0]: items published
                "v_mov_b32_e32 v43, %[cur]\n\t"
                "s_cmp_lg_u32 %[slot], %[R]\n\t"
                "s_cselect_b32 %[slot], %[slot], 0\n\t"
                "ds_write_b32 v40, v43 offset:56\n\t"              // ctl[14]: tiles accounted
                "s_mov_b32 %[code], 0\n"
                "9:"
                : [code] "=&s"(code), [nsym] "+s"(nsym), [origin] "+s"(origin), [lit] "+s"(lit_from), [am] "+s"(am), [run0] "+s"(run0),
                  [or0] "+s"(origin_r0), [odd0] "+s"(odd_r0), [head] "+s"(head), [cur] "+s"(cur), [slot] "+s"(cur_slot)
                : [evt] "s"(ev_tail), [tseen] "s"(tail_seen + opened * 0x40000000u), [actl] "s"(a_ctl), [avis] "v"(a_vis), [aga] "v"(a_ga), [ahdr] "v"(a_hdr), [alw] "v"(a_lw),
                  [recb] "n"(StageCfg::REC_WORDS * 4u), [Q] "n"(StageCfg::Q), [Qm] "n"(StageCfg::Q - 1u), [itemb] "n"(StageCfg::ITEM_WORDS * 4u), [R] "n"(StageCfg::R)
                : "scc", "memory", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76",
                  "s77", "s78", "s79", "s80", "s81", "s82", "s83", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
            if (code == 0u) { e_nsym = nsym; e_origin = origin; e_lit_from = lit_from; TSQ_TRACE_ALL(13, cur - 1u); continue; }
        }
#endif
        snapshot();
        // (a tile's events -- queries and the lanes visited in front of them -- are rare; the usual tile goes straight through below)
        while (__builtin_expect(ev_is_cur != 0u, 0)) { handle_event(); snapshot(); }
        if (walked == cur) break;                     // nothing walked is left and the next event is not this tile's: kEvEnd
        // ---- the tile at hand is complete: its remaining visited lanes are its last segment
        const uint64_t vis = (uint64_t)rdlane(vw, 0) | ((uint64_t)rdlane(vw, 1) << 32);
        const uint64_t V = vis & ~got;
        qmask = opened ? qmask : 0ull;
        open_tile(ga);
        if (V != 0ull) account_segment(V);
        flush_pending();                              // the tile's item goes to the builder
        finish_tile();
    }
    flush_pending();
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { g_enc_stats[41] = st_[8]; g_enc_stats[42] = st_[9]; g_enc_stats[43] = TSQ_TOTAL(); g_enc_stats[16] = nsym; g_enc_stats[21] = st_[21]; g_enc_stats[22] = st_[22]; g_enc_stats[28] = st_[28]; g_enc_stats[44] = st_[15]; }
#endif
    {
        volatile lds_u32_t* it = slot_begin();
        if (lane == 0) { it[0] = kItemEnd; it[4] = nsym; }
        slot_publish();
    }
}

// (amdgpu_num_sgpr: left to itself the compiler takes 105 scalar registers; capped at 96 -- a few cold values go to lanes of a
//  vector register -- the kernel is 0.4 ms faster at 239 blocks: 88 40.5 ms, 80 41.0, 72 41.6, no cap 40.6.  Either way seven wave slots
//  per SIMD: two workgroups share a CU with twelve wavefronts each at most; with 80 there are eight, and two workgroups of fourteen
//  fit -- the lean layout is no faster for it, 36.2 GB/s both ways.)
template <bool EXT, bool WINDOW>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_sgpr(96))) void enc_stage_kernel(const uint8_t* __restrict__ in, uint64_t n_total, uint64_t readable, uint64_t stride,
                                                        uint8_t* __restrict__ slots, uint32_t* __restrict__ sizes,
                                                        uint16_t* __restrict__ tables, int32_t* __restrict__ status)
{
    using StageCfg = StageCfgT<WINDOW>;
    extern __shared__ __attribute__((aligned(16))) uint8_t stage_lds[];
    const uint32_t b = blockIdx.x, lane = threadIdx.x & 63u;
    // Wave w of a workgroup runs on SIMD w % 4 (read back from HW_ID in the instrumented build).  WALK, the serial stage, gets a SIMD
    // almost to itself (NEAR, a light stage, shares it): the two other wavefronts of SIMD 0 leave right after the prologue.
    //   SIMD 0: WALK, NEAR      SIMD 1: ORBIT even, MATCH even, HASH, BUILDER     SIMD 2: ORBIT odd, MATCH odd, TWINS, EMIT     SIMD 3: ACCOUNT, COMMIT, IN, BUILDER 2
    // (measured: ACCOUNT next to WALK on SIMD 0 costs 8 to 13 ms -- WALK polls without sleeping and starves the wavefront whose
    //  answers it waits for; a third ORBIT wavefront +1.4 ms; s_setprio, COMMIT / EMIT / HASH on SIMD 0, NEAR on SIMD 2 or 3: within 0.3 %)
    enum : uint32_t { kRoleWalk, kRoleOrbit0, kRoleOrbit1, kRoleAccount, kRoleNone, kRoleMatch0, kRoleMatch1, kRoleBuilder, kRoleHash, kRoleTwins, kRoleEmit, kRoleCommit, kRoleNear, kRoleIn, kRoleBuilder1 };
    // (the lean layout -- two workgroups per CU -- launches its twelve working wavefronts only: 2 x 13 already do not fit a CU's wave slots)
    constexpr uint32_t role_map[16] = { kRoleWalk, kRoleOrbit0, kRoleOrbit1, kRoleAccount, kRoleNear, kRoleMatch0, kRoleMatch1, kRoleCommit,
                                        kRoleNone, kRoleHash, kRoleTwins, kRoleIn, kRoleNone, kRoleBuilder, kRoleEmit, kRoleBuilder1 };
    // (lean layout, measured at 4 GiB: this placement 30.7 GB/s on text against 29.4 for the round's earlier one)
    constexpr uint32_t role_map_lean[16] = { kRoleWalk, kRoleOrbit0, kRoleOrbit1, kRoleAccount, kRoleNear, kRoleMatch0, kRoleMatch1, kRoleBuilder,
                                             kRoleCommit, kRoleHash, kRoleTwins, kRoleEmit, kRoleNone, kRoleNone, kRoleNone, kRoleNone };
    uint32_t role = kRoleNone;
#pragma unroll
    for (uint32_t w = 0; w < 16u; ++w) role = (threadIdx.x >> 6) == w ? (WINDOW ? role_map[w] : role_map_lean[w]) : role;
    role = uniform(role);
    // block b of the launch lies at in + b * stride (stride = 4 MiB: one contiguous buffer; larger: a shard's blocks, each
    // followed by its own look-ahead bytes); its length follows from the virtual total n_total = (blocks - 1) * 4 MiB + last
    const uint64_t start = (uint64_t)b * stride;
    const uint64_t avail = readable - start;
    const uint64_t vstart = (uint64_t)b << kBlockBits;
    const uint32_t n = n_total - vstart < kBlockSize ? (uint32_t)(n_total - vstart) : kBlockSize;
    const uint8_t* src = in + start;
    uint8_t* out = slots + (size_t)b * kSlotSize;
    uint16_t* table = tables + (size_t)b * kHashEntries;

    {   // tsqInit (tsq_context.cpp:77-80), every wavefront of the workgroup
        uint4* t4 = reinterpret_cast<uint4*>(table);
        for (uint32_t k = threadIdx.x; k < kHashEntries * 2 / 16; k += blockDim.x) t4[k] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x < 64) reinterpret_cast<uint32_t*>(stage_lds + StageCfg::off_ctl)[threadIdx.x] = 0;
        uint4* o4 = reinterpret_cast<uint4*>(stage_lds + StageCfg::off_owner);          // owner image: no valid entries
        for (uint32_t k = threadIdx.x; k < (StageCfg::OWN_MASK + 1u) / 16; k += blockDim.x) o4[k] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) { out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16); }
    }
    __syncthreads();
    lds_u8_t* lds3 = (lds_u8_t*)stage_lds;
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && lane == 0) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); g_enc_trace[15 * 256 + (threadIdx.x >> 6)] = id; }
#endif
    if (role == kRoleWalk) stage_walk<EXT, WINDOW>(src, avail, n, lds3, lane);
    else if (role == kRoleAccount) stage_account<EXT, WINDOW>(n, lds3, lane);
    else if (role == kRoleCommit) stage_commit<WINDOW>(n, table, lds3, lane);
    else if (role == kRoleHash) stage_hash<WINDOW>(src, avail, n, lds3, lane, table);
    else if (role == kRoleTwins) stage_twins<WINDOW>(n, lds3, lane);
    else if (role == kRoleIn) stage_in<WINDOW>(n, lds3, lane);
    else if (role == kRoleNear) stage_near<EXT, WINDOW>(n, lds3, lane);
    else if (!FusedMO<EXT, WINDOW>::value && (role == kRoleMatch0 || role == kRoleMatch1)) stage_match<EXT, WINDOW>(src, avail, n, table, lds3, lane, role == kRoleMatch1 ? 1u : 0u);
    else if (!FusedMO<EXT, WINDOW>::value && (role == kRoleOrbit0 || role == kRoleOrbit1)) stage_orbit<EXT, WINDOW>(n, lds3, lane, role == kRoleOrbit1 ? 1u : 0u);
    else if (role == kRoleEmit) stream_emitter<StageCfg>(src, avail, out, lds3, lane, b, sizes, status);
    else if (role == kRoleBuilder) stream_builder<StageCfg>(lds3, lane, 0u);
    else if (role == kRoleBuilder1) stream_builder<StageCfg>(lds3, lane, 1u);
    // (the fused form last in the chain, the pairs where they always were: the order of this chain decides where the compiler lays the
    //  stages' code, and the headline kernel's layout is worth 0.4 ms)
    else if (FusedMO<EXT, WINDOW>::value && (role == kRoleOrbit0 || role == kRoleOrbit1 || role == kRoleMatch0 || role == kRoleMatch1))
        stage_match_orbit<EXT, WINDOW>(src, avail, n, table, lds3, lane, role == kRoleOrbit0 ? 0u : role == kRoleOrbit1 ? 1u : role == kRoleMatch0 ? 2u : 3u);   // (consecutive tiles on different SIMDs: 1, 2, 1, 2)
#ifdef TSQ_SPINS
    if (blockIdx.x == 0 && lane == 0) g_enc_spins[threadIdx.x >> 6] = reinterpret_cast<uint32_t*>(stage_lds + StageCfg::off_ctl)[48u + (threadIdx.x >> 6)];
    if (blockIdx.x == 0 && role == kRoleEmit && lane == 0) for (uint32_t q = 0; q < 4u; ++q) g_enc_spins[16u + q] = reinterpret_cast<uint32_t*>(stage_lds + StageCfg::off_ctl)[44u + q];   // (EMIT leaves last)
#endif
}

}  // namespace tsq
