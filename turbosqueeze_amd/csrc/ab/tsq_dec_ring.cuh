// tsq_dec_ring.cuh -- wave-parallel block decoder with an LDS history ring, byte-granular copies (A/B variants 8 and 9: superseded by
// tsq_dec_sym.cuh, not in the product library).
//
// Phase structure: speculative group parse at every offset, pointer doubling, chain follow, group
// scan, symbol records, pointer-jumping copy resolution (the first parallel decoder, now
// tsq_dec_fast.cuh, had the same phases), with the two things that one's profile asked for:
//   * the last 64 KiB of output live in an LDS ring.  Match sources reach at most 65534 bytes
//     before the start of their symbol pair (tsq_decode.cpp:73), i.e. never further back than the
//     ring, so history bytes are read from LDS at byte granularity instead of gathering 16 bytes
//     (a 128-byte line each) from L2 -- that gather was 25 % of the old kernel's time;
//   * every phase that chases dependent LDS reads (speculative parse, doubling, pointer jumping)
//     is written as level-by-level loops over several independent items per lane, so the reads
//     of one level are in flight together instead of one ~64-cycle latency after another.
// The group output length table of the old kernel is gone (one lane per true group recomputes it),
// which pays for the ring in the 160 KB LDS budget.
#pragma once

#include <type_traits>

#include "../tsq_common.cuh"
#include "../tsq_dec_common.cuh"

namespace tsq {

// RING_ON: the standard layout (64 KiB history ring, 6 KiB chunks, ~150 KB of LDS: one block per CU).
// !RING_ON: the lean layout for more blocks than CUs -- no ring (history bytes are gathered from the block's own output in
// L2), 5 KiB chunks, ~71 KB of LDS: two blocks share a CU.
template <bool RING_ON>
struct RingCfgT {
    static constexpr uint32_t T = 1024;
    static constexpr uint32_t S = RING_ON ? 6144 : 5120;   // stream bytes per chunk
    static constexpr uint32_t SPAD = 160;
    static constexpr uint32_t OUTC = 2 * S;    // output bytes per chunk image
    static constexpr uint32_t D = 4;
    static constexpr uint32_t HOP = 1u << D;
    static constexpr uint32_t MAXG = RING_ON ? 544 : 448;  // >= S / 13 + 2 * HOP
    static constexpr uint32_t MAXSN = MAXG / HOP + 2;
    static constexpr uint32_t PER = S / T;     // offsets per lane
    static constexpr uint32_t OPER = OUTC / T; // image bytes per lane
    static constexpr uint16_t RES = 0xFFFF;
    static constexpr uint32_t RING = RING_ON ? 65536 : 0;
};
template <class Cfg>
struct RingLdsT {
    static constexpr uint32_t sbuf = 0;                                          // S + SPAD + 32
    static constexpr uint32_t nx1 = sbuf + Cfg::S + Cfg::SPAD + 32;              // u16[S]
    static constexpr uint32_t ja = nx1 + 2 * Cfg::S;
    static constexpr uint32_t jb = ja + 2 * Cfg::S;
    static constexpr uint32_t syms = nx1;                                        // DecSym[8 * MAXG] over nx1, ja, jb (dead after P4)
    static constexpr uint32_t srcp = jb + 2 * Cfg::S;                            // u16[OUTC + 16]
    static constexpr uint32_t obuf = srcp + 2 * (Cfg::OUTC + 16);                // u8[OUTC + 32]
    static constexpr uint32_t ring = obuf + Cfg::OUTC + 32;                      // u8[RING]
    static constexpr uint32_t gstart = ring + Cfg::RING;                         // u16[MAXG]
    static constexpr uint32_t glen = gstart + 2 * Cfg::MAXG;                     // u16[MAXG]
    static constexpr uint32_t gout = glen + 2 * Cfg::MAXG;                       // u32[MAXG]
    static constexpr uint32_t sn = gout + 4 * Cfg::MAXG;                         // u16[MAXSN + pad]
    static constexpr uint32_t wsum = sn + 2 * ((Cfg::MAXSN + 7) & ~7u);
    static constexpr uint32_t misc = wsum + 64;
    static constexpr uint32_t total = misc + 64;
    static_assert(8 * Cfg::MAXG * sizeof(DecSym) <= 6 * Cfg::S, "symbol records must fit the dead tables");
    static_assert(Cfg::S % Cfg::T == 0 && Cfg::OUTC % Cfg::T == 0, "per-lane item counts");
    static_assert(Cfg::MAXG >= Cfg::S / 13 + 2 * Cfg::HOP, "group table");
};
using RingCfg = RingCfgT<true>;
using RingLds = RingLdsT<RingCfg>;
using LeanCfg = RingCfgT<false>;
using LeanLds = RingLdsT<LeanCfg>;
static_assert(RingLds::total <= 160 * 1024, "LDS budget");
static_assert(2 * LeanLds::total <= 160 * 1024, "two lean blocks per CU");

template <bool RING_ON>
__global__ __launch_bounds__(1024, RING_ON ? 4 : 8) void dec_ring_kernel(const uint8_t* __restrict__ container,
                                                              const FrameInfo* __restrict__ frames,
                                                              uint8_t* __restrict__ outbuf,
                                                              int32_t* __restrict__ status)
{
    using C = RingCfgT<RING_ON>;
    using RingLds = RingLdsT<C>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* const s_raw = lds + RingLds::sbuf;
    uint16_t* const nx1 = reinterpret_cast<uint16_t*>(lds + RingLds::nx1);
    uint16_t* const ja = reinterpret_cast<uint16_t*>(lds + RingLds::ja);
    uint16_t* const jb = reinterpret_cast<uint16_t*>(lds + RingLds::jb);
    DecSym* const syms = reinterpret_cast<DecSym*>(lds + RingLds::syms);
    uint16_t* const srcp = reinterpret_cast<uint16_t*>(lds + RingLds::srcp);
    uint8_t* const o_raw = lds + RingLds::obuf;
    uint8_t* const ring = lds + RingLds::ring;
    uint16_t* const gstart = reinterpret_cast<uint16_t*>(lds + RingLds::gstart);
    uint16_t* const glen = reinterpret_cast<uint16_t*>(lds + RingLds::glen);
    uint32_t* const gout = reinterpret_cast<uint32_t*>(lds + RingLds::gout);
    uint16_t* const sn = reinterpret_cast<uint16_t*>(lds + RingLds::sn);
    uint32_t* const wsum = reinterpret_cast<uint32_t*>(lds + RingLds::wsum);
    uint32_t* const misc = reinterpret_cast<uint32_t*>(lds + RingLds::misc);
    // misc[0] n super nodes, [1] groups in chunk, [2] first group over the image budget,
    // [3] group that completes the block, [4] error, [5] exit offset of the chain

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (*status != 0) return;
    const FrameInfo f = frames[blockIdx.x];
    const uint8_t* const in = container + f.stream_at;
    uint8_t* const out = outbuf + f.out_at;
    const uint32_t in_len = f.stream_len, size = f.out_len, ext = f.ext;
    const bool out_aligned = ((uintptr_t)out & 15u) == 0u;

#ifdef TSQ_STATS
    unsigned long long st_[16] = {0};
#endif
    TSQD_T0();
    uint32_t sp = 3, op = 0;
    uint32_t stamp = 0;            // pointer-jumping rounds so far (never repeats, so the flag word needs no reset)
    if (tid == 0) { misc[4] = 0; misc[6] = 0; }
    __syncthreads();

    while (op < size) {
        // ---------------- P0: stage the chunk.  sbuf[k] = in[sp + k]; zeros beyond the stream.
        const uint32_t avail = in_len - sp;
        const uint32_t slim = avail < C::S ? avail : C::S;
        const uint32_t skew = (uint32_t)((uintptr_t)(in + sp) & 15u);
        uint8_t* const sbuf = s_raw + skew;
        {
            const uint8_t* gbase = in + sp - skew;
            const uint32_t want = skew + (avail < C::S + C::SPAD ? avail : C::S + C::SPAD);
            for (uint32_t w = tid; w < (C::S + C::SPAD + 32) / 16; w += C::T) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if ((w << 4) < want) v = *reinterpret_cast<const uint4*>(gbase + (w << 4));
                *reinterpret_cast<uint4*>(s_raw + (w << 4)) = v;
            }
        }
        if (tid == 0) { misc[0] = 0; misc[1] = 0; misc[2] = 0xFFFFFFFFu; misc[3] = 0xFFFFFFFFu; misc[5] = 0; }
        // every image byte starts as "final"; P6a overwrites the entries of bytes that point into this chunk
        for (uint32_t w = tid; w < (2 * (C::OUTC + 16)) / 16; w += C::T)
            *reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(srcp) + (w << 4)) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        __syncthreads();
        for (uint32_t k = tid; k < C::SPAD + 16; k += C::T) { uint32_t o = slim + k; if (o >= avail && o < C::S + C::SPAD + 16) sbuf[o] = 0; }
        __syncthreads();
        TSQD_ACC(0); TSQD_CNT(12, 1);

        // ---------------- P1: speculative group parse at every offset, PER offsets per lane level by level
        {
            uint32_t c[C::PER], p[C::PER];
#pragma unroll
            for (uint32_t k = 0; k < C::PER; ++k) { const uint32_t o = tid + k * C::T; c[k] = sbuf[o]; p[k] = o + 1u; }
#pragma unroll
            for (uint32_t pr = 0; pr < 4; ++pr) {
                uint32_t sb[C::PER];
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) sb[k] = sbuf[p[k] < C::S + C::SPAD ? p[k] : 0u];
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) {
                    const uint32_t hi = sb[k] >> 4, lo = sb[k] & 15u;
                    const uint32_t lit_hi = (c[k] >> (7u - 2u * pr)) & 1u, lit_lo = (c[k] >> (6u - 2u * pr)) & 1u;
                    p[k] += 1u + (lit_hi ? hi + 1u : 2u) + (lit_lo ? lo + 1u : 2u);
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < C::PER; ++k) {
                const uint32_t o = tid + k * C::T;
                nx1[o] = (uint16_t)(o < slim ? p[k] : C::S + C::SPAD);
            }
        }
        __syncthreads();
        TSQD_ACC(1);

        // ---------------- P2: J = next^(2^D), both reads of a pass issued for all PER offsets before use
        {
            const uint16_t* src = nx1;
            uint16_t* dst = ja;
#pragma unroll
            for (uint32_t d = 0; d < C::D; ++d) {
                uint32_t x[C::PER], y[C::PER];
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) x[k] = src[tid + k * C::T];
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) y[k] = src[x[k] < slim ? x[k] : 0u];
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) dst[tid + k * C::T] = (uint16_t)(x[k] < slim ? y[k] : x[k]);
                __syncthreads();
                src = dst;
                dst = (dst == ja) ? jb : ja;
            }
        }
        // passes write ja, jb, ja, jb: with D = 4 the result is in jb
        const uint16_t* const J = (C::D & 1u) ? ja : jb;
        TSQD_ACC(2);

        // ---------------- P3: one lane follows the chain of super nodes
        if (tid == 0) {
            uint32_t x = 0, k = 0;
            while (x < slim && k < C::MAXSN) { sn[k++] = (uint16_t)x; x = J[x]; }
            misc[0] = k;
            if (k >= C::MAXSN && x < slim) misc[4] = kErrStream;
        }
        __syncthreads();
        const uint32_t nsn = misc[0];
        TSQD_ACC(3);

        // ---------------- P4: expand super nodes into group starts; one lane per group computes its output length
        if (tid < nsn) {
            uint32_t x = sn[tid], cnt = 0;
            for (uint32_t t = 0; t < C::HOP && x < slim; ++t) { gstart[tid * C::HOP + t] = (uint16_t)x; x = nx1[x]; cnt++; }
            if (tid == nsn - 1) { misc[1] = (nsn - 1) * C::HOP + cnt; misc[5] = x; }
        }
        __syncthreads();
        uint32_t ng = misc[1];
        {
            uint32_t v = 0;
            if (tid < ng) {
                const uint32_t x = gstart[tid];
                const uint32_t c = sbuf[x];
                uint32_t p = x + 1u;
#pragma unroll
                for (uint32_t pr = 0; pr < 4; ++pr) {
                    const uint32_t sb = sbuf[p];
                    const uint32_t hi = sb >> 4, lo = sb & 15u;
                    const uint32_t lit_hi = (c >> (7u - 2u * pr)) & 1u, lit_lo = (c >> (6u - 2u * pr)) & 1u;
                    v += ((!lit_hi && ext && hi < 3u) ? (hi + 2u) << 4 : hi + 1u) + ((!lit_lo && ext && lo < 3u) ? (lo + 2u) << 4 : lo + 1u);
                    p += 1u + (lit_hi ? hi + 1u : 2u) + (lit_lo ? lo + 1u : 2u);
                }
                glen[tid] = (uint16_t)v;
            }
            uint32_t incl = v;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t up = __shfl_up(incl, d); if (lane >= d) incl += up; }
            if (lane == 63) wsum[wid] = incl;
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t w = 0; w < wid; ++w) before += wsum[w];
            const uint32_t excl = before + incl - v;
            if (tid < ng) {
                gout[tid] = op + excl;
                if (excl + 512u > C::OUTC) atomicMin(&misc[2], tid);
                if (op + excl + v >= size) atomicMin(&misc[3], tid);
            }
        }
        __syncthreads();
        uint32_t next_sp, next_op;
        bool last_chunk = false;
        {
            const uint32_t cut = misc[2], fin = misc[3];
            if (fin != 0xFFFFFFFFu && fin < cut) { ng = fin + 1; last_chunk = true; next_sp = sp; next_op = size; }
            else if (cut != 0xFFFFFFFFu) { ng = cut; next_sp = sp + gstart[cut]; next_op = gout[cut]; }
            else { next_sp = sp + misc[5]; next_op = ng ? gout[ng - 1] + glen[ng - 1] : op; }
        }
        if (ng == 0 || (!last_chunk && next_sp >= in_len)) {
            if (tid == 0) atomicMax(status, kErrStream);
            return;
        }
        const uint32_t image_len = next_op - op;
        const uint32_t oskew = (uint32_t)((uintptr_t)(out + op) & 15u);
        uint8_t* const obuf = o_raw + oskew;
        __syncthreads();
        TSQD_ACC(4);

        // ---------------- P5: symbol records, one lane per PAIR (four lanes per group).  A pair's stream position and output
        // position follow from the pairs before it, so its lane first walks those (sizes only); that is redundant work, but
        // the phase is a chain of dependent LDS reads, and a quarter of the lanes doing four pairs each took twice as long.
        for (uint32_t gi = tid; gi < ng * 4u; gi += C::T) {
            const uint32_t g = gi >> 2, mine = gi & 3u;
            const uint32_t x = gstart[g];
            const uint32_t c = sbuf[x];
            uint32_t p = x + 1, j = gout[g];
            for (uint32_t pr = 0; pr < mine; ++pr) {                      // the pairs in front: only how far they move p and j
                if (j < size) {
                    const uint32_t sb = sbuf[p];
                    p++;
#pragma unroll
                    for (uint32_t sidx = 0; sidx < 2; ++sidx) {
                        if (j < size) {
                            const uint32_t nib = sidx == 0 ? sb >> 4 : sb & 15u;
                            const uint32_t lit = (c >> (7u - (2u * pr + sidx))) & 1u;
                            const uint32_t room = size - j;
                            const uint32_t len = lit ? nib + 1u : ((ext && nib < 3u) ? (nib + 2u) << 4 : nib + 1u);
                            p += lit ? len : 2u;
                            j += len < room ? len : room;
                        }
                    }
                }
            }
            uint32_t bad = 0;
            DecSym* rec = syms + g * 8u + mine * 2u;
            uint32_t sb = 0;
            const uint32_t origin = j;
            if (j < size) { if (p >= avail) bad = 1; sb = sbuf[p]; p++; }
#pragma unroll
            for (uint32_t sidx = 0; sidx < 2; ++sidx) {
                DecSym r; r.out_rel = 0; r.len = 0; r.kind = 0; r.a = 0;
                if (j < size && !bad) {
                    const uint32_t nib = sidx == 0 ? sb >> 4 : sb & 15u;
                    const uint32_t lit = (c >> (7u - (2u * mine + sidx))) & 1u;
                    const uint32_t room = size - j;
                    if (lit) {
                        const uint32_t len = nib + 1u, take = len < room ? len : room;
                        if (p + take > avail) bad = 1;
                        r.out_rel = (uint16_t)(j - op); r.len = (uint8_t)take; r.kind = 1; r.a = p;
                        p += len; j += take;
                    } else {
                        if (p + 2u > avail) bad = 1;
                        const uint32_t off = (uint32_t)sbuf[p] | ((uint32_t)sbuf[p + 1] << 8);
                        p += 2;
                        const uint32_t len = (ext && nib < 3u) ? (nib + 2u) << 4 : nib + 1u;
                        const uint32_t take = len < room ? len : room;
                        if (off > origin || take > off) bad = 1;
                        r.out_rel = (uint16_t)(j - op); r.len = (uint8_t)take; r.kind = 2; r.a = origin - off;
                        j += take;
                    }
                    if (bad) r.kind = 0;
                }
                rec[sidx] = r;
            }
            if (bad) misc[4] = kErrStream;
        }
        __syncthreads();
        if (misc[4] != 0) { if (tid == 0) atomicMax(status, (int32_t)misc[4]); return; }
        TSQD_ACC(5);

        // ---------------- P6a: scatter literal bytes and history bytes (from the ring); record in-chunk sources
        for (uint32_t s = tid; s < ng * 8u; s += C::T) {
            const DecSym r = syms[s];
            if (r.kind == 1) {
                for (uint32_t t = 0; t < r.len; ++t) obuf[r.out_rel + t] = sbuf[r.a + t];
            } else if (r.kind == 2) {
                for (uint32_t t = 0; t < r.len; ++t) {
                    const uint32_t a = r.a + t, q = r.out_rel + t;
                    if (a < op) obuf[q] = RING_ON ? ring[a & (C::RING - 1u)] : out[a];
                    else srcp[q] = (uint16_t)(a - op);
                }
            }
        }
        __syncthreads();
        TSQD_ACC(6);

        // ---------------- P6b: pointer jumping over the COMPACTED set of pending bytes.  About a third of the image
        // bytes point into the chunk; they are gathered into a dense list (block-wide scan) so that a round costs
        // in proportion to them, not to the image.  A round follows two links (all reads before all writes).
        {
            uint16_t* const plist = reinterpret_cast<uint16_t*>(lds + RingLds::syms);      // symbol records are dead now
            uint32_t mine = 0;
            uint32_t ptr0[C::OPER];
#pragma unroll
            for (uint32_t r = 0; r < C::OPER; ++r) {
                const uint32_t q = tid + r * C::T;
                ptr0[r] = q < image_len ? srcp[q] : C::RES;
            }
#pragma unroll
            for (uint32_t r = 0; r < C::OPER; ++r) if (ptr0[r] != C::RES) mine++;
            uint32_t incl = mine;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t up = __shfl_up(incl, d); if (lane >= d) incl += up; }
            __syncthreads();                                       // every lane has read its symbol records (P6a)
            if (lane == 63) wsum[wid] = incl;
            __syncthreads();
            uint32_t at = incl - mine, n_pending = 0;
            for (uint32_t w = 0; w < C::T / 64; ++w) { const uint32_t c = wsum[w]; if (w < wid) at += c; n_pending += c; }
#pragma unroll
            for (uint32_t r = 0; r < C::OPER; ++r) if (ptr0[r] != C::RES) plist[at++] = (uint16_t)(tid + r * C::T);
            __syncthreads();
            TSQD_ACC(9);
            const uint32_t slots = (n_pending + C::T - 1u) / C::T;                          // block-uniform, usually 3-4 of 12
            // the rounds, compiled for a fixed number of list slots per lane: with the full 12 every round would pay
            // for 12 guarded (mostly skipped) bodies per loop
            auto run_rounds = [&](auto slots_c) {
                constexpr uint32_t N = decltype(slots_c)::value;
                uint32_t q1[N], p1[N];
                uint32_t pending = 0;
#pragma unroll
                for (uint32_t r = 0; r < N; ++r) {
                    q1[r] = 0; p1[r] = C::RES;
                    const uint32_t i = tid + r * C::T;
                    if (i < n_pending) { q1[r] = plist[i]; pending |= 1u << r; }
                }
#pragma unroll
                for (uint32_t r = 0; r < N; ++r) if ((pending >> r) & 1u) p1[r] = srcp[q1[r]];
                for (uint32_t round = 0; round < 24; ++round) {
                    ++stamp;
                    const bool wave_pending = __ballot(pending != 0u) != 0ull;
                    if (wave_pending && lane == 0) misc[6] = stamp;
                    __syncthreads();
                    if (misc[6] != stamp) break;
                    TSQD_CNT(13, 1);
                    uint32_t p2[N], p3[N], v1[N], v2[N];
#pragma unroll
                    for (uint32_t r = 0; r < N; ++r) { const uint32_t i1 = (pending >> r) & 1u ? p1[r] : 0u; p2[r] = srcp[i1]; v1[r] = obuf[i1]; }
#pragma unroll
                    for (uint32_t r = 0; r < N; ++r) { const uint32_t i2 = (((pending >> r) & 1u) && p2[r] != C::RES) ? p2[r] : 0u; p3[r] = srcp[i2]; v2[r] = obuf[i2]; }
                    __syncthreads();
#pragma unroll
                    for (uint32_t r = 0; r < N; ++r) {
                        if ((pending >> r) & 1u) {
                            const uint32_t q = q1[r];
                            if (p2[r] == C::RES) { obuf[q] = (uint8_t)v1[r]; srcp[q] = C::RES; pending &= ~(1u << r); }
                            else if (p3[r] == C::RES) { obuf[q] = (uint8_t)v2[r]; srcp[q] = C::RES; pending &= ~(1u << r); }
                            else { srcp[q] = (uint16_t)p3[r]; p1[r] = p3[r]; }
                        }
                    }
                }
            };
            if (slots <= 2u) run_rounds(std::integral_constant<uint32_t, 2>{});
            else if (slots <= 4u) run_rounds(std::integral_constant<uint32_t, 4>{});
            else if (slots <= 6u) run_rounds(std::integral_constant<uint32_t, 6>{});
            else run_rounds(std::integral_constant<uint32_t, C::OPER>{});
        }
        TSQD_ACC(7);

        // ---------------- P7: image -> HBM (aligned 16-byte stores) and -> history ring
        {
            const uint32_t head = (16u - oskew) & 15u;
            const uint32_t hb = head < image_len ? head : image_len;
            if (tid < hb) { out[op + tid] = obuf[tid]; if (RING_ON) ring[(op + tid) & (C::RING - 1u)] = obuf[tid]; }
            const uint32_t words = image_len > hb ? (image_len - hb) >> 4 : 0;
            for (uint32_t w = tid; w < words; w += C::T) {
                const uint4 v = *reinterpret_cast<const uint4*>(obuf + hb + (w << 4));
                *reinterpret_cast<uint4*>(out + op + hb + (w << 4)) = v;
                const uint32_t ri = (op + hb + (w << 4)) & (C::RING - 1u);
                if (!RING_ON) {}
                else if (out_aligned) *reinterpret_cast<uint4*>(ring + ri) = v;     // same 16-byte phase as the output address
                else { const uint32_t wd[4] = {v.x, v.y, v.z, v.w}; for (uint32_t t = 0; t < 16; ++t) ring[(ri + t) & (C::RING - 1u)] = (uint8_t)(wd[t >> 2] >> (8u * (t & 3u))); }
            }
            const uint32_t tail_at = hb + (words << 4);
            if (tid < image_len - tail_at) { out[op + tail_at + tid] = obuf[tail_at + tid]; if (RING_ON) ring[(op + tail_at + tid) & (C::RING - 1u)] = obuf[tail_at + tid]; }
        }
        __syncthreads();
        op = next_op;
        sp = next_sp;
        TSQD_ACC(8); TSQD_CNT(14, ng);
        if (last_chunk) break;
    }
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && tid == 0) for (int q = 0; q < 16; ++q) g_dec_stats[q] = st_[q];
#endif
}

}  // namespace tsq
