// tsq_dec_sym.cuh -- block decoder: speculative parse, one lane per symbol pair, byte lanes with asynchronous pointer jumping (kernel variant 0).
//
// The reference walks the stream with one dependent load per symbol (tsq_decode.cpp:62-88) and copies every symbol with a
// 16-byte load/store.  Here each 6 KiB chunk of stream is staged in LDS and handled in data-parallel phases by one
// workgroup of 16 wavefronts:
//
//   P0  stage the chunk (prefetched into registers while the previous chunk was being copied).
//   P1  every byte offset is parsed AS IF a group (control byte + 4 pairs) started there: first every byte is taken as a size byte
//       (the stream length of the pair it would head under each of the four control-bit pairs, packed in one word: arithmetic
//       only), then four chained look-ups of those words, one per pair.
//   P2  pointer doubling: next^2, next^4, next^8, next^16 (all kept).
//   P3  one lane follows next^16 from the known chunk start (one dependent LDS hop per 16 groups), with the LDS to itself.
//   P4  one lane per group: its start from the hop it hangs on and the bits of its index (next^8, ^4, ^2, ^1: four
//       dependent reads, all groups in parallel), then its four pairs' stream positions and output offsets; a block-wide
//       scan gives the groups' output positions.
//   P5  symbols -> bytes, with aligned, conflict-free LDS traffic only (an unaligned ds access costs 3.5x an aligned one on
//       gfx950, tools/micro/lds_unaligned.hip):
//       (a) one lane per PAIR decodes and validates its two symbols and drops one record word per symbol (kind, source delta)
//           at the symbol's first byte index;
//       (b) one lane per 12 output bytes (three aligned words of the ring): a "last record so far" scan hands every byte its
//           symbol's record;
//       (c) every byte fetches its value -- literal bytes from the stream buffer, history bytes from the 64 KiB of previous
//           output kept in the ring -- or, when its source byte lies in this same chunk, a pointer to it;
//       (d) asynchronous pointer jumping without barriers resolves those pointers in O(log chain depth) steps (every occurrence
//           of a frequent word copies the one before it, so chains are as long as the chunk has occurrences);
//       the chunk's bytes are composed IN the ring (77 KiB of LDS: 64 KiB of history + the chunk being built).
//   P7  the chunk's bytes go from the ring to HBM with aligned 16-byte stores, by the half of the workgroup that idles in the next
//       chunk's P4.
//
// Offsets and stream bounds are validated (the reference validates nothing); status codes as the oracle's decoder.
#pragma once

#include <type_traits>

#include "tsq_common.cuh"
#include "tsq_dec_common.cuh"

namespace tsq {

struct SymCfg {
    static constexpr uint32_t T = 1024;
    static constexpr uint32_t S = 6144;                    // stream bytes per chunk
    static constexpr uint32_t SPAD = 160;                  // a group is at most 133 bytes
    static constexpr uint32_t OUTC = 2 * S;                // output bytes per chunk at most
    static constexpr uint32_t HOP = 16;
    static constexpr uint32_t MAXG = 512;                  // >= S / 13 + 2 * HOP, a multiple of HOP
    static constexpr uint32_t MAXSN = MAXG / HOP + 2;
    static constexpr uint32_t PER = S / T;                 // stream offsets per lane in P1 / P2
    static constexpr uint32_t TERM = S + SPAD;             // "no group here": beyond every real offset
    static constexpr uint32_t R = 65536 + OUTC + 64;       // ring: 64 KiB of history + the chunk being built (a multiple of 16)
    static constexpr uint32_t RPAD = 64;                   // slack behind the ring (the bytes past an image's end inherit its last record)
    static constexpr uint32_t SWORDS = (S + SPAD + 16) / 16;   // 16-byte words of stream staged per chunk
};
struct SymLds {
    static constexpr uint32_t sbuf = 0;                                             // u8[S + SPAD + 16]
    static constexpr uint32_t j1 = sbuf + 16 * SymCfg::SWORDS;                      // u8[S]: next - offset
    // u16[JN] each: next^2 .. next^16 of every offset; the entries S .. TERM hold TERM ("the chain has left the chunk": a fixed
    // point), so that a look-up needs no range test
    static constexpr uint32_t JN = (SymCfg::S + SymCfg::SPAD + 8u) & ~7u;
    static constexpr uint32_t j2 = j1 + SymCfg::S;
    static constexpr uint32_t j4 = j2 + 2 * JN;
    static constexpr uint32_t j8 = j4 + 2 * JN;
    static constexpr uint32_t j16 = j8 + 2 * JN;
    static constexpr uint32_t recw = j1;                                            // P5: u32[OUTC + 16] symbol records by first byte index, over j1 .. j16
    static constexpr uint32_t ent = j1;                                             // P5: u16[OUTC + 16] byte entries (once the records are read)
    static constexpr uint32_t plist = ent + 2 * (SymCfg::OUTC + 16);                // P5: u16[OUTC] per wavefront, the bytes that wait for a source
    static constexpr uint32_t gstart = j16 + 2 * JN;                                // u16[MAXG]
    static constexpr uint32_t glen = gstart + 2 * SymCfg::MAXG;                     // u16[MAXG]
    static constexpr uint32_t gout = glen + 2 * SymCfg::MAXG;                       // u32[MAXG]
    static constexpr uint32_t pairs = gout + 4 * SymCfg::MAXG;                      // u32[4 * MAXG]: stream pos | out offset << 13 | 2 control bits << 30
    static constexpr uint32_t sn = pairs + 16 * SymCfg::MAXG;                       // u16[MAXSN + pad]
    static constexpr uint32_t wsum = (sn + 2 * ((SymCfg::MAXSN + 7) & ~7u) + 15) & ~15u;   // u32[16]
    static constexpr uint32_t misc = wsum + 64;                                     // u32[16]
    static constexpr uint32_t lut = misc + 64;                                      // u16[1024]: (two control bits, size byte) -> stream bytes of the pair
    static constexpr uint32_t ring = (lut + 2048 + 15) & ~15u;                      // u8[R + RPAD]
    static constexpr uint32_t total = ring + SymCfg::R + SymCfg::RPAD;               // (the product asks for nothing it does not use)
    static_assert(SymCfg::R % 16 == 0, "ring phase");
    static_assert(recw % 16 == 0 && recw + 4 * (SymCfg::OUTC + 16) <= gstart && plist % 16 == 0 && plist + 2 * SymCfg::OUTC <= gstart,
                  "records, byte entries and waiting lists fit the dead doubling tables");
    static_assert(SymCfg::OUTC == 12 * SymCfg::T, "twelve bytes per lane");
    static_assert(SymCfg::S % SymCfg::T == 0 && SymCfg::SWORDS <= SymCfg::T, "lane counts");
    static_assert(SymCfg::MAXG <= SymCfg::T / 2 && SymCfg::MAXG % SymCfg::HOP == 0 && SymCfg::MAXG >= SymCfg::S / 13 + 2 * SymCfg::HOP, "group table");
};
static_assert(SymLds::total <= 160 * 1024, "LDS budget");

// length of a symbol from its nibble (tsq_decode.cpp:66-88,174-224)
__device__ __forceinline__ uint32_t sym_out_len(uint32_t nib, uint32_t lit, uint32_t ext)
{
    return lit ? nib + 1u : ((ext && nib < 3u) ? (nib + 2u) << 4 : nib + 1u);
}

// stream bytes and output bytes of the pair whose size byte is `sb` and whose control bits are `cc` (bit 1: first symbol is a
// literal, bit 0: second) (tsq_decode.cpp:66-88,174-224)
__device__ __forceinline__ void pair_lens(uint32_t sb, uint32_t cc, uint32_t ext, uint32_t& slen, uint32_t& olen)
{
    const uint32_t hi = sb >> 4, lo = sb & 15u;
    const uint32_t lit_hi = cc & 2u, lit_lo = cc & 1u;
    const uint32_t o_hi = (!lit_hi && ext && hi < 3u) ? (hi + 2u) << 4 : hi + 1u;
    const uint32_t o_lo = (!lit_lo && ext && lo < 3u) ? (lo + 2u) << 4 : lo + 1u;
    slen = 1u + (lit_hi ? hi + 1u : 2u) + (lit_lo ? lo + 1u : 2u);
    olen = o_hi + o_lo;
}

__global__ __launch_bounds__(1024) void dec_sym_kernel(const uint8_t* __restrict__ container, const FrameInfo* __restrict__ frames,
                                                       uint8_t* __restrict__ outbuf, int32_t* __restrict__ status)
{
    using C = SymCfg;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* const s_raw = lds + SymLds::sbuf;
    uint8_t* const j1 = lds + SymLds::j1;
    uint16_t* const j2 = reinterpret_cast<uint16_t*>(lds + SymLds::j2);
    uint16_t* const j4 = reinterpret_cast<uint16_t*>(lds + SymLds::j4);
    uint16_t* const j8 = reinterpret_cast<uint16_t*>(lds + SymLds::j8);
    uint16_t* const j16 = reinterpret_cast<uint16_t*>(lds + SymLds::j16);
    uint16_t* const gstart = reinterpret_cast<uint16_t*>(lds + SymLds::gstart);
    uint16_t* const glen = reinterpret_cast<uint16_t*>(lds + SymLds::glen);
    uint32_t* const gout = reinterpret_cast<uint32_t*>(lds + SymLds::gout);
    uint32_t* const pairs = reinterpret_cast<uint32_t*>(lds + SymLds::pairs);
    uint16_t* const sn = reinterpret_cast<uint16_t*>(lds + SymLds::sn);
    uint32_t* const wsum = reinterpret_cast<uint32_t*>(lds + SymLds::wsum);
    uint32_t* const misc = reinterpret_cast<uint32_t*>(lds + SymLds::misc);
    uint8_t* const ring = lds + SymLds::ring;
    uint32_t* const recw = reinterpret_cast<uint32_t*>(lds + SymLds::recw);
    // misc[0] super nodes, [1] groups in chunk, [2] first group over the image budget, [3] group that completes the block,
    // [4] error, [5] exit offset of the chain, [9], [10] instrumented builds only

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    // Another block may have reported an error: leave, but all together (thread 0 reads the word, the workgroup branches on its copy
    // in LDS -- a per-thread read could split the workgroup while other blocks are still writing the word).
    if (tid == 0) misc[11] = (uint32_t)*status;
    __syncthreads();
    if (misc[11] != 0u) return;
    const FrameInfo f = frames[blockIdx.x];
    // The descriptor may come straight from an untrusted container (tsqa_decode_blocks_async): a stream shorter than its 3-byte
    // header, longer than a block slot, or an output longer than a block is refused before anything is read through it.
    if (f.stream_len < 3u || f.stream_len > kSlotSize || f.out_len > kBlockSize) {
        if (tid == 0) atomicMax(status, kErrStream);
        return;
    }
    const uint8_t* const in = container + f.stream_at;
    uint8_t* const out = outbuf + f.out_at;
    const uint32_t in_len = f.stream_len, size = f.out_len, ext = f.ext;
    // ring address of output position p: (p + oskew) mod R, so that 16-byte words of the ring are 16-byte words of HBM
    const uint32_t oskew = (uint32_t)((uintptr_t)out & 15u);

#ifdef TSQ_STATS
    unsigned long long st_[16] = {0};
    unsigned long long wj_[3] = {0, 0, 0};
#endif
    TSQD_T0();
    if (tid == 0) { misc[4] = 0; misc[9] = 0; misc[10] = 0; }
    uint32_t sp = 3, op = 0;
    uint32_t ring_op = oskew;            // ring address of position op
    // what P7 still has to write out: the previous chunk's image
    uint32_t prev_op = 0, prev_len = 0, prev_ring = 0;

    auto flush_image = [&](uint32_t first_tid, uint32_t n_threads) {
        // image bytes [prev_op, prev_op + prev_len) from the ring to HBM: head bytes up to the first aligned word, aligned
        // 16-byte words, tail bytes.  Threads first_tid .. first_tid + n_threads - 1 take part.
        if (prev_len == 0 || tid < first_tid) return;
        const uint32_t t = tid - first_tid;
        const uint32_t head = (16u - (prev_ring & 15u)) & 15u;
        const uint32_t hb = head < prev_len ? head : prev_len;
        if (t < hb) out[prev_op + t] = ring[prev_ring + t];                       // (the ring end is a multiple of 16: no wrap inside the head)
        const uint32_t words = (prev_len - hb) >> 4;
        uint32_t ra = prev_ring + hb; ra -= ra >= C::R ? C::R : 0u;
        for (uint32_t w = t; w < words; w += n_threads) {
            uint32_t a = ra + (w << 4); a -= a >= C::R ? C::R : 0u;
            *reinterpret_cast<uint4*>(out + prev_op + hb + (w << 4)) = *reinterpret_cast<const uint4*>(ring + a);
        }
        const uint32_t tail_at = hb + (words << 4);
        if (t < prev_len - tail_at) { uint32_t a = ra + (words << 4) + t; a -= a >= C::R ? C::R : 0u; out[prev_op + tail_at + t] = ring[a]; }
    };

    // P0 of the first chunk
    uint4 pre = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](uint32_t at) {
        // 16-byte word `tid` of the chunk that starts at stream offset `at`.  (Unaligned 16-byte global loads: the stream buffer in
        // LDS then starts exactly at the chunk, and every LDS access to it is naturally aligned.)  Only whole words are loaded here,
        // with no control flow behind the load, so that nothing waits for it before P0 of the next chunk; the last, partial word
        // of a stream is fetched byte by byte in that P0.
        const uint32_t avail = in_len - at;
        const uint32_t lim = avail < C::S + C::SPAD ? avail : C::S + C::SPAD;
        pre = make_uint4(0, 0, 0, 0);
        if (tid < C::SWORDS && (tid << 4) + 16u <= lim) __builtin_memcpy(&pre, in + at + (tid << 4), 16);
    };
    prefetch(sp);
    // the words prefetched into `pre` (chunk at stream offset `at`, `av` stream bytes from there) go to the stream buffer
    auto stage_words = [&](uint32_t at, uint32_t av) {
        if (tid < C::SWORDS) {
            uint4 w = pre;
            const uint32_t lim = av < C::S + C::SPAD ? av : C::S + C::SPAD, o = tid << 4;
            if (o < lim && o + 16u > lim) {                                       // the stream's last, partial word (once per block)
                uint32_t b[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; o + k < lim; ++k) b[k >> 2] |= (uint32_t)in[at + o + k] << (8u * (k & 3u));
                w = make_uint4(b[0], b[1], b[2], b[3]);
            }
            *reinterpret_cast<uint4*>(s_raw + (tid << 4)) = w;
        }
    };
    bool staged_ahead = false;
    __syncthreads();

    while (op < size) {
        const uint32_t avail = in_len - sp;
        const uint32_t slim = avail < C::S ? avail : C::S;
        uint8_t* const sbuf = s_raw;
        // ---------------- P0: the chunk (loaded a chunk ago) goes to LDS.  sbuf[k] = in[sp + k]; zeros beyond the stream.
        // (Only the block's first chunk is staged here: every later one goes to LDS behind the byte fetch of the chunk before it, beside
        //  that chunk's pointer jumping -- the stream buffer is dead from there on, and the words have long arrived: `stage_next` below.)
        if (!staged_ahead) stage_words(sp, avail);
        if (tid == 0) { misc[0] = 0; misc[1] = 0; misc[2] = 0xFFFFFFFFu; misc[3] = 0xFFFFFFFFu; misc[5] = 0; }
        // (pipelined: the chunk's stream is in sbuf and j1 holds its parse at every offset -- both made during the previous chunk's copy
        //  phases; this barrier also ends the previous chunk's ring write, which reads the byte entries the doubling tables now overwrite)
        __syncthreads();
        TSQD_ACC(0); TSQD_CNT(12, 1);

        // ---------------- P1 + P2: speculative group parse at every offset, then next^2 .. next^16.  Lane t owns offsets t, t + T, ...:
        // the lanes of a wavefront touch consecutive bytes, so neither their own entries nor the entries they point to (about one
        // group further on, again consecutive) collide in the LDS banks.  A lane keeps its own entries in registers from pass to
        // pass.  An offset at or beyond slim is terminal (TERM).
        {
            uint32_t x[C::PER], y[C::PER];
            uint32_t c[C::PER];
            // (A) every byte of the chunk taken as a size byte: the stream length of the pair it would head, for each of the four
            //     control-bit pairs, packed in one word: 5 | 4 + lo << 8 | 4 + hi << 16 | 3 + hi + lo << 24 (tsq_decode.cpp:66-88: a
            //     literal takes nibble + 1 bytes, a match two).  One lane per aligned word of the chunk, arithmetic only.  The table
            //     lies over the doubling tables (dead until P2).
            {
                uint32_t* const pl = reinterpret_cast<uint32_t*>(lds + SymLds::j4);
                constexpr uint32_t NW = (C::S + C::SPAD) / 4u;
#pragma unroll
                for (uint32_t k = 0; k < (NW + C::T - 1u) / C::T; ++k) {
                    const uint32_t w = tid + k * C::T;
                    if (w < NW) {
                        const uint32_t v = reinterpret_cast<const uint32_t*>(sbuf)[w];
                        uint32_t q[4];
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t hi = (v >> (8u * b + 4u)) & 15u, lo = (v >> (8u * b)) & 15u;
                            const uint32_t a = lo | (lo << 16), h2 = hi | (hi << 8);
                            q[b] = (a << 8) + 0x03040405u + (h2 << 16);
                        }
                        *reinterpret_cast<uint4*>(pl + 4u * w) = make_uint4(q[0], q[1], q[2], q[3]);
                    }
                }
            }
            __syncthreads();
            // (B) the four pairs of the group that would start at each offset: one table word per pair
            {
                const uint32_t* const pl = reinterpret_cast<const uint32_t*>(lds + SymLds::j4);
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) { const uint32_t o = tid + k * C::T; c[k] = (uint32_t)sbuf[o] << 3; x[k] = o + 1u; }
#pragma unroll
                for (uint32_t pr = 0; pr < 4; ++pr) {
#pragma unroll
                    for (uint32_t k = 0; k < C::PER; ++k) y[k] = pl[x[k]];                     // x < S + 133: inside the padded buffer
#pragma unroll
                    for (uint32_t k = 0; k < C::PER; ++k) x[k] += __builtin_amdgcn_ubfe(y[k], (c[k] >> (6u - 2u * pr)) & 0x18u, 8u);
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < C::PER; ++k) { const uint32_t o = tid + k * C::T; j1[o] = (uint8_t)(x[k] - o); x[k] = o < slim ? x[k] : C::TERM; }
            __syncthreads();
            TSQD_ACC(1);
#pragma unroll
            for (uint32_t k = 0; k < C::PER; ++k) { const uint32_t a = x[k] < slim ? x[k] : 0u; y[k] = a + j1[a]; }
#pragma unroll
            for (uint32_t k = 0; k < C::PER; ++k) { x[k] = x[k] < slim ? y[k] : C::TERM; j2[tid + k * C::T] = (uint16_t)x[k]; }
            // (the tables' tails: every offset from S to TERM is terminal)
            if (tid <= C::SPAD) { j2[C::S + tid] = (uint16_t)C::TERM; j4[C::S + tid] = (uint16_t)C::TERM; j8[C::S + tid] = (uint16_t)C::TERM; j16[C::S + tid] = (uint16_t)C::TERM; }
            __syncthreads();
            const uint16_t* src = j2;
            uint16_t* const dsts[3] = {j4, j8, j16};
#pragma unroll
            for (uint32_t d = 0; d < 3; ++d) {
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) y[k] = src[x[k]];                       // x <= TERM, and src[TERM] == TERM
#pragma unroll
                for (uint32_t k = 0; k < C::PER; ++k) { x[k] = y[k]; dsts[d][tid + k * C::T] = (uint16_t)x[k]; }
                __syncthreads();
                src = dsts[d];
            }
        }
        TSQD_ACC(2);

        // ---------------- P3: one lane follows next^16 from the chunk start  ||  P7 of the previous chunk on the other waves
        if (wid == 0) {
            // (the whole first wavefront walks, every lane the same chain: no lane mask to set up and restore; two hops per loop test --
            //  the table's tail is a fixed point, so the second look-up is safe wherever the first one lands)
            uint32_t x = 0, k = 0;
            while (x < slim && k < C::MAXSN) {
                const uint32_t x1 = j16[x];
                const uint32_t x2 = j16[x1];                                        // x1 <= TERM, and j16[TERM] == TERM
                sn[k++] = (uint16_t)x;
                if (x1 < slim && k < C::MAXSN) { sn[k++] = (uint16_t)x1; x = x2; }
                else x = x1;
            }
            if (lane == 0) { misc[0] = k; if (k >= C::MAXSN && x < slim) misc[4] = kErrStream; }
        }
        __syncthreads();
        const uint32_t nsn = misc[0];
        TSQD_ACC(3);

        // ---------------- P4: one lane per group.  Group 16 k + r starts where r's bits lead from super node k.
        // (The upper half of the workgroup has no group to look after: it writes the PREVIOUS chunk's bytes to HBM meanwhile -- P7.)
        // (measured in round 5: a build without this flush runs 4.558 against 4.553 ms, the flush beside P3's chain instead 4.572 -- it hides completely)
        if (tid >= C::T / 2) flush_image(C::T / 2, C::T / 2);
        {
            uint32_t x = C::TERM;
            if (tid < nsn * C::HOP) {
                x = sn[tid >> 4];
                if (tid & 8u) x = j8[x];
                if (tid & 4u) x = j4[x];
                if (tid & 2u) x = j2[x];
                if (tid & 1u) x = x < slim ? (uint32_t)(x + j1[x]) : C::TERM;
            }
            uint32_t v = 0;
            if (x < slim) {
                gstart[tid] = (uint16_t)x;
                const uint32_t c = sbuf[x];
                uint32_t p = x + 1u, pw[4];
#pragma unroll
                for (uint32_t pr = 0; pr < 4; ++pr) {
                    const uint32_t cc = (c >> (6u - 2u * pr)) & 3u;
                    pw[pr] = p | (v << 13) | (cc << 30);                  // p < 2^13, v < 2^14 (the image budget)
                    uint32_t sl, ol;
                    pair_lens(sbuf[p], cc, ext, sl, ol);
                    p += sl;
                    v += ol;
                }
                *reinterpret_cast<uint4*>(pairs + tid * 4u) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
                glen[tid] = (uint16_t)v;
                if (p >= slim) { misc[1] = tid + 1u; misc[5] = p; }       // the chunk's last group: the chain leaves the chunk here
            }
            const uint32_t incl = wave_scan_add(v);
            if (lane == 63) wsum[wid] = incl;
            __syncthreads();
            // output bytes of the wavefronts before this one: lane w takes wavefront w's total, one more scan, one readlane
            const uint32_t totals = wave_scan_add(lane < C::T / 64u ? wsum[lane] : 0u);
            const uint32_t before = wid ? (uint32_t)__builtin_amdgcn_readlane((int)totals, (int)wid - 1) : 0u;
            const uint32_t excl = before + incl - v;
            if (x < slim) {
                gout[tid] = op + excl;
                if (excl + 512u + 16u > C::OUTC) atomicMin(&misc[2], tid);
                if (op + excl + v >= size) atomicMin(&misc[3], tid);
            }
            // the doubling tables are dead from here on (every lane is past its last look-up in them): the record words of P5, which lie
            // over them, are cleared now, under the barrier that is needed anyway
            for (uint32_t w = tid; w < (C::OUTC + 16) / 4; w += C::T) *reinterpret_cast<uint4*>(recw + 4u * w) = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        uint32_t ng = misc[1];
        uint32_t next_sp, next_op;
        bool last_chunk = false;
        {
            const uint32_t cut = misc[2], fin = misc[3];
            if (fin != 0xFFFFFFFFu && fin < cut) { ng = fin + 1; last_chunk = true; next_sp = sp; next_op = size; }
            else if (cut != 0xFFFFFFFFu) { ng = cut; next_sp = sp + gstart[cut]; next_op = gout[cut]; }
            else { next_sp = sp + misc[5]; next_op = ng ? gout[ng - 1] + glen[ng - 1] : op; }
        }
        if (misc[4] != 0 || ng == 0 || (!last_chunk && next_sp >= in_len)) {
            if (tid == 0) atomicMax(status, kErrStream);
            return;
        }
        const uint32_t image_len = next_op - op;
        // the next chunk's stream is on its way while this one is copied
        if (!last_chunk) prefetch(next_sp);
        TSQD_ACC(4);

        // ---------------- P5: symbols -> bytes.
        // Image index i = (position - op) + lead, where lead = bytes of the ring word that holds position op which belong to the
        // previous chunk: index 0 is a 4-byte aligned ring address, lane t owns indices [12 t, 12 t + 12) = three aligned ring words.
        const uint32_t lead = ring_op & 3u;
        const uint32_t a0 = ring_op - lead;                                            // ring address of index 0
        // (a) one lane per PAIR: decode its two symbols, validate them, and drop a record word at the first byte index of every run
        //     of bytes that come from one place: 0x40000000 | pointer flag << 31 | 24-bit signed D.
        //       flag 0: the byte at index i is found at LDS address i + D (stream buffer for literals, ring for history);
        //       flag 1: the byte at index i is a copy of the byte at index i + D of this same chunk (D < 0).
        //     A literal is one run; a match is up to three (history before the ring's end, history after it, bytes of this chunk).
        if (tid == 0 && lead) recw[0] = 0x40000000u | (SymLds::ring + a0);          // the bytes in front of position op in the first word: kept
        uint32_t bad = 0;
        static_assert(4 * C::MAXG <= 2 * C::T, "at most two pairs per lane");
#pragma unroll
        for (uint32_t rep = 0; rep < 2; ++rep) {
            const uint32_t gi = tid + rep * C::T;
            if (gi >= ng * 4u) break;
            const uint32_t g = gi >> 2;
            const uint32_t pw = pairs[gi];
            uint32_t p = pw & 0x1FFFu, j = gout[g] + ((pw >> 13) & 0x3FFFu);
            const uint32_t origin = j;
            uint32_t sb = 0;
            if (j < size) { if (p >= avail) bad = 1; sb = sbuf[p]; p++; }
#pragma unroll
            for (uint32_t sidx = 0; sidx < 2; ++sidx) {
                if (j < size && !bad) {
                    const uint32_t nib = sidx == 0 ? sb >> 4 : sb & 15u;
                    const uint32_t lit = (pw >> (31u - sidx)) & 1u;
                    const uint32_t room = size - j;
                    const uint32_t ij = j - op + lead;
                    if (lit) {
                        const uint32_t len = nib + 1u, take = len < room ? len : room;
                        if (p + take > avail) bad = 1;
                        else recw[ij] = 0x40000000u | ((p - ij) & 0xFFFFFFu);
                        p += len; j += take;
                    } else {
                        if (p + 2u > avail) bad = 1;
                        const uint32_t off = (uint32_t)sbuf[p] | ((uint32_t)sbuf[p + 1] << 8);
                        p += 2;
                        const uint32_t len = sym_out_len(nib, 0, ext);
                        const uint32_t take = len < room ? len : room;
                        if (off > origin || take > off) bad = 1;
                        if (!bad) {
                            const uint32_t a = origin - off;                          // source position
                            const uint32_t n_hist = a >= op ? 0u : (op - a < take ? op - a : take);
                            if (n_hist) {
                                uint32_t x0 = a0 + C::R - ((op - a) - lead);            // ring address of the first source byte (index a - op + lead < lead)
                                x0 -= x0 >= C::R ? C::R : 0u;
                                recw[ij] = 0x40000000u | ((SymLds::ring + x0 - ij) & 0xFFFFFFu);
                                if (x0 + n_hist > C::R) { const uint32_t n1 = C::R - x0; recw[ij + n1] = 0x40000000u | ((SymLds::ring - (ij + n1)) & 0xFFFFFFu); }
                            }
                            if (n_hist < take) recw[ij + n_hist] = 0xC0000000u | ((a - j) & 0xFFFFFFu);   // source index - own index < 0
                        }
                        j += take;
                    }
                }
            }
        }
        if (bad) misc[4] = kErrStream;
        __syncthreads();
        if (misc[4] != 0) { if (tid == 0) atomicMax(status, (int32_t)misc[4]); return; }
        TSQD_ACC(5);
        // (b) one lane per 12 bytes: every byte takes the record of the symbol it lies in (the last record at or before it)
        typedef __attribute__((address_space(3))) uint16_t lds_u16;
        lds_u16* const le = (lds_u16*)(lds + SymLds::ent);                             // byte entries: 0x8000 | value when final, else source index
        lds_u16* const wl = (lds_u16*)(lds + SymLds::plist) + 768u * wid;               // this wavefront's waiting list
        const uint32_t own = 12u * tid;
        uint32_t r[12];
        {
            const uint4 q0 = *reinterpret_cast<const uint4*>(recw + own), q1 = *reinterpret_cast<const uint4*>(recw + own + 4u),
                        q2 = *reinterpret_cast<const uint4*>(recw + own + 8u);
            r[0] = q0.x; r[1] = q0.y; r[2] = q0.z; r[3] = q0.w; r[4] = q1.x; r[5] = q1.y; r[6] = q1.z; r[7] = q1.w; r[8] = q2.x; r[9] = q2.y; r[10] = q2.z; r[11] = q2.w;
#pragma unroll
            for (uint32_t k = 1; k < 12; ++k) r[k] = r[k] ? r[k] : r[k - 1];
            // the last record of the lanes before this one: a scan with "the later non-zero word wins" over the lanes' last records
            // (six DPP steps on the record itself; round 5 scanned a lane number and fetched the record with two ds_bpermute: two
            // LDS round trips per chunk on every wavefront)
            const uint32_t upto = wave_scan_last(r[11]);                                                            // inclusive
            uint32_t carry = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)upto, 0x138, 0xF, 0xF, false);           // wave_shr:1 : the lanes strictly before
            if (lane == 63) wsum[wid] = upto;
            __syncthreads();                                                       // (also: every lane has taken its records out of `recw`)
            {   // the last record of the wavefronts before this one: lane w looks at wavefront w's, the highest one that has any wins
                const uint32_t ws = lane < 16u ? wsum[lane] : 0u;
                const uint64_t m = __ballot(ws != 0u && lane < wid);
                const uint32_t prev = m ? (uint32_t)__builtin_amdgcn_readlane((int)ws, 63 - __builtin_clzll(m)) : 0u;
                carry = carry ? carry : prev;
            }
#pragma unroll
            for (uint32_t k = 0; k < 12; ++k) r[k] = r[k] ? r[k] : carry;
        }
        // (c) every byte whose record names an LDS address is fetched at once (literal bytes, history bytes, the bytes kept in the
        //     first word); every byte whose source lies in this chunk points at it and goes onto the wavefront's waiting list.
        //     Entry per byte: 0x8000 | value when final, else the index of the source byte.
        const bool live = own < lead + image_len;
        uint32_t pend = 0;
        if (live) {
            uint32_t v[12], by[12];
#pragma unroll
            for (uint32_t k = 0; k < 12; ++k) {
                v[k] = own + k + (uint32_t)((int32_t)(r[k] << 8) >> 8);               // LDS address of the byte, or index of its source
                pend |= (r[k] >> 31) << k;
            }
#pragma unroll
            for (uint32_t k = 0; k < 12; ++k) by[k] = lds[(r[k] >> 31) ? 0u : v[k]];
#pragma unroll
            for (uint32_t k = 0; k < 12; ++k) v[k] = (r[k] >> 31) ? v[k] : (0x8000u | by[k]);
#pragma unroll
            for (uint32_t w = 0; w < 3; ++w)
                *reinterpret_cast<uint2*>(lds + SymLds::ent + 2u * own + 8u * w) = make_uint2(v[4 * w] | (v[4 * w + 1] << 16), v[4 * w + 2] | (v[4 * w + 3] << 16));
        }
        uint32_t n_wait;                                                               // wavefront-uniform
        {
            const uint32_t cnt = (uint32_t)__builtin_popcount(pend);
            const uint32_t incl = wave_scan_add(cnt);
            n_wait = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t at = incl - cnt;
#pragma unroll
            for (uint32_t k = 0; k < 12; ++k) {
                if ((pend >> k) & 1u) wl[at] = (uint16_t)(own + k);
                at += (pend >> k) & 1u;
            }
        }
        __syncthreads();
        TSQD_ACC(6);
        // the NEXT chunk's stream goes to LDS now (P0 of chunk k + 1): nothing reads the stream buffer any more (the literal bytes were
        // fetched above), and the words were requested behind P4, a dozen thousand cycles ago
        if (!last_chunk) { stage_words(next_sp, in_len - next_sp); staged_ahead = true; }
        // (d) asynchronous pointer jumping, no barriers, one lane per waiting byte: it reads its source's entry; a final entry
        //     carries the value, any other entry is a pointer further back (entries only ever move towards the chain's root, so a
        //     stale read is still a valid ancestor).  Chains of any depth (every occurrence of a frequent word copies the one
        //     before it; runs of a short period) shrink geometrically.  Relaxed LDS atomics: plain ds_read / ds_write that the
        //     compiler neither caches nor serialises.
        {
            // The list is padded to whole passes of 64 with a spare entry of this wavefront (final from the start: a lane that sits
            // on it re-writes what it read), and the loop is compiled for the number of passes so that it is straight-line code:
            // all reads of an iteration in flight together, no branches.
            // (Round 6 tried two ways of evening out the wavefronts' lists -- the last wavefronts hold 340 waiting bytes per chunk, the
            //  first 54, the ones beyond the image's end none: tools/phase_stats.py --: the image's rows of sixteen lanes dealt round the
            //  wavefronts DOUBLES the loop's iterations, 4.74 ms against 4.49; ONE list for the workgroup cut into sixteen equal stretches
            //  takes 900 cycles per chunk off this phase and puts 1 350 onto the one before it (a barrier and a prefix over the
            //  wavefronts' counts in front of the list's stores), 4.52 against 4.46.)
            const uint32_t spare = C::OUTC + wid;
            if (lane == 0) le[spare] = 0x8000u;
            const uint32_t passes = (n_wait + 63u) >> 6;
            const uint32_t padded = passes <= 2u ? 2u : passes <= 4u ? 4u : passes <= 6u ? 6u : passes <= 8u ? 8u : 12u;
            if (passes) for (uint32_t it = n_wait + lane; it < padded * 64u; it += 64u) wl[it] = (uint16_t)spare;
#ifdef TSQ_STATS
            uint32_t iters_ = 0;
            const unsigned long long wj0_ = __builtin_amdgcn_s_memtime();
#endif
            auto jump = [&](auto passes_c) {
                constexpr uint32_t P = decltype(passes_c)::value;
                uint32_t q[P], ptr[P];
#pragma unroll
                for (uint32_t ps = 0; ps < P; ++ps) q[ps] = wl[ps * 64u + lane];
#pragma unroll
                for (uint32_t ps = 0; ps < P; ++ps) ptr[ps] = __hip_atomic_load(&le[q[ps]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (;;) {
                    uint32_t e[P], open = 0;
#pragma unroll
                    for (uint32_t ps = 0; ps < P; ++ps) e[ps] = __hip_atomic_load(&le[(ptr[ps] & 0x8000u) ? q[ps] : ptr[ps]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                    for (uint32_t ps = 0; ps < P; ++ps) {
                        __hip_atomic_store(&le[q[ps]], (uint16_t)e[ps], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        ptr[ps] = e[ps];
                        open |= (e[ps] & 0x8000u) ^ 0x8000u;
                    }
#ifdef TSQ_STATS
                    iters_++;
#endif
                    if (__ballot(open != 0u) == 0ull) break;
                }
            };
            if (passes == 0u) {}
            else if (passes <= 2u) jump(std::integral_constant<uint32_t, 2>{});
            else if (passes <= 4u) jump(std::integral_constant<uint32_t, 4>{});
            else if (passes <= 6u) jump(std::integral_constant<uint32_t, 6>{});
            else if (passes <= 8u) jump(std::integral_constant<uint32_t, 8>{});
            else jump(std::integral_constant<uint32_t, 12>{});
#ifdef TSQ_STATS
            if (lane == 0) { atomicMax(&misc[9], iters_); atomicAdd(&misc[10], n_wait); }
            // per wavefront: when it left the pointer jumping (imbalance between the wavefronts shows as the barrier wait behind it)
            if (lane == 0 && blockIdx.x == 0) { wj_[0] += __builtin_amdgcn_s_memtime() - wj0_; wj_[1] += n_wait; wj_[2] += iters_; }
#endif
        }
        __syncthreads();
        if (live) {                                                                    // every entry is final now: three aligned ring words per lane
            const uint2 e0 = *reinterpret_cast<const uint2*>(lds + SymLds::ent + 2u * own), e1 = *reinterpret_cast<const uint2*>(lds + SymLds::ent + 2u * own + 8u),
                        e2 = *reinterpret_cast<const uint2*>(lds + SymLds::ent + 2u * own + 16u);
            const uint32_t ev[6] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y};
#pragma unroll
            for (uint32_t w = 0; w < 3; ++w) {
                uint32_t x = a0 + own + 4u * w; x -= x >= C::R ? C::R : 0u;
                const uint32_t lo = ev[2 * w], hi = ev[2 * w + 1];
                *reinterpret_cast<uint32_t*>(ring + x) = (lo & 0xFFu) | ((lo >> 8) & 0xFF00u) | ((hi & 0xFFu) << 16) | ((hi >> 16) << 24);
            }
        }
        // (no barrier here: nothing reads the ring, and nothing overwrites the entries, before the barrier at the top of the next chunk)
        TSQD_ACC(7);
#ifdef TSQ_STATS
        if (tid == 0) { st_[14] += misc[9]; st_[11] += misc[10]; misc[9] = 0; misc[10] = 0; }
#endif

        // ---------------- the image is complete: it goes to HBM during the next chunk's P3 (or right now, after the last chunk)
        prev_op = op; prev_len = image_len; prev_ring = ring_op;
        ring_op += image_len; ring_op -= ring_op >= C::R ? C::R : 0u;
        op = next_op;
        sp = next_sp;
        TSQD_ACC(8);
        if (last_chunk) break;
    }
    __syncthreads();
    flush_image(0, C::T);
#ifdef TSQ_STATS
    if (blockIdx.x == 0 && tid == 0) for (int q = 0; q < 16; ++q) g_dec_stats[q] = st_[q];
    if (blockIdx.x == 0 && lane == 0) for (int q = 0; q < 3; ++q) g_dec_wave[wid * 3 + q] = wj_[q];
#endif
}

}  // namespace tsq
