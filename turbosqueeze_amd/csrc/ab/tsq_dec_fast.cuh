// tsq_dec_fast.cuh -- wave-parallel block decoder for gfx950 (A/B variant 2: superseded, not in the product library).
//
// What tsq_decode.cpp:42-315 does with one dependent load per symbol (the position of every
// size byte depends on all lengths before it) is restated as data-parallel phases over one
// LDS-resident chunk of the block stream at a time.  One workgroup (16 wavefronts) per block.
//
//   P0  stage S stream bytes (+ look-ahead) into LDS with 16-byte global loads
//   P1  speculative group parse: EVERY byte offset o is treated as if a group (control byte +
//       4 pairs) started there -> next[o] (offset of the following group) and glen[o] (bytes the
//       group produces).  Only the true group starts will be used, but computing all of them
//       removes the serial dependence from the expensive part.
//   P2  pointer doubling on next[]: J[o] = next^(2^D)[o]   (D gather passes over LDS)
//   P3  one lane follows J from the known chunk start: S / (22 * 2^D) dependent hops instead of
//       one per symbol
//   P4  lanes expand each hop back into its 2^D group starts; block-wide exclusive scan of the
//       group output lengths gives every group its output position
//   P5  one lane per group: 8 symbol records (kind, length, stream offset or match source)
//   P6  one lane per symbol scatters literal bytes / history bytes into the LDS output image and
//       records, for match bytes whose source lies inside this chunk, a pointer to the source byte;
//       the pointers are resolved by pointer jumping (log rounds, robust against long copy
//       chains such as period-N data), not by replaying the copies in order
//   P7  the finished chunk image goes to HBM with aligned 16-byte stores
//
// Match sources are relative to the output position at the start of the symbol PAIR
// (tsq_decode.cpp:69,73,82) and must end before it; that is checked (the reference checks
// nothing), which also makes the pointer graph acyclic.  Output is clamped at the size header.
#pragma once

#include "../tsq_common.cuh"
#include "../tsq_dec_common.cuh"

namespace tsq {

struct DecCfg {
    static constexpr uint32_t T = 1024;        // threads per workgroup
    static constexpr uint32_t S = 8192;        // stream bytes per chunk
    static constexpr uint32_t SPAD = 160;      // look-ahead a speculative group parse may touch (<= 133)
    static constexpr uint32_t OUTC = 16384;    // output bytes per chunk image
    static constexpr uint32_t D = 4;           // doubling passes: one hop = 16 groups
    static constexpr uint32_t HOP = 1u << D;
    static constexpr uint32_t MAXG = 768;      // >= S / 13 + 2 * HOP  (13 = shortest group)
    static constexpr uint32_t MAXSN = MAXG / HOP + 2;
    static constexpr uint16_t RES = 0xFFFF;    // "byte already final" marker in the pointer table
};


// LDS layout (bytes).  The symbol records reuse the doubling tables, which are dead after P4.
struct DecLds {
    static constexpr uint32_t sbuf = 0;                                        // S + SPAD + 32
    static constexpr uint32_t nx1 = sbuf + DecCfg::S + DecCfg::SPAD + 32;      // u16[S]
    static constexpr uint32_t ja = nx1 + 2 * DecCfg::S;                        // u16[S]
    static constexpr uint32_t jb = ja + 2 * DecCfg::S;                         // u16[S]
    static constexpr uint32_t gl = jb + 2 * DecCfg::S;                         // u16[S]
    static constexpr uint32_t syms = ja;                                       // DecSym[8 * MAXG] over ja, jb, gl
    static constexpr uint32_t srcp = gl + 2 * DecCfg::S;                       // u16[OUTC + 16]
    static constexpr uint32_t obuf = srcp + 2 * (DecCfg::OUTC + 16);           // u8[OUTC + 32]
    static constexpr uint32_t gstart = obuf + DecCfg::OUTC + 32;               // u16[MAXG]
    static constexpr uint32_t glen = gstart + 2 * DecCfg::MAXG;                // u16[MAXG]
    static constexpr uint32_t gout = glen + 2 * DecCfg::MAXG;                  // u32[MAXG]
    static constexpr uint32_t sn = gout + 4 * DecCfg::MAXG;                    // u16[MAXSN]
    static constexpr uint32_t wsum = sn + 2 * ((DecCfg::MAXSN + 7) & ~7u);     // u32[16]
    static constexpr uint32_t misc = wsum + 64;                                // u32[16]
    static constexpr uint32_t total = misc + 64;
};
static_assert(8 * DecCfg::MAXG * sizeof(DecSym) <= 6 * DecCfg::S, "symbol records must fit the dead tables");
static_assert(DecLds::total <= 160 * 1024, "LDS budget");


__global__ __launch_bounds__(DecCfg::T) void dec_fast_kernel(const uint8_t* __restrict__ container,
                                                             const FrameInfo* __restrict__ frames,
                                                             uint8_t* __restrict__ outbuf,
                                                             int32_t* __restrict__ status)
{
    using C = DecCfg;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* const s_raw = lds + DecLds::sbuf;
    uint16_t* const nx1 = reinterpret_cast<uint16_t*>(lds + DecLds::nx1);
    uint16_t* const ja = reinterpret_cast<uint16_t*>(lds + DecLds::ja);
    uint16_t* const jb = reinterpret_cast<uint16_t*>(lds + DecLds::jb);
    uint16_t* const gl = reinterpret_cast<uint16_t*>(lds + DecLds::gl);
    DecSym* const syms = reinterpret_cast<DecSym*>(lds + DecLds::syms);
    uint16_t* const srcp = reinterpret_cast<uint16_t*>(lds + DecLds::srcp);
    uint8_t* const o_raw = lds + DecLds::obuf;
    uint16_t* const gstart = reinterpret_cast<uint16_t*>(lds + DecLds::gstart);
    uint16_t* const glen = reinterpret_cast<uint16_t*>(lds + DecLds::glen);
    uint32_t* const gout = reinterpret_cast<uint32_t*>(lds + DecLds::gout);
    uint16_t* const sn = reinterpret_cast<uint16_t*>(lds + DecLds::sn);
    uint32_t* const wsum = reinterpret_cast<uint32_t*>(lds + DecLds::wsum);
    uint32_t* const misc = reinterpret_cast<uint32_t*>(lds + DecLds::misc);
    // misc[0] n super nodes, [1] groups in chunk, [2] first group over the image budget,
    // [3] group that completes the block, [4] error, [5] exit offset of the chain

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (*status != 0) return;
    const FrameInfo f = frames[blockIdx.x];
    const uint8_t* const in = container + f.stream_at;
    uint8_t* const out = outbuf + f.out_at;
    const uint32_t in_len = f.stream_len, size = f.out_len, ext = f.ext;

#ifdef TSQ_STATS
    unsigned long long st_[16] = {0};
#endif
    TSQD_T0();
    uint32_t sp = 3;      // stream position of the next group start
    uint32_t op = 0;      // output position reached
    if (tid == 0) misc[4] = 0;
    __syncthreads();

    while (op < size) {
        // ---------------- P0: stage the chunk.  sbuf[k] = in[sp + k]; zeros beyond the stream.
        const uint32_t avail = in_len - sp;                       // stream bytes left from sp (sp < in_len checked below)
        const uint32_t slim = avail < C::S ? avail : C::S;        // offsets >= slim are not group starts of this chunk
        const uint32_t skew = (uint32_t)((uintptr_t)(in + sp) & 15u);
        uint8_t* const sbuf = s_raw + skew;
        {
            const uint8_t* gbase = in + sp - skew;                // 16-byte aligned
            const uint32_t want = skew + (avail < C::S + C::SPAD ? avail : C::S + C::SPAD);
            for (uint32_t w = tid; w < (C::S + C::SPAD + 32) / 16; w += C::T) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if ((w << 4) < want) v = *reinterpret_cast<const uint4*>(gbase + (w << 4));
                *reinterpret_cast<uint4*>(s_raw + (w << 4)) = v;
            }
        }
        if (tid == 0) { misc[0] = 0; misc[1] = 0; misc[2] = 0xFFFFFFFFu; misc[3] = 0xFFFFFFFFu; misc[5] = 0; }
        __syncthreads();
        // bytes staged past the end of this block's stream belong to the next frame: mask them
        // so that a speculative parse of the final group cannot depend on them
        for (uint32_t k = tid; k < C::SPAD + 16; k += C::T) { uint32_t o = slim + k; if (o >= avail && o < C::S + C::SPAD + 16) sbuf[o] = 0; }
        __syncthreads();

        TSQD_ACC(0); TSQD_CNT(12, 1);
        // ---------------- P1: speculative group parse at every offset
        for (uint32_t o = tid; o < C::S; o += C::T) {
            uint32_t nxt, olen = 0;
            if (o < slim) {
                const uint32_t c = sbuf[o];
                uint32_t p = o + 1;
#pragma unroll
                for (uint32_t pr = 0; pr < 4; ++pr) {
                    const uint32_t sb = sbuf[p];
                    const uint32_t hi = sb >> 4, lo = sb & 15u;
                    const uint32_t lit_hi = (c >> (7u - 2u * pr)) & 1u, lit_lo = (c >> (6u - 2u * pr)) & 1u;
                    const uint32_t len_hi = (!lit_hi && ext && hi < 3u) ? (hi + 2u) << 4 : hi + 1u;
                    const uint32_t len_lo = (!lit_lo && ext && lo < 3u) ? (lo + 2u) << 4 : lo + 1u;
                    p += 1u + (lit_hi ? hi + 1u : 2u) + (lit_lo ? lo + 1u : 2u);
                    olen += len_hi + len_lo;
                }
                nxt = p;
            } else {
                nxt = C::S + C::SPAD;            // never a group start: leave the chunk
            }
            nx1[o] = (uint16_t)nxt;
            gl[o] = (uint16_t)olen;
        }
        __syncthreads();

        TSQD_ACC(1);
        // ---------------- P2: J = next^(2^D)
        {
            const uint16_t* src = nx1;
            uint16_t* dst = ja;
#pragma unroll
            for (uint32_t d = 0; d < C::D; ++d) {
                for (uint32_t o = tid; o < C::S; o += C::T) {
                    uint32_t x = src[o];
                    if (x < slim) x = src[x];
                    dst[o] = (uint16_t)x;
                }
                __syncthreads();
                src = dst;
                dst = (dst == ja) ? jb : ja;
            }
        }
        const uint16_t* const J = (C::D & 1u) ? ja : jb;          // D passes: ja, jb, ja, jb ...

        TSQD_ACC(2);
        // ---------------- P3: one lane follows the chain of super nodes
        if (tid == 0) {
            uint32_t x = 0, k = 0;
            while (x < slim && k < C::MAXSN) { sn[k++] = (uint16_t)x; x = J[x]; }
            misc[0] = k;
            if (k >= C::MAXSN && x < slim) misc[4] = kErrStream;  // cannot happen for S/13 groups
        }
        __syncthreads();
        const uint32_t nsn = misc[0];

        TSQD_ACC(3);
        // ---------------- P4: expand super nodes into group starts; scan group lengths
        if (tid < nsn) {
            uint32_t x = sn[tid], cnt = 0;
            for (uint32_t t = 0; t < C::HOP && x < slim; ++t) {
                gstart[tid * C::HOP + t] = (uint16_t)x;
                x = nx1[x];
                cnt++;
            }
            if (tid == nsn - 1) { misc[1] = (nsn - 1) * C::HOP + cnt; misc[5] = x; }
        }
        __syncthreads();
        uint32_t ng = misc[1];
        {
            // exclusive scan of glen over the ng (<= MAXG <= T) groups
            uint32_t v = 0;
            if (tid < ng) { v = gl[gstart[tid]]; glen[tid] = (uint16_t)v; }
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
                if (excl + 512u > C::OUTC) atomicMin(&misc[2], tid);                 // does not fit the image
                if (op + excl + v >= size) atomicMin(&misc[3], tid);                 // completes the block
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
        if (ng == 0 || (!last_chunk && next_sp >= in_len)) {       // stream exhausted before the block is complete
            if (tid == 0) atomicMax(status, kErrStream);
            return;
        }
        const uint32_t image_len = next_op - op;                   // bytes this chunk produces (<= OUTC)
        const uint32_t oskew = (uint32_t)((uintptr_t)(out + op) & 15u);
        uint8_t* const obuf = o_raw + oskew;                       // obuf[q] <-> out[op + q], mutually 16-byte aligned
        __syncthreads();                                           // tables dead from here: syms may overwrite them

        TSQD_ACC(4);
        // ---------------- P5: symbol records, one lane per group
        if (tid < ng) {
            const uint32_t x = gstart[tid];
            const uint32_t c = sbuf[x];
            uint32_t p = x + 1, j = gout[tid];
            uint32_t bad = 0;
            DecSym* rec = syms + tid * 8u;
#pragma unroll
            for (uint32_t pr = 0; pr < 4; ++pr) {
                uint32_t sb = 0;
                const uint32_t origin = j;
                if (j < size) { if (p >= avail) bad = 1; sb = sbuf[p]; p++; }
#pragma unroll
                for (uint32_t s = 0; s < 2; ++s) {
                    DecSym r; r.out_rel = 0; r.len = 0; r.kind = 0; r.a = 0;
                    if (j < size && !bad) {
                        const uint32_t nib = s == 0 ? sb >> 4 : sb & 15u;
                        const uint32_t lit = (c >> (7u - (2u * pr + s))) & 1u;
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
                            if (off > origin || take > off) bad = 1;          // before the block / overlaps the pair
                            r.out_rel = (uint16_t)(j - op); r.len = (uint8_t)take; r.kind = 2; r.a = origin - off;
                            j += take;
                        }
                        if (bad) r.kind = 0;
                    }
                    rec[pr * 2u + s] = r;
                }
            }
            if (bad) misc[4] = kErrStream;
        }
        __syncthreads();
        if (misc[4] != 0) { if (tid == 0) atomicMax(status, (int32_t)misc[4]); return; }

        TSQD_ACC(5);
        // ---------------- P6a: scatter literal and history bytes; record in-chunk sources
        for (uint32_t s = tid; s < ng * 8u; s += C::T) {
            const DecSym r = syms[s];
            if (r.kind == 1) {
                for (uint32_t t = 0; t < r.len; ++t) { obuf[r.out_rel + t] = sbuf[r.a + t]; srcp[r.out_rel + t] = C::RES; }
            } else if (r.kind == 2) {
                // source bytes below `op` are final in HBM: fetch them 16 at a time (reading past `op` inside this
                // block's own output region is harmless: those bytes are simply not used); source bytes at or
                // beyond `op` belong to this chunk and are recorded as pointers
                const uint32_t sa = r.a;
                for (uint32_t t0 = 0; t0 < r.len; t0 += 16u) {
                    const uint32_t a0 = sa + t0;
                    const uint32_t cnt = r.len - t0 < 16u ? r.len - t0 : 16u;
                    if (a0 < op && a0 + 16u <= size) {
                        TSQD_CNT(9, 1);
                        uint4 v;
                        __builtin_memcpy(&v, out + a0, 16);
                        const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
                        for (uint32_t t = 0; t < cnt; ++t) {
                            const uint32_t a = a0 + t, q = r.out_rel + t0 + t;
                            if (a < op) { obuf[q] = (uint8_t)(t < 8u ? lo >> (8u * t) : hi >> (8u * (t - 8u))); srcp[q] = C::RES; }
                            else srcp[q] = (uint16_t)(a - op);
                        }
                    } else {
                        for (uint32_t t = 0; t < cnt; ++t) {
                            const uint32_t a = a0 + t, q = r.out_rel + t0 + t;
                            if (a < op) { obuf[q] = out[a]; srcp[q] = C::RES; TSQD_CNT(10, 1); }
                            else srcp[q] = (uint16_t)(a - op);
                        }
                    }
                }
            }
        }
        __syncthreads();

        TSQD_ACC(6);
        // ---------------- P6b: pointer jumping until every byte of the image is final
        {
            // each thread owns image bytes q = tid + r*T and remembers which of them still hold a pointer;
            // a round follows two links at once (all reads before all writes), so chains shrink 3x per round
            uint32_t pending = 0;
#pragma unroll
            for (uint32_t r = 0; r < C::OUTC / C::T; ++r) {
                const uint32_t q = tid + r * C::T;
                if (q < image_len && srcp[q] != C::RES) pending |= 1u << r;
            }
            TSQD_CNT(11, __builtin_popcount(pending)); TSQD_CNT(15, image_len);
            for (uint32_t round = 0; round < 24; ++round) {
                if (!__syncthreads_or((int)(pending != 0u))) break;
                TSQD_CNT(13, 1);
                uint16_t np[C::OUTC / C::T];
                uint8_t nv[C::OUTC / C::T];
#pragma unroll
                for (uint32_t r = 0; r < C::OUTC / C::T; ++r) {
                    np[r] = C::RES; nv[r] = 0;
                    if (pending & (1u << r)) {
                        const uint32_t p1 = srcp[tid + r * C::T];
                        const uint32_t p2 = srcp[p1];
                        if (p2 == C::RES) { nv[r] = obuf[p1]; }
                        else {
                            const uint32_t p3 = srcp[p2];
                            if (p3 == C::RES) nv[r] = obuf[p2];
                            else np[r] = (uint16_t)p3;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (uint32_t r = 0; r < C::OUTC / C::T; ++r) {
                    if (pending & (1u << r)) {
                        const uint32_t q = tid + r * C::T;
                        if (np[r] == C::RES) { obuf[q] = nv[r]; srcp[q] = C::RES; pending &= ~(1u << r); }
                        else srcp[q] = np[r];
                    }
                }
            }
        }

        TSQD_ACC(7);
        // ---------------- P7: image -> HBM (head bytes, aligned 16-byte words, tail bytes)
        {
            const uint32_t head = (16u - oskew) & 15u;
            const uint32_t hb = head < image_len ? head : image_len;
            if (tid < hb) out[op + tid] = obuf[tid];
            const uint32_t words = image_len > hb ? (image_len - hb) >> 4 : 0;
            for (uint32_t w = tid; w < words; w += C::T)
                *reinterpret_cast<uint4*>(out + op + hb + (w << 4)) = *reinterpret_cast<const uint4*>(obuf + hb + (w << 4));
            const uint32_t tail_at = hb + (words << 4);
            if (tid < image_len - tail_at) out[op + tail_at + tid] = obuf[tail_at + tid];
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
