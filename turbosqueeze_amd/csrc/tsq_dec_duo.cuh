// tsq_dec_duo.cuh -- block decoder on TWO workgroups per block (kernel variant 3; chosen by itself when a launch has at most half as
// many blocks as the device has CUs -- every GPU of a multi-GPU enwik9 job).
//
// A block's decode is two serial chains (tsq_decode.cpp:62-88): the PARSE chain -- where a chunk of the stream starts follows from
// where the previous one ended -- and the COPY chain -- a chunk's bytes may copy the 64 KiB before them.  dec_sym_kernel runs both
// on one CU, one after the other per chunk.  Here they run on two CUs of the same XCD, side by side:
//
//   PARSE workgroup  phases P0..P4 of tsq_dec_sym.cuh (stage the chunk, speculative group parse at every offset, pointer doubling,
//                    the chain from the known start, one lane per group -> pair words and output offsets), then hands the chunk's
//                    pair words, group output offsets and a header to the COPY workgroup through a four-deep ring of records in
//                    global memory (they stay in the XCD's L2) and moves on to the next chunk at once.
//   COPY workgroup   phase P5 of tsq_dec_sym.cuh (pairs -> records -> bytes, history in the 64 KiB LDS ring) and the stores to HBM.
//
// Workgroup w runs on XCD w % 8 (the hardware deals workgroups round-robin): w -> (xcd = w % 8, slot = w / 8), role = slot & 1,
// block = (slot / 2) * 8 + xcd puts a block's two workgroups on one XCD, the PARSE one dispatched first.  Every wait is bounded and
// also ends when another block has reported an error.
#pragma once

#include <type_traits>

#include "tsq_common.cuh"
#include "tsq_dec_common.cuh"
#include "tsq_dec_sym.cuh"

namespace tsq {

struct DuoCfg {
    static constexpr uint32_t SLOTS = 4;                                   // chunk records in flight per block
    static constexpr uint32_t HDR = 8;                                     // header words: 0 sp, 1 op, 2 groups, 3 next op, 4 flags, 5 next sp
    static constexpr uint32_t REC_WORDS = ((HDR + 5 * SymCfg::MAXG) + 63u) & ~63u;
    static constexpr uint32_t FLAG_STRIDE = 64;                            // u32 words per block: [0] records produced, [32] records consumed,
                                                                           // [8 + 8 p ..]: the entry mailbox of PARSE workgroup p (sequence, delta | flags << 16, op, record)
    static constexpr uint32_t kLast = 1, kError = 2;
    static constexpr uint32_t kAbort = 0xFFFFFFFFu;
    static constexpr uint32_t kSpinLimit = 1u << 24;                       // default number of polls (with s_sleep) before a wait gives up: seconds
                                                                           // (the launch passes the context's limit: tsqa_set_decode_wait_limit)
};

// polls are relaxed loads (an acquire load invalidates the CU's vector cache every time: hundreds of polling workgroups would keep
// the L2 busy with nothing); one acquire fence follows the poll that succeeds
__device__ __forceinline__ uint32_t duo_load_acquire(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void duo_acquire_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void duo_store_release(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------------------------------------------------------ PARSE
// The stream is parsed in FIXED windows: window k covers stream offsets [3 + k S, 3 + (k + 1) S + pad).  The speculative part of the
// parse (P0..P2: every offset parsed as if a group started there, pointer doubling) does not depend on where the chunk's first group
// really starts, only the chain (P3) and the groups (P4) do -- and the first group of window k + 1 starts less than 133 bytes into it
// (where the last group of window k ends).  So NP PARSE workgroups take the windows in turn: each prepares its window on its own and
// waits only for the ENTRY (offset of the first group, output position, record number) that the workgroup of the window before
// publishes as soon as its own chain has reached its window's end.  The block's serial parse chain is then P3 + P4 per window.
// NP = 1: the same code, the entry is handed over in registers.
template <uint32_t NP>
__device__ __forceinline__ void duo_parse(const uint8_t* __restrict__ in, uint32_t in_len, uint32_t size, uint32_t ext, uint32_t me,
                                          uint32_t* __restrict__ ring_g, uint32_t* __restrict__ flags, int32_t* __restrict__ status, uint8_t* lds,
                                          uint32_t spin_limit)
{
    using C = SymCfg;
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
    uint16_t* const lut = reinterpret_cast<uint16_t*>(lds + SymLds::lut);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;

    if (size == 0u) return;                              // (an empty block has no chunks: no workgroup has anything to do)
    if (tid == 0) misc[4] = 0;
    { uint32_t sl, ol; pair_lens(tid & 255u, tid >> 8, 0u, sl, ol); lut[tid] = (uint16_t)sl; }
    // the entry of this workgroup's next window, when it is handed over in registers (NP == 1, or window 0)
    uint32_t e_delta = 0, e_op = 0, e_rec = 0;
    bool e_known = me == 0u;
    uint32_t* const entry_in = flags + 8u + 8u * me;                        // [0] sequence, [1] delta | flags << 16, [2] op, [3] record number
    uint32_t* const entry_out = flags + 8u + 8u * ((me + 1u) % NP);
    uint32_t n_in = 0, n_out = 0;                                           // entries received / sent so far
    // whole 16-byte words of window q, loaded an iteration ahead (the windows lie at fixed places)
    uint4 pre = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](uint32_t q) {
        const uint64_t at = 3ull + (uint64_t)q * C::S;
        pre = make_uint4(0, 0, 0, 0);
        if (at >= in_len) return;
        const uint32_t av = in_len - (uint32_t)at;
        const uint32_t lim = av < C::S + C::SPAD ? av : C::S + C::SPAD;
        if (tid < C::SWORDS && (tid << 4) + 16u <= lim) __builtin_memcpy(&pre, in + at + (tid << 4), 16);
    };
    prefetch(me);
    auto give_up = [&](uint32_t why) {
        // (uniform: every thread calls it) tell the others and leave
        if (tid == 0) {
            if (why == 2u) atomicMax(status, kErrStall);       // (a sibling workgroup did not show up in time: not the stream's fault)
            duo_store_release(flags + 32, DuoCfg::kAbort);
        }
    };
    __syncthreads();

    for (uint32_t k = me;; k += NP) {
        const uint64_t w64 = 3ull + (uint64_t)k * C::S;
        const bool beyond = w64 >= in_len;                                   // no such window: only the "finished" entry can come
        const uint32_t W = beyond ? in_len : (uint32_t)w64;
        const uint32_t avail = in_len - W;
        const uint32_t slim = avail < C::S ? avail : C::S;
        uint8_t* const sbuf = s_raw;
        if (!beyond) {
            // ---------------- P0: the window (loaded an iteration ago) goes to LDS.  sbuf[q] = in[W + q]; zeros beyond the stream.
            if (tid < C::SWORDS) {
                uint4 w = pre;
                const uint32_t lim = avail < C::S + C::SPAD ? avail : C::S + C::SPAD, o = tid << 4;
                if (o < lim && o + 16u > lim) {                                  // the stream's last, partial word (once per block)
                    uint32_t b[4] = {0, 0, 0, 0};
                    for (uint32_t q = 0; o + q < lim; ++q) b[q >> 2] |= (uint32_t)in[W + o + q] << (8u * (q & 3u));
                    w = make_uint4(b[0], b[1], b[2], b[3]);
                }
                *reinterpret_cast<uint4*>(s_raw + (tid << 4)) = w;
            }
            prefetch(k + NP);                                                    // this workgroup's next window is on its way
            __syncthreads();
            // ---------------- P1 + P2: speculative group parse at every offset, then next^2 .. next^16 (tsq_dec_sym.cuh)
            uint32_t x[C::PER], y[C::PER], c[C::PER];
#pragma unroll
            for (uint32_t q = 0; q < C::PER; ++q) { const uint32_t o = tid + q * C::T; c[q] = sbuf[o]; x[q] = o + 1u; }
#pragma unroll
            for (uint32_t pr = 0; pr < 4; ++pr) {
#pragma unroll
                for (uint32_t q = 0; q < C::PER; ++q) y[q] = sbuf[x[q]];
#pragma unroll
                for (uint32_t q = 0; q < C::PER; ++q) y[q] = lut[(((c[q] >> (6u - 2u * pr)) & 3u) << 8) | y[q]];
#pragma unroll
                for (uint32_t q = 0; q < C::PER; ++q) x[q] += y[q];
            }
#pragma unroll
            for (uint32_t q = 0; q < C::PER; ++q) { const uint32_t o = tid + q * C::T; j1[o] = (uint8_t)(x[q] - o); x[q] = o < slim ? x[q] : C::TERM; }
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < C::PER; ++q) { const uint32_t a = x[q] < slim ? x[q] : 0u; y[q] = a + j1[a]; }
#pragma unroll
            for (uint32_t q = 0; q < C::PER; ++q) { x[q] = x[q] < slim ? y[q] : C::TERM; j2[tid + q * C::T] = (uint16_t)x[q]; }
            __syncthreads();
            const uint16_t* srcj = j2;
            uint16_t* const dsts[3] = {j4, j8, j16};
#pragma unroll
            for (uint32_t d = 0; d < 3; ++d) {
#pragma unroll
                for (uint32_t q = 0; q < C::PER; ++q) y[q] = srcj[x[q] < slim ? x[q] : 0u];
#pragma unroll
                for (uint32_t q = 0; q < C::PER; ++q) { x[q] = x[q] < slim ? y[q] : C::TERM; dsts[d][tid + q * C::T] = (uint16_t)x[q]; }
                __syncthreads();
                srcj = dsts[d];
            }
        }
        // ---------------- the window's entry: where its first group starts, the output position there, the record number
        uint32_t delta, op, rec_no;
        if (e_known) { delta = e_delta; op = e_op; rec_no = e_rec; e_known = false; }
        else {
            if (tid == 0) {
                uint32_t why = 0, spins = 0;
                for (;;) {
                    if (duo_load_acquire(entry_in) > n_in) break;
                    if ((spins & 63u) == 63u && (duo_load_acquire(flags + 32) == DuoCfg::kAbort || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { why = 1; break; }
                    if (++spins > spin_limit) { why = 2; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                duo_acquire_fence();
                misc[6] = why;
                if (!why) { misc[7] = __builtin_nontemporal_load(entry_in + 1); misc[8] = __builtin_nontemporal_load(entry_in + 2); misc[9] = __builtin_nontemporal_load(entry_in + 3); }
            }
            __syncthreads();
            if (misc[6] != 0) { give_up(misc[6]); return; }
            n_in++;
            const uint32_t d = misc[7];
            if (d >> 16) return;                                                 // finished (or failed) in an earlier window
            delta = d & 0xFFFFu; op = misc[8]; rec_no = misc[9];
            __syncthreads();                                                     // (misc[7..9] are read: they may be written again)
        }
        if (beyond) {                                                            // the chain runs past the stream's end: malformed
            if (tid == 0) atomicMax(status, kErrStream);
            // tell the COPY workgroup (through a record, in order)
            // (fall through to the hand-over below with `bad`)
        }
        // ---------------- the chunks of this window: normally one; more when the LDS image budget of the COPY workgroup cuts one short
        for (;;) {
            if (tid == 0) { misc[0] = 0; misc[1] = 0; misc[2] = 0xFFFFFFFFu; misc[3] = 0xFFFFFFFFu; misc[5] = 0; }
            __syncthreads();
            // ---------------- P3: one lane follows next^16 from the chunk's first group
            if (tid == 0 && !beyond) {
                uint32_t x = delta, q = 0;
                while (x < slim && q < C::MAXSN) { sn[q++] = (uint16_t)x; x = j16[x]; }
                misc[0] = q;
                if (q >= C::MAXSN && x < slim) misc[4] = kErrStream;
            }
            __syncthreads();
            const uint32_t nsn = misc[0];
            // ---------------- P4: one lane per group
            {
                uint32_t x = C::TERM;
                if (tid < nsn * C::HOP) {
                    x = sn[tid >> 4];
                    if (tid & 8u) x = x < slim ? j8[x] : C::TERM;
                    if (tid & 4u) x = x < slim ? j4[x] : C::TERM;
                    if (tid & 2u) x = x < slim ? j2[x] : C::TERM;
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
                        pw[pr] = p | (v << 13) | (cc << 30);
                        uint32_t sl, ol;
                        pair_lens(sbuf[p], cc, ext, sl, ol);
                        p += sl;
                        v += ol;
                    }
                    *reinterpret_cast<uint4*>(pairs + tid * 4u) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
                    glen[tid] = (uint16_t)v;
                    if (p >= slim) { misc[1] = tid + 1u; misc[5] = p; }
                }
                const uint32_t incl = wave_scan_add(v);
                if (lane == 63) wsum[wid] = incl;
                __syncthreads();
                const uint32_t totals = wave_scan_add(lane < C::T / 64u ? wsum[lane] : 0u);
                const uint32_t before = wid ? (uint32_t)__builtin_amdgcn_readlane((int)totals, (int)wid - 1) : 0u;
                const uint32_t excl = before + incl - v;
                if (x < slim) {
                    gout[tid] = op + excl;
                    if (excl + 512u + 16u > C::OUTC) atomicMin(&misc[2], tid);
                    if (op + excl + v >= size) atomicMin(&misc[3], tid);
                }
            }
            __syncthreads();
            uint32_t ng = misc[1];
            uint32_t next_op, next_delta = 0;
            bool last_chunk = false, cut_short = false;
            {
                const uint32_t cut = misc[2], fin = misc[3];
                if (fin != 0xFFFFFFFFu && fin < cut) { ng = fin + 1; last_chunk = true; next_op = size; }
                else if (cut != 0xFFFFFFFFu) { ng = cut; cut_short = true; next_delta = gstart[cut]; next_op = gout[cut]; }
                else { next_delta = misc[5] - C::S; next_op = ng ? gout[ng - 1] + glen[ng - 1] : op; }
            }
            // (a chunk that neither completes the block nor is cut short must have run to the end of a FULL window)
            const bool bad = beyond || misc[4] != 0 || ng == 0 || (!last_chunk && !cut_short && (slim < C::S || misc[5] < C::S || next_delta >= C::SPAD));
            // ---------------- the next window's workgroup gets its entry as early as possible
            if (!cut_short || bad) {
                const uint32_t fl = bad ? 2u : last_chunk ? 1u : 0u;
                if (NP == 1u) {
                    if (fl) { /* nothing to hand over */ } else { e_delta = next_delta; e_op = next_op; e_rec = rec_no + 1u; e_known = true; }
                } else {
                    if (tid == 0) {
                        __builtin_nontemporal_store((fl << 16) | next_delta, entry_out + 1);
                        __builtin_nontemporal_store(next_op, entry_out + 2);
                        __builtin_nontemporal_store(rec_no + 1u, entry_out + 3);
                        duo_store_release(entry_out, n_out + 1u);
                    }
                    n_out++;
                }
            }
            // ---------------- hand the chunk to the COPY workgroup: records go out in order, into a free slot
            if (tid == 0) {
                uint32_t why = 0, spins = 0;
                for (;;) {
                    const uint32_t done = duo_load_acquire(flags + 32);
                    if (done == DuoCfg::kAbort || ((spins & 255u) == 255u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { why = 1; break; }
                    if (rec_no - done < DuoCfg::SLOTS && (NP == 1u || duo_load_acquire(flags) == rec_no)) break;
                    if (++spins > spin_limit) { why = 2; break; }
                    __builtin_amdgcn_s_sleep(8);                                   // (nobody waits for this workgroup while its ring is full)
                }
                duo_acquire_fence();
                misc[6] = why;
            }
            __syncthreads();
            if (misc[6] != 0) { give_up(misc[6]); return; }
            uint32_t* const rec = ring_g + (size_t)(rec_no % DuoCfg::SLOTS) * DuoCfg::REC_WORDS;
            if (!bad) {
                for (uint32_t w = tid; w < 4u * ng; w += C::T) rec[DuoCfg::HDR + w] = pairs[w];
                for (uint32_t w = tid; w < ng; w += C::T) rec[DuoCfg::HDR + 4u * C::MAXG + w] = gout[w];
            }
            if (tid == 0) {
                rec[0] = W; rec[1] = op; rec[2] = ng; rec[3] = next_op;
                rec[4] = (last_chunk ? DuoCfg::kLast : 0u) | (bad ? DuoCfg::kError : 0u);
                rec[5] = cut_short ? W : W + C::S;                            // where the next record's window starts (for the COPY side's prefetch)
            }
            // Every wavefront waits for ITS record stores to be acknowledged before the barrier (the workgroup-scope fence of
            // __syncthreads() does not: no s_waitcnt vmcnt(0)), so that they are complete when thread 0 releases at agent scope (its
            // release writes the L2 back once for the whole workgroup; an agent-scope release fence in every thread -- tried -- costs
            // a write-back per wavefront and more than doubled the kernel's time).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) duo_store_release(flags, rec_no + 1u);            // ... and thread 0 publishes them
            if (bad) { if (tid == 0) atomicMax(status, kErrStream); return; }
            if (last_chunk) return;
            if (!cut_short) break;
            delta = next_delta; op = next_op; rec_no++;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------- COPY
struct DuoCopyLds {
    static constexpr uint32_t sbuf = 0;                                             // u8[S + SPAD + 16]
    static constexpr uint32_t recw = sbuf + 16 * SymCfg::SWORDS;                     // u32[OUTC + 16] symbol records, then u16 byte entries over them
    static constexpr uint32_t ent = recw;
    static constexpr uint32_t plist = ent + 2 * (SymCfg::OUTC + 16);                // u16[OUTC] waiting lists
    static constexpr uint32_t wsum = recw + 4 * (SymCfg::OUTC + 16);                // u32[16]
    static constexpr uint32_t misc = wsum + 64;                                     // u32[16]
    static constexpr uint32_t ring = (misc + 64 + 15) & ~15u;                       // u8[R + RPAD]
    static constexpr uint32_t total = ring + SymCfg::R + SymCfg::RPAD;
    static_assert(recw % 16 == 0 && plist % 16 == 0 && plist + 2 * SymCfg::OUTC <= wsum, "records, byte entries and waiting lists");
};
static_assert(DuoCopyLds::total <= 160 * 1024 && SymLds::total <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void duo_copy(const uint8_t* __restrict__ in, uint32_t in_len, uint32_t size, uint32_t ext, uint8_t* __restrict__ out,
                                         const uint32_t* __restrict__ ring_g, uint32_t* __restrict__ flags, int32_t* __restrict__ status, uint8_t* lds,
                                         uint32_t spin_limit)
{
    using C = SymCfg;
    uint8_t* const sbuf = lds + DuoCopyLds::sbuf;
    uint32_t* const recw = reinterpret_cast<uint32_t*>(lds + DuoCopyLds::recw);
    uint32_t* const wsum = reinterpret_cast<uint32_t*>(lds + DuoCopyLds::wsum);
    uint32_t* const misc = reinterpret_cast<uint32_t*>(lds + DuoCopyLds::misc);
    uint8_t* const ring = lds + DuoCopyLds::ring;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    // ring address of output position p: (p + oskew) mod R, so that 16-byte words of the ring are 16-byte words of HBM
    const uint32_t oskew = (uint32_t)((uintptr_t)out & 15u);
    uint32_t ring_op = oskew;
    uint32_t prev_op = 0, prev_len = 0, prev_ring = 0;
    uint32_t k = 0, made_seen = 0;
    uint32_t hdr_n[6] = {0, 0, 0, 0, 0, 0}, pw_n[2] = {0, 0}, go_n[2] = {0, 0};
    bool have_next = false;
    if (size == 0u) return;
    if (tid == 0) { misc[4] = 0; }

    auto flush_image = [&]() {
        // image bytes [prev_op, prev_op + prev_len) from the ring to HBM: head bytes up to the first aligned word, aligned
        // 16-byte words, tail bytes
        if (prev_len == 0) return;
        const uint32_t head = (16u - (prev_ring & 15u)) & 15u;
        const uint32_t hb = head < prev_len ? head : prev_len;
        if (tid < hb) out[prev_op + tid] = ring[prev_ring + tid];
        const uint32_t words = (prev_len - hb) >> 4;
        uint32_t ra = prev_ring + hb; ra -= ra >= C::R ? C::R : 0u;
        for (uint32_t w = tid; w < words; w += C::T) {
            uint32_t a = ra + (w << 4); a -= a >= C::R ? C::R : 0u;
            *reinterpret_cast<uint4*>(out + prev_op + hb + (w << 4)) = *reinterpret_cast<const uint4*>(ring + a);
        }
        const uint32_t tail_at = hb + (words << 4);
        if (tid < prev_len - tail_at) { uint32_t a = ra + (words << 4) + tid; a -= a >= C::R ? C::R : 0u; out[prev_op + tail_at + tid] = ring[a]; }
    };
    uint4 pre = make_uint4(0, 0, 0, 0);
    uint32_t pre_sp = 0xFFFFFFFFu;                                                    // the stream offset `pre` was loaded from
    auto prefetch = [&](uint32_t at) {
        const uint32_t avail = in_len - at;
        const uint32_t lim = avail < C::S + C::SPAD ? avail : C::S + C::SPAD;
        pre = make_uint4(0, 0, 0, 0);
        if (tid < C::SWORDS && (tid << 4) + 16u <= lim) __builtin_memcpy(&pre, in + at + (tid << 4), 16);
        pre_sp = at;
    };
    prefetch(3u);
    __syncthreads();

    for (;;) {
        // ---------------- the previous chunk's bytes go to HBM; thread 0 waits for this chunk's record, if it is not known to be there
        flush_image();
        if (made_seen <= k) {
            if (tid == 0) {
                uint32_t give_up = 0, spins = 0, made = 0;
                for (;;) {
                    made = duo_load_acquire(flags);
                    if (made > k) break;
                    if ((spins & 255u) == 255u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { give_up = 1; break; }
                    if (++spins > spin_limit) { give_up = 2; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                duo_acquire_fence();
                misc[6] = give_up; misc[7] = made;
            }
            __syncthreads();
            if (misc[6] != 0) {
                if (tid == 0) { duo_store_release(flags + 32, DuoCfg::kAbort); if (misc[6] == 2) atomicMax(status, kErrStall); }
                return;
            }
            made_seen = misc[7];
            have_next = false;
        }
        const uint32_t* const rec = ring_g + (size_t)(k % DuoCfg::SLOTS) * DuoCfg::REC_WORDS;
        // the record's header, this thread's pair words and group output offsets: loaded during the previous chunk when the record
        // was already there (the PARSE side runs up to four chunks ahead), else now
        if (!have_next) {
#pragma unroll
            for (uint32_t q = 0; q < 6; ++q) hdr_n[q] = __builtin_nontemporal_load(rec + q);
#pragma unroll
            for (uint32_t rep = 0; rep < 2; ++rep) {
                const uint32_t gi = tid + rep * C::T;
                pw_n[rep] = rec[DuoCfg::HDR + gi];
                go_n[rep] = rec[DuoCfg::HDR + 4u * C::MAXG + (gi >> 2)];
            }
        }
        const uint32_t sp = hdr_n[0], op = hdr_n[1], ng = hdr_n[2], next_op = hdr_n[3], flg = hdr_n[4], next_sp = hdr_n[5];
        const uint32_t pw_g[2] = {pw_n[0], pw_n[1]}, go_g[2] = {go_n[0], go_n[1]};
        have_next = false;
        if (flg & DuoCfg::kError) return;                                              // (the PARSE workgroup has reported it)
        const bool last_chunk = (flg & DuoCfg::kLast) != 0u;
        const uint32_t image_len = next_op - op;
        const uint32_t avail = in_len - sp;
        // ---------------- the chunk's stream to LDS (prefetched a chunk ago when the header was there in time), the record words cleared
        if (pre_sp != sp) prefetch(sp);
        if (tid < C::SWORDS) {
            uint4 w = pre;
            const uint32_t lim = avail < C::S + C::SPAD ? avail : C::S + C::SPAD, o = tid << 4;
            if (o < lim && o + 16u > lim) {
                uint32_t b[4] = {0, 0, 0, 0};
                for (uint32_t q = 0; o + q < lim; ++q) b[q >> 2] |= (uint32_t)in[sp + o + q] << (8u * (q & 3u));
                w = make_uint4(b[0], b[1], b[2], b[3]);
            }
            *reinterpret_cast<uint4*>(sbuf + (tid << 4)) = w;
        }
        for (uint32_t w = tid; w < (C::OUTC + 16) / 4; w += C::T) *reinterpret_cast<uint4*>(recw + 4u * w) = make_uint4(0, 0, 0, 0);
        if (!last_chunk) prefetch(next_sp);
        __syncthreads();

        // ---------------- P5 (tsq_dec_sym.cuh): symbols -> bytes
        const uint32_t lead = ring_op & 3u;
        const uint32_t a0 = ring_op - lead;
        if (tid == 0 && lead) recw[0] = 0x40000000u | (DuoCopyLds::ring + a0);
        uint32_t bad = 0;
#pragma unroll
        for (uint32_t rep = 0; rep < 2; ++rep) {
            const uint32_t gi = tid + rep * C::T;
            if (gi >= ng * 4u) break;
            const uint32_t pw = pw_g[rep];
            uint32_t p = pw & 0x1FFFu, j = go_g[rep] + ((pw >> 13) & 0x3FFFu);
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
                        else recw[ij] = 0x40000000u | ((DuoCopyLds::sbuf + p - ij) & 0xFFFFFFu);
                        p += len; j += take;
                    } else {
                        if (p + 2u > avail) bad = 1;
                        const uint32_t off = (uint32_t)sbuf[p] | ((uint32_t)sbuf[p + 1] << 8);
                        p += 2;
                        const uint32_t len = sym_out_len(nib, 0, ext);
                        const uint32_t take = len < room ? len : room;
                        if (off > origin || take > off) bad = 1;
                        if (!bad) {
                            const uint32_t a = origin - off;
                            const uint32_t n_hist = a >= op ? 0u : (op - a < take ? op - a : take);
                            if (n_hist) {
                                uint32_t x0 = a0 + C::R - ((op - a) - lead);
                                x0 -= x0 >= C::R ? C::R : 0u;
                                recw[ij] = 0x40000000u | ((DuoCopyLds::ring + x0 - ij) & 0xFFFFFFu);
                                if (x0 + n_hist > C::R) { const uint32_t n1 = C::R - x0; recw[ij + n1] = 0x40000000u | ((DuoCopyLds::ring - (ij + n1)) & 0xFFFFFFu); }
                            }
                            if (n_hist < take) recw[ij + n_hist] = 0xC0000000u | ((a - j) & 0xFFFFFFu);
                        }
                        j += take;
                    }
                }
            }
        }
        if (bad) misc[4] = kErrStream;
        __syncthreads();
        k++;
        if (misc[4] != 0) {
            if (tid == 0) { atomicMax(status, (int32_t)misc[4]); duo_store_release(flags + 32, DuoCfg::kAbort); }
            return;
        }
        // the record's slot is free (every thread holds what it needed from it in registers: the barrier above waited for the loads)
        if (tid == 0) __hip_atomic_store(flags + 32, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!last_chunk && made_seen > k) {                                          // the next record is there already: fetch it now
            const uint32_t* const rn = ring_g + (size_t)(k % DuoCfg::SLOTS) * DuoCfg::REC_WORDS;
#pragma unroll
            for (uint32_t q = 0; q < 6; ++q) hdr_n[q] = __builtin_nontemporal_load(rn + q);
#pragma unroll
            for (uint32_t rep = 0; rep < 2; ++rep) {
                const uint32_t gi = tid + rep * C::T;
                pw_n[rep] = rn[DuoCfg::HDR + gi];
                go_n[rep] = rn[DuoCfg::HDR + 4u * C::MAXG + (gi >> 2)];
            }
            have_next = true;
        }
        typedef __attribute__((address_space(3))) uint16_t lds_u16;
        lds_u16* const le = (lds_u16*)(lds + DuoCopyLds::ent);
        lds_u16* const wl = (lds_u16*)(lds + DuoCopyLds::plist) + 768u * wid;
        const uint32_t own = 12u * tid;
        uint32_t r[12];
        {
            const uint4 q0 = *reinterpret_cast<const uint4*>(recw + own), q1 = *reinterpret_cast<const uint4*>(recw + own + 4u),
                        q2 = *reinterpret_cast<const uint4*>(recw + own + 8u);
            r[0] = q0.x; r[1] = q0.y; r[2] = q0.z; r[3] = q0.w; r[4] = q1.x; r[5] = q1.y; r[6] = q1.z; r[7] = q1.w; r[8] = q2.x; r[9] = q2.y; r[10] = q2.z; r[11] = q2.w;
#pragma unroll
            for (uint32_t q = 1; q < 12; ++q) r[q] = r[q] ? r[q] : r[q - 1];
            const uint32_t key = wave_scan_max(r[11] ? lane + 1u : 0u);
            const uint32_t from = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x138, 0xF, 0xF, false);
            uint32_t carry = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((from ? from - 1u : 0u) << 2), (int)r[11]);
            carry = from ? carry : 0u;
            const uint32_t upto = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((key ? key - 1u : 0u) << 2), (int)r[11]);
            if (lane == 63) wsum[wid] = key ? upto : 0u;
            __syncthreads();
            {
                const uint32_t ws = lane < 16u ? wsum[lane] : 0u;
                const uint64_t m = __ballot(ws != 0u && lane < wid);
                const uint32_t prev = m ? (uint32_t)__builtin_amdgcn_readlane((int)ws, 63 - __builtin_clzll(m)) : 0u;
                carry = carry ? carry : prev;
            }
#pragma unroll
            for (uint32_t q = 0; q < 12; ++q) r[q] = r[q] ? r[q] : carry;
        }
        const bool live = own < lead + image_len;
        uint32_t pend = 0;
        if (live) {
            uint32_t v[12], by[12];
#pragma unroll
            for (uint32_t q = 0; q < 12; ++q) {
                v[q] = own + q + (uint32_t)((int32_t)(r[q] << 8) >> 8);
                pend |= (r[q] >> 31) << q;
            }
#pragma unroll
            for (uint32_t q = 0; q < 12; ++q) by[q] = lds[(r[q] >> 31) ? 0u : v[q]];
#pragma unroll
            for (uint32_t q = 0; q < 12; ++q) v[q] = (r[q] >> 31) ? v[q] : (0x8000u | by[q]);
#pragma unroll
            for (uint32_t w = 0; w < 3; ++w)
                *reinterpret_cast<uint2*>(lds + DuoCopyLds::ent + 2u * own + 8u * w) = make_uint2(v[4 * w] | (v[4 * w + 1] << 16), v[4 * w + 2] | (v[4 * w + 3] << 16));
        }
        uint32_t n_wait;
        {
            const uint32_t cnt = (uint32_t)__builtin_popcount(pend);
            const uint32_t incl = wave_scan_add(cnt);
            n_wait = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            uint32_t at = incl - cnt;
#pragma unroll
            for (uint32_t q = 0; q < 12; ++q) {
                if ((pend >> q) & 1u) wl[at] = (uint16_t)(own + q);
                at += (pend >> q) & 1u;
            }
        }
        __syncthreads();
        {
            const uint32_t spare = C::OUTC + wid;
            if (lane == 0) le[spare] = 0x8000u;
            const uint32_t passes = (n_wait + 63u) >> 6;
            const uint32_t padded = passes <= 2u ? 2u : passes <= 4u ? 4u : passes <= 6u ? 6u : passes <= 8u ? 8u : 12u;
            if (passes) for (uint32_t it = n_wait + lane; it < padded * 64u; it += 64u) wl[it] = (uint16_t)spare;
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
                    if (__ballot(open != 0u) == 0ull) break;
                }
            };
            if (passes == 0u) {}
            else if (passes <= 2u) jump(std::integral_constant<uint32_t, 2>{});
            else if (passes <= 4u) jump(std::integral_constant<uint32_t, 4>{});
            else if (passes <= 6u) jump(std::integral_constant<uint32_t, 6>{});
            else if (passes <= 8u) jump(std::integral_constant<uint32_t, 8>{});
            else jump(std::integral_constant<uint32_t, 12>{});
        }
        __syncthreads();
        if (live) {
            const uint2 e0 = *reinterpret_cast<const uint2*>(lds + DuoCopyLds::ent + 2u * own), e1 = *reinterpret_cast<const uint2*>(lds + DuoCopyLds::ent + 2u * own + 8u),
                        e2 = *reinterpret_cast<const uint2*>(lds + DuoCopyLds::ent + 2u * own + 16u);
            const uint32_t ev[6] = {e0.x, e0.y, e1.x, e1.y, e2.x, e2.y};
#pragma unroll
            for (uint32_t w = 0; w < 3; ++w) {
                uint32_t x = a0 + own + 4u * w; x -= x >= C::R ? C::R : 0u;
                const uint32_t lo = ev[2 * w], hi = ev[2 * w + 1];
                *reinterpret_cast<uint32_t*>(ring + x) = (lo & 0xFFu) | ((lo >> 8) & 0xFF00u) | ((hi & 0xFFu) << 16) | ((hi >> 16) << 24);
            }
        }
        prev_op = op; prev_len = image_len; prev_ring = ring_op;
        ring_op += image_len; ring_op -= ring_op >= C::R ? C::R : 0u;
        __syncthreads();                                                               // the image is complete in the ring
        if (last_chunk) break;
    }
    flush_image();
}

// NP PARSE workgroups and one COPY workgroup per block, all on one XCD: workgroup w -> (xcd = w % 8, slot = w / 8), role = slot % (NP + 1),
// block = (slot / (NP + 1)) * 8 + xcd; the COPY workgroup (role NP) is dispatched last.
#ifdef TSQ_STATS
// instrumented builds: the XCD (XCC_ID) every workgroup of the last launch really ran on (tools/duo_xcd.py: how often do a block's
// workgroups share one, as the index mapping below assumes?)
__device__ uint32_t g_duo_xcc[2048];
#endif
template <uint32_t NP>
__global__ __launch_bounds__(1024) void dec_duo_kernel(const uint8_t* __restrict__ container, const FrameInfo* __restrict__ frames, uint32_t n_blocks,
                                                       uint8_t* __restrict__ outbuf, int32_t* __restrict__ status,
                                                       uint32_t* __restrict__ ring_g, uint32_t* __restrict__ flags_g, uint32_t spin_limit)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t w = blockIdx.x, xcd = w & 7u, slot = w >> 3;
    const uint32_t role = slot % (NP + 1u), b = (slot / (NP + 1u)) * 8u + xcd;
#ifdef TSQ_STATS
    if (threadIdx.x == 0 && w < 2048u) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); g_duo_xcc[w] = 0x80000000u | (id & 15u); }
#endif
    if (b >= n_blocks) return;
    // An error reported before this launch (the frame walk refused the container: the descriptors are not even written) or by
    // another block: leave, all threads together.  (A block's workgroups may read different values while another block is failing;
    // every wait inside also watches the status word, so none is left waiting for a partner that left here.)
    {
        uint32_t* const seen = reinterpret_cast<uint32_t*>(lds);
        if (threadIdx.x == 0) seen[0] = (uint32_t)__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t st = seen[0];
        __syncthreads();
        if (st != 0u) return;
    }
    const FrameInfo f = frames[b];
    // (the frame descriptor may come from an untrusted container through tsqa_decode_blocks_async: bounds first)
    if (f.stream_len < 3u || f.stream_len > kSlotSize || f.out_len > kBlockSize) {
        if (threadIdx.x == 0) atomicMax(status, kErrStream);
        return;
    }
    uint32_t* const ring_b = ring_g + (size_t)b * DuoCfg::SLOTS * DuoCfg::REC_WORDS;
    uint32_t* const flags = flags_g + (size_t)b * DuoCfg::FLAG_STRIDE;
    if (role < NP) duo_parse<NP>(container + f.stream_at, f.stream_len, f.out_len, f.ext, role, ring_b, flags, status, lds, spin_limit);
    else duo_copy(container + f.stream_at, f.stream_len, f.out_len, f.ext, outbuf + f.out_at, ring_b, flags, status, lds, spin_limit);
}

}  // namespace tsq
