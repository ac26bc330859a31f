/*
 * tsq_oracle.c -- CPU restatement of turbosqueeze's per-block codec.
 *
 * TEST INFRASTRUCTURE ONLY (see tsq_oracle.h).  Written from the behavioural
 * specification in SURVEY.md section 8a; every function cites the reference
 * lines whose behaviour it restates.  Plain C11, no dependencies.
 */
#define _GNU_SOURCE               /* pthread barriers, clock_gettime, pthread_setaffinity_np, CPU_SET */
#include "tsq_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>

/* ---- unaligned little-endian loads (tsq_common.h:115-118, tsq_encode.cpp:74,126) ---- */
static inline uint32_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* platform.h:30-38: trailing zero count that yields 64 for 0. */
static inline uint32_t ctz64_or_64(uint64_t v) { return v ? (uint32_t)__builtin_ctzll(v) : 64u; }

uint32_t tsqo_bound(uint32_t n)
{
    /* every input byte its own literal symbol: n payload, n/8 controls, n/2 sizes */
    return 3u + n + (n >> 3) + (n >> 1) + 8u;
}

/* -------------------------------------------------------------------------
 * Encoder
 * ---------------------------------------------------------------------- */

/* Output cursor and symbol bookkeeping (tsq_encode.cpp:57-61). */
typedef struct {
    uint8_t *out;
    uint32_t j;        /* next free output byte */
    uint32_t ctl_at;   /* position of the control byte being filled */
    uint32_t sz_at;    /* position of the size byte being filled */
    uint32_t nsym;     /* symbols emitted so far */
    uint32_t origin;   /* input position at the start of the current pair (rep_last_i) */
    uint32_t hw;       /* one past the highest output byte written so far: bytes at or beyond it
                          are still "zero-filled" (canonical condition) without a memset */
} emitter;

/* Allocate the next output byte as a control/size byte.  If nothing has been stored there yet
 * it holds the zero of the zero-filled buffer; otherwise it keeps the literal spill. */
static inline uint32_t alloc_byte(emitter *e)
{
    uint32_t at = e->j++;
    if (at >= e->hw) { e->out[at] = 0; e->hw = at + 1; }
    return at;
}

/*
 * Account one symbol: shift its literal/match bit into the control byte and
 * its nibble into the size byte; open a new control byte every 8 symbols and
 * a new size byte every 2 symbols, control first (tsq_encode.cpp:93-95,
 * 113-115, 157-159).  `origin_if_pair_closes` becomes the pair origin when
 * this symbol completes a pair.
 */
static inline void account_symbol(emitter *e, uint32_t is_literal, uint32_t nibble,
                                  uint32_t origin_if_pair_closes)
{
    e->nsym++;
    e->out[e->ctl_at] = (uint8_t)((e->out[e->ctl_at] << 1) | is_literal);
    if ((e->nsym & 7u) == 0) e->ctl_at = alloc_byte(e);
    e->out[e->sz_at] = (uint8_t)((e->out[e->sz_at] << 4) | nibble);
    if ((e->nsym & 1u) == 0) { e->sz_at = alloc_byte(e); e->origin = origin_if_pair_closes; }
}

/*
 * Emit input[from, to) as literal symbols of at most 16 bytes.  Each chunk is
 * written with a full 16-byte store whatever its length (tsq_encode.cpp:88,108
 * via tsq_common.h:44-50); the spill is what later shows up in never-written
 * control/size bytes.
 */
static inline uint32_t emit_literals(emitter *e, const uint8_t *in, uint32_t from, uint32_t to)
{
    while (to - from > 0) {
        uint32_t len = to - from > 16 ? 16 : to - from;
        memcpy(e->out + e->j, in + from, 16);
        if (e->j + 16 > e->hw) e->hw = e->j + 16;
        from += len;
        e->j += len;
        account_symbol(e, 1u, len - 1u, from);
    }
    return from;
}

/*
 * Look up the candidate for position i and record i in the table
 * (tsq_encode.cpp:74-79, 162-167).  The table keeps the low 16 bits of the
 * most recent position per 17-bit hash; the candidate is the unique position
 * congruent to that value in [i-65536, i-1].
 */
static inline uint32_t probe_and_insert(uint16_t *table, const uint8_t *in, uint32_t i,
                                        uint32_t *word)
{
    uint32_t w = ld32(in + i);
    uint32_t h = (w ^ (w >> 12)) & (TSQO_HASH_ENTRIES - 1u);
    uint32_t lo = table[h];
    uint32_t pos = (i & 0xFFFF0000u) + lo;
    if (lo >= (i & 0xFFFFu)) pos -= 65536u;
    table[h] = (uint16_t)i;
    *word = w;
    return pos;
}

/* offset in [4, 0xFFFE] (tsq_encode.cpp:100,145: (offset-4) < 0xFFFB unsigned). */
static inline int offset_ok(uint32_t offset) { return (offset - 4u) < 0xFFFBu; }

/* Common prefix of in+a and in+b in bytes, capped at `cap` (16 no-ext:
 * tsq_encode.cpp:126-137; 64 ext: tsq_encode.cpp:276-290). */
static inline uint32_t common_prefix(const uint8_t *in, uint32_t a, uint32_t b, uint32_t cap)
{
    uint32_t k = ctz64_or_64(ld64(in + a) ^ ld64(in + b)) >> 3;
    if (k == 8) {
        uint32_t nb;
        do {
            a += 8; b += 8;
            nb = ctz64_or_64(ld64(in + a) ^ ld64(in + b)) >> 3;
            k += nb;
        } while (nb == 8 && k < cap);
    }
    return k;
}

/* Match length -> size nibble (tsq_encode.cpp:44-45). */
static inline uint32_t length_nibble(uint32_t k)
{
    if (k >= 64) return 2;
    if (k >= 48) return 1;
    if (k >= 32) return 0;
    if (k >= 17) return 15;
    return k - 1;      /* 4..16 -> 3..15 */
}

/* Bytes consumed by a match with that nibble (tsq_encode.cpp:154,307). */
static inline uint32_t nibble_span(uint32_t m) { return m < 3 ? (m + 2u) << 4 : m + 1u; }

uint32_t tsqo_encode_block(const uint8_t *in, uint32_t n, uint8_t *out,
                           uint32_t ext, uint16_t *table)
{
    const uint32_t cap = ext ? 64u : 16u;
    emitter e;
    uint32_t i = 0, pending, pos, word, offset;

    memset(table, 0, TSQO_HASH_ENTRIES * sizeof(uint16_t));  /* tsq_context.cpp:77-80 */
    /* canonical zero-filled output, realised lazily through emitter.hw */
    out[0] = (uint8_t)n; out[1] = (uint8_t)(n >> 8); out[2] = (uint8_t)(n >> 16);
    out[3] = 0; out[4] = 0;
    e.out = out; e.ctl_at = 3; e.sz_at = 4; e.j = 5; e.nsym = 0; e.origin = 0; e.hw = 5;

    do {
        pending = i;   /* first not-yet-emitted input byte (last_i) */

        /* scan for the next match start; position 0 is never probed (tsq_encode.cpp:70-100) */
        do {
            i++;
            pos = probe_and_insert(table, in, i, &word);
            offset = e.origin - pos;   /* taken before any forced flush below */
            if (i - pending > 31)
                pending = emit_literals(&e, in, pending, i);
        } while (i < n && !(word == ld32(in + pos) && offset_ok(offset)));

        pending = emit_literals(&e, in, pending, i);   /* tsq_encode.cpp:103-118 */
        if (!(i < n)) break;

        /* chain of back-to-back matches (tsq_encode.cpp:123-170) */
        do {
            uint32_t k = common_prefix(in, i, pos, cap);
            uint32_t room = e.origin - pos;
            uint32_t m;
            /* the source must end before the pair origin the decoder copies relative to */
            if (k > room) k = room - 1u;
            if (k < 4) break;
            offset = e.origin - pos;
            if (!offset_ok(offset)) break;

            m = length_nibble(k);
            out[e.j++] = (uint8_t)offset;
            out[e.j++] = (uint8_t)(offset >> 8);
            if (e.j > e.hw) e.hw = e.j;
            i += nibble_span(m);
            account_symbol(&e, 0u, m, i);

            pos = probe_and_insert(table, in, i, &word);
            offset = e.origin - pos;
        } while (i < n - 5u && word == ld32(in + pos) && offset_ok(offset));
    } while (i < n);

    /* pad the last group: literal control bits; an odd final nibble moves to the
     * high half exactly once (tsq_encode.cpp:176-186) */
    {
        int shifted = 0;
        while ((e.nsym & 7u) != 0) {
            out[e.ctl_at] = (uint8_t)((out[e.ctl_at] << 1) | 1u);
            if (!shifted && (e.nsym & 1u) != 0) { out[e.sz_at] = (uint8_t)(out[e.sz_at] << 4); shifted = 1; }
            e.nsym++;
        }
    }
    return e.j;
}

/* -------------------------------------------------------------------------
 * Decoder
 * ---------------------------------------------------------------------- */

uint32_t tsqo_decode_block(const uint8_t *in, uint32_t in_len, uint8_t *out,
                           uint32_t ext, int *status)
{
    uint32_t size, i = 3, j = 0;
    int st = 0;

    if (in_len < 3) { if (status) *status = 2; return 0; }
    size = (uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16);
    if (size > TSQO_BLOCK_SZ) { if (status) *status = 1; return 0; }   /* tsq_decode.cpp:53,146 */

    /* One group = control byte + up to four pairs; both symbols of a pair are
     * positioned relative to the output cursor at the pair start
     * (tsq_decode.cpp:62-88, 155-229).  Output is clamped at `size`, which is
     * what the reference's slack-and-truncate amounts to (tsq_decode.cpp:125). */
    while (j < size && !st) {
        uint32_t control, p;
        if (i >= in_len) { st = 2; break; }
        control = in[i++];
        for (p = 0; p < 4 && j < size && !st; p++) {
            uint32_t sizes, origin = j, s;
            if (i >= in_len) { st = 2; break; }
            sizes = in[i++];
            for (s = 0; s < 2 && j < size; s++) {
                uint32_t nib = s == 0 ? sizes >> 4 : sizes & 15u;
                uint32_t is_lit = (control >> (7u - (2u * p + s))) & 1u;
                uint32_t len, take;
                if (is_lit) {
                    len = nib + 1u;
                    take = len < size - j ? len : size - j;
                    if (i + take > in_len) { st = 2; break; }
                    memcpy(out + j, in + i, take);
                    i += len;
                } else {
                    uint32_t off, src;
                    if (i + 2u > in_len) { st = 2; break; }
                    off = ld16(in + i);
                    i += 2;
                    len = (ext && nib < 3u) ? (nib + 2u) << 4 : nib + 1u;   /* tsq_decode.cpp:174-191 */
                    if (off > origin) { st = 3; break; }
                    src = origin - off;
                    take = len < size - j ? len : size - j;
                    if (src + take > origin) { st = 4; break; }
                    memcpy(out + j, out + src, take);
                }
                j += take;
            }
        }
    }
    if (status) *status = st;
    return st ? 0 : size;
}

/* -------------------------------------------------------------------------
 * Container (turbosqueeze.cpp:48-147, tsq_threads.cpp:218-239,333-335,513-524)
 * ---------------------------------------------------------------------- */

size_t tsqo_compress_bound(size_t n)
{
    size_t nb = (n + TSQO_BLOCK_SZ - 1) / TSQO_BLOCK_SZ;
    return 16 + nb * (3 + (size_t)tsqo_bound(TSQO_BLOCK_SZ) + 16);
}

typedef struct {
    const uint8_t *in; size_t n; uint32_t ext;
    uint8_t **slots; uint32_t *sizes; size_t nb;
    int tid, nthreads;
} enc_job;

static void *enc_worker(void *arg)
{
    enc_job *w = (enc_job *)arg;
    uint16_t *table = (uint16_t *)malloc(TSQO_HASH_ENTRIES * sizeof(uint16_t));
    size_t b;
    for (b = (size_t)w->tid; b < w->nb; b += (size_t)w->nthreads) {   /* tsq_threads.cpp:71 */
        size_t at = b * TSQO_BLOCK_SZ;
        uint32_t len = (uint32_t)(w->n - at < TSQO_BLOCK_SZ ? w->n - at : TSQO_BLOCK_SZ);
        w->sizes[b] = tsqo_encode_block(w->in + at, len, w->slots[b], w->ext, table);
    }
    free(table);
    return NULL;
}

size_t tsqo_compress(const uint8_t *in, size_t n, uint8_t *out, uint32_t ext, int threads)
{
    size_t nb = (n + TSQO_BLOCK_SZ - 1) / TSQO_BLOCK_SZ, b, at;
    /* one arena, a slot per block (the reference mallocs TSQ_OUTPUT_SZ * n_blocks, tsq_threads.cpp:339) */
    const size_t stride = ((size_t)tsqo_bound(TSQO_BLOCK_SZ) + 16 + 4095) & ~(size_t)4095;
    /* kept across calls (grown on demand, never freed, not re-entrant): a timed second call
     * does not pay first-touch page faults again, which is the CPU's best case */
    static uint8_t *arena; static size_t arena_cap;
    if ((nb ? nb : 1) * stride > arena_cap) { free(arena); arena_cap = (nb ? nb : 1) * stride; arena = (uint8_t *)malloc(arena_cap); }
    uint8_t **slots = (uint8_t **)calloc(nb ? nb : 1, sizeof(*slots));
    uint32_t *sizes = (uint32_t *)calloc(nb ? nb : 1, sizeof(*sizes));
    enc_job jobs[256];
    pthread_t th[256];
    int t;
    uint64_t total = n;
    uint32_t nb32 = (uint32_t)nb;

    for (b = 0; b < nb; b++) slots[b] = arena + b * stride;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    for (t = 0; t < threads; t++) {
        jobs[t] = (enc_job){ in, n, ext, slots, sizes, nb, t, threads };
        if (threads == 1) enc_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, enc_worker, &jobs[t]);
    }
    if (threads > 1) for (t = 0; t < threads; t++) pthread_join(th[t], NULL);

    memcpy(out, "TSQ1", 4);
    memcpy(out + 4, &nb32, 4);
    memcpy(out + 8, &total, 8);
    at = 16;
    for (b = 0; b < nb; b++) {
        uint32_t frame = sizes[b] | (ext ? 0x800000u : 0u);   /* tsq_threads.cpp:218-219 */
        out[at] = (uint8_t)frame; out[at + 1] = (uint8_t)(frame >> 8); out[at + 2] = (uint8_t)(frame >> 16);
        memcpy(out + at + 3, slots[b], sizes[b]);
        at += 3 + sizes[b];
    }
    free(slots); free(sizes);
    return at;
}

size_t tsqo_decompressed_size(const uint8_t *in, size_t n)
{
    uint64_t total;
    if (n < 16 || memcmp(in, "TSQ1", 4) != 0) return (size_t)-1;
    memcpy(&total, in + 8, 8);
    return (size_t)total;
}

typedef struct {
    const uint8_t *in; const size_t *frame_at; const uint32_t *frame_len; const uint32_t *frame_ext;
    const size_t *out_at; uint8_t *out; size_t out_cap; size_t nb; int tid, nthreads; int bad;
} dec_job;

static void *dec_worker(void *arg)
{
    dec_job *w = (dec_job *)arg;
    size_t b;
    for (b = (size_t)w->tid; b < w->nb; b += (size_t)w->nthreads) {
        const uint8_t *s = w->in + w->frame_at[b];
        uint32_t usize;
        int st = 0;
        if (w->frame_len[b] < 3) { w->bad = 1; continue; }
        usize = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
        if (usize > TSQO_BLOCK_SZ || w->out_at[b] + usize > w->out_cap) { w->bad = 1; continue; }
        tsqo_decode_block(s, w->frame_len[b], w->out + w->out_at[b], w->frame_ext[b], &st);
        if (st) w->bad = 1;
    }
    return NULL;
}

size_t tsqo_decompress(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, int threads)
{
    uint32_t nb32;
    size_t nb, b, at = 16, oat = 0;
    size_t *frame_at, *out_at;
    uint32_t *frame_len, *frame_ext;
    dec_job jobs[256];
    pthread_t th[256];
    int t, bad = 0;

    if (n < 16 || memcmp(in, "TSQ1", 4) != 0) return (size_t)-1;
    memcpy(&nb32, in + 4, 4);
    nb = nb32;
    frame_at = (size_t *)calloc(nb ? nb : 1, sizeof(size_t));
    out_at = (size_t *)calloc(nb ? nb : 1, sizeof(size_t));
    frame_len = (uint32_t *)calloc(nb ? nb : 1, sizeof(uint32_t));
    frame_ext = (uint32_t *)calloc(nb ? nb : 1, sizeof(uint32_t));
    /* the frame walk is serial: block k starts at 16 + sum(3 + size_j) (tsq_threads.cpp:513-524) */
    for (b = 0; b < nb && !bad; b++) {
        uint32_t frame, len;
        if (at + 3 > n) { bad = 1; break; }
        frame = (uint32_t)in[at] | ((uint32_t)in[at + 1] << 8) | ((uint32_t)in[at + 2] << 16);
        len = frame & 0x7FFFFFu;
        if (len < 3 || len > TSQO_OUTPUT_SZ || at + 3 + len > n) { bad = 1; break; }
        frame_at[b] = at + 3; frame_len[b] = len; frame_ext[b] = frame >> 23;
        out_at[b] = oat;
        oat += (size_t)in[at + 3] | ((size_t)in[at + 4] << 8) | ((size_t)in[at + 5] << 16);
        at += 3 + len;
    }
    if (!bad) {
        if (threads < 1) threads = 1;
        if (threads > 256) threads = 256;
        for (t = 0; t < threads; t++) {
            jobs[t] = (dec_job){ in, frame_at, frame_len, frame_ext, out_at, out, out_cap, nb, t, threads, 0 };
            if (threads == 1) dec_worker(&jobs[t]);
            else pthread_create(&th[t], NULL, dec_worker, &jobs[t]);
        }
        for (t = 0; t < threads; t++) {
            if (threads > 1) pthread_join(th[t], NULL);
            bad |= jobs[t].bad;
        }
    }
    free(frame_at); free(out_at); free(frame_len); free(frame_ext);
    return bad ? (size_t)-1 : oat;
}

uint64_t tsqo_fnv1a64(const uint8_t *p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    size_t i;
    for (i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

/* -------------------------------------------------------------------------
 * CPU baseline driver for bench.py (SURVEY.md 8d protocol): encode then decode
 * of one buffer on a persistent pthread pool, block b -> thread b % T (the
 * reference's assignment, tsq_threads.cpp:71), wall clock per phase, one
 * untimed warm pass and `reps` timed ones (every time is returned: the caller
 * takes median and best).  `enc`/`dec` are the reference's own
 * tsqEncode/tsqDecode from oracle/_ref when supplied, else the port above.
 *
 *  - threads are pinned, one per allowed CPU in order (pin != 0);
 *  - every worker first-touches the slots, decode scratch and output pages
 *    of the blocks it owns, so the pages are local to its NUMA node;
 *  - shape 0 "idealised": workers write their results in place and that is
 *    all (decode of the reference's over-copying decoder goes through a
 *    private buffer and is copied by the worker itself);
 *    shape 1 "reference-shaped": as the reference's pipeline, ONE writer
 *    thread takes the finished blocks in block order and copies them to the
 *    final buffer -- frames into a contiguous container on encode
 *    (tsq_threads.cpp:226-239), blocks into the output on decode (:648) --
 *    while the workers run.
 * ---------------------------------------------------------------------- */
typedef void (*tsqo_enc_fn)(void *ctx, uint8_t *in, uint8_t *out, uint32_t *outsz, uint32_t insz, uint32_t ext);
typedef void (*tsqo_dec_fn)(uint8_t *in, uint8_t *out, uint32_t *outsz, uint32_t insz, uint32_t ext);

typedef struct {
    tsqo_enc_fn enc; tsqo_dec_fn dec;
    const uint8_t *in; size_t n; uint32_t ext; size_t nb;
    uint8_t *slots; size_t stride; uint32_t *sizes; uint8_t *back;
    uint8_t *container; uint8_t *stage;            /* shape 1: gathered frames; per-block decode staging */
    volatile int *done;                            /* shape 1: per-block completion flags */
    int nthreads; int phase;                       /* 0 encode, 1 decode, 2 quit, 3 first touch */
    int shape; int pin; int ncpu; int *cpus;
    pthread_barrier_t go, fin;
} bench_shared;
typedef struct { bench_shared *sh; int tid; } bench_arg;

static void pin_to(bench_shared *sh, int slot)
{
#ifdef __linux__
    if (sh->pin && sh->ncpu > 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(sh->cpus[slot % sh->ncpu], &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof set, &set);
    }
#else
    (void)sh; (void)slot;
#endif
}

static void *bench_worker(void *p)
{
    bench_arg *a = (bench_arg *)p;
    bench_shared *sh = a->sh;
    uint16_t *table;
    uint8_t *scratch;
    struct { uint16_t *refhash; } ctx;
    pin_to(sh, a->tid);
    table = (uint16_t *)aligned_alloc(128, TSQO_HASH_ENTRIES * sizeof(uint16_t));
    scratch = (uint8_t *)malloc(TSQO_BLOCK_SZ + 4096);
    ctx.refhash = table;
    memset(scratch, 0, TSQO_BLOCK_SZ + 4096);
    for (;;) {
        size_t b;
        pthread_barrier_wait(&sh->go);
        if (sh->phase == 2) break;
        for (b = (size_t)a->tid; b < sh->nb; b += (size_t)sh->nthreads) {
            size_t at = b * TSQO_BLOCK_SZ;
            uint32_t len = (uint32_t)(sh->n - at < TSQO_BLOCK_SZ ? sh->n - at : TSQO_BLOCK_SZ), sz = 0;
            uint8_t *slot = sh->slots + b * sh->stride;
            if (sh->phase == 3) {                   /* first touch by the owner; canonical zero-filled output */
                memset(slot, 0, sh->stride);
                memset(sh->back + at, 0, len);
                if (sh->stage) memset(sh->stage + b * (size_t)(TSQO_BLOCK_SZ + 4096), 0, TSQO_BLOCK_SZ + 4096);
            } else if (sh->phase == 0) {
                if (sh->enc) { memset(table, 0, TSQO_HASH_ENTRIES * sizeof(uint16_t)); sh->enc(&ctx, (uint8_t *)sh->in + at, slot, &sz, len, sh->ext); }
                else sz = tsqo_encode_block(sh->in + at, len, slot, sh->ext, table);
                sh->sizes[b] = sz;
                if (sh->shape) __atomic_store_n(&sh->done[b], 1, __ATOMIC_RELEASE);
            } else {
                uint8_t *dst = sh->shape ? sh->stage + b * (size_t)(TSQO_BLOCK_SZ + 4096) : (sh->dec ? scratch : sh->back + at);
                if (sh->dec) sh->dec(slot, dst, &sz, sh->sizes[b], sh->ext);
                else { int st; tsqo_decode_block(slot, sh->sizes[b], dst, sh->ext, &st); }
                if (sh->shape) __atomic_store_n(&sh->done[b], 1, __ATOMIC_RELEASE);
                else if (sh->dec) memcpy(sh->back + at, scratch, len);
            }
        }
        pthread_barrier_wait(&sh->fin);
    }
    free(table); free(scratch);
    return NULL;
}

/* shape 1: the single ordered writer (compression_write_worker / decompression_write_worker) */
static void *bench_writer(void *p)
{
    bench_shared *sh = (bench_shared *)p;
    pin_to(sh, sh->nthreads);
    for (;;) {
        size_t b, cur = 16;
        pthread_barrier_wait(&sh->go);
        if (sh->phase == 2) break;
        if (sh->phase == 0 || sh->phase == 1) {
            for (b = 0; b < sh->nb; b++) {
                size_t at = b * TSQO_BLOCK_SZ;
                uint32_t len = (uint32_t)(sh->n - at < TSQO_BLOCK_SZ ? sh->n - at : TSQO_BLOCK_SZ);
                while (!__atomic_load_n(&sh->done[b], __ATOMIC_ACQUIRE)) sched_yield();
                sh->done[b] = 0;
                if (sh->phase == 0) {
                    uint32_t frame = sh->sizes[b] | (sh->ext ? 0x800000u : 0u);
                    sh->container[cur] = (uint8_t)frame; sh->container[cur + 1] = (uint8_t)(frame >> 8); sh->container[cur + 2] = (uint8_t)(frame >> 16);
                    memcpy(sh->container + cur + 3, sh->slots + b * sh->stride, sh->sizes[b]);
                    cur += 3 + sh->sizes[b];
                } else memcpy(sh->back + at, sh->stage + b * (size_t)(TSQO_BLOCK_SZ + 4096), len);
            }
        }
        pthread_barrier_wait(&sh->fin);
    }
    return NULL;
}

static double wall_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

/* `in` needs TSQO_HALO readable bytes after in[n-1].  enc_seconds / dec_seconds: `reps` entries each.
 * Returns 0 when the round trip is exact. */
int tsqo_cpubench2(void *enc, void *dec, const uint8_t *in, size_t n, uint32_t ext, int threads, int reps, int shape, int pin,
                   double *enc_seconds, double *dec_seconds, uint64_t *compressed_bytes)
{
    bench_shared sh;
    pthread_t *th, wr;
    bench_arg *args;
    int t, r, bad, parties;
    size_t b;
    if (threads < 1) threads = 1;
    memset(&sh, 0, sizeof sh);
    sh.enc = (tsqo_enc_fn)enc; sh.dec = (tsqo_dec_fn)dec; sh.in = in; sh.n = n; sh.ext = ext;
    sh.shape = shape ? 1 : 0; sh.pin = pin;
    sh.nb = (n + TSQO_BLOCK_SZ - 1) / TSQO_BLOCK_SZ;
    sh.stride = ((size_t)tsqo_bound(TSQO_BLOCK_SZ) + 4096 + 4095) & ~(size_t)4095;
    sh.slots = (uint8_t *)malloc(sh.nb * sh.stride);            /* untouched: the owners fault their pages in */
    sh.sizes = (uint32_t *)calloc(sh.nb, sizeof(uint32_t));
    sh.back = (uint8_t *)malloc(n + 4096);
    if (sh.shape) {
        sh.container = (uint8_t *)malloc(16 + sh.nb * (3 + (size_t)tsqo_bound(TSQO_BLOCK_SZ)));
        sh.stage = (uint8_t *)malloc(sh.nb * (size_t)(TSQO_BLOCK_SZ + 4096));
        sh.done = (volatile int *)calloc(sh.nb, sizeof(int));
        if (sh.container) memset(sh.container, 0, 16 + sh.nb * (3 + (size_t)tsqo_bound(TSQO_BLOCK_SZ)));   /* the writer's pages, touched once */
    }
    sh.nthreads = threads;
#ifdef __linux__
    {
        cpu_set_t set;
        int c;
        sh.cpus = (int *)malloc(sizeof(int) * CPU_SETSIZE);
        if (sched_getaffinity(0, sizeof set, &set) == 0)
            for (c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &set)) sh.cpus[sh.ncpu++] = c;
    }
#endif
    parties = threads + 1 + sh.shape;
    pthread_barrier_init(&sh.go, NULL, (unsigned)parties);
    pthread_barrier_init(&sh.fin, NULL, (unsigned)parties);
    th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    args = (bench_arg *)malloc(sizeof(bench_arg) * (size_t)threads);
    for (t = 0; t < threads; t++) { args[t].sh = &sh; args[t].tid = t; pthread_create(&th[t], NULL, bench_worker, &args[t]); }
    if (sh.shape) pthread_create(&wr, NULL, bench_writer, &sh);
    sh.phase = 3;
    pthread_barrier_wait(&sh.go); pthread_barrier_wait(&sh.fin);
    for (r = 0; r <= reps; r++) {                 /* r == 0 is the warm pass */
        double t0, t1, t2;
        sh.phase = 0; t0 = wall_now();
        pthread_barrier_wait(&sh.go); pthread_barrier_wait(&sh.fin);
        t1 = wall_now(); sh.phase = 1;
        pthread_barrier_wait(&sh.go); pthread_barrier_wait(&sh.fin);
        t2 = wall_now();
        if (r > 0) { if (enc_seconds) enc_seconds[r - 1] = t1 - t0; if (dec_seconds) dec_seconds[r - 1] = t2 - t1; }
    }
    sh.phase = 2;
    pthread_barrier_wait(&sh.go);
    for (t = 0; t < threads; t++) pthread_join(th[t], NULL);
    if (sh.shape) pthread_join(wr, NULL);
    bad = memcmp(sh.back, in, n) != 0;
    if (compressed_bytes) { uint64_t c = 16; for (b = 0; b < sh.nb; b++) c += 3 + sh.sizes[b]; *compressed_bytes = c; }
    pthread_barrier_destroy(&sh.go); pthread_barrier_destroy(&sh.fin);
    free(th); free(args); free(sh.slots); free(sh.sizes); free(sh.back); free(sh.container); free(sh.stage); free((void *)sh.done); free(sh.cpus);
    return bad;
}

/* the first form of the driver: idealised shape, unpinned, best of `reps` */
int tsqo_cpubench(void *enc, void *dec, const uint8_t *in, size_t n, uint32_t ext, int threads, int reps,
                  double *enc_seconds, double *dec_seconds, uint64_t *compressed_bytes)
{
    double *te = (double *)malloc(sizeof(double) * (size_t)(reps > 0 ? reps : 1)), *td = (double *)malloc(sizeof(double) * (size_t)(reps > 0 ? reps : 1));
    double be = 1e30, bd = 1e30;
    int r, bad = tsqo_cpubench2(enc, dec, in, n, ext, threads, reps, 0, 0, te, td, compressed_bytes);
    for (r = 0; r < reps; r++) { if (te[r] < be) be = te[r]; if (td[r] < bd) bd = td[r]; }
    if (enc_seconds) *enc_seconds = be;
    if (dec_seconds) *dec_seconds = bd;
    free(te); free(td);
    return bad;
}
