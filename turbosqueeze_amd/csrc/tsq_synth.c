/*
 * tsq_synth.c -- deterministic synthetic inputs for tests and bench.py (not part of the codec).
 *
 * enwik9 is not available offline, so the benchmark input is "enwik-shaped" text from a seeded
 * Zipf word model: 2^17 pseudo-words, ranks ~ Zipf(s), separated by spaces, with
 * wiki/XML markup tokens and a few numbers mixed in.  The exponent s is calibrated so that the
 * reference codec's no-ext ratio lands at 0.62 +- 0.01, the ratio README.md:93 reports for
 * enwik9.  Output is a pure function of (seed, s, n): it is generated in independent 1 MiB
 * chunks, so the thread count does not change the bytes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VOCAB_BITS 17
#define VOCAB (1u << VOCAB_BITS)
#define CHUNK (1u << 20)
#define MAXW 14

static inline uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double unit(uint64_t *s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }

typedef struct {
    uint8_t words[VOCAB][MAXW];
    uint8_t wlen[VOCAB];
    uint32_t alias[VOCAB];
    uint32_t cut[VOCAB];       /* probability threshold scaled to 2^32 */
    double s;
    uint64_t seed;
    int ready;
} model_t;

static model_t *g_model;

static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
static const double letter_w[26] = { 12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8,
                                     2.4, 2.4, 2.2, 2.0, 2.0, 1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07 };

static unsigned poisson(uint64_t *st, double lambda)
{
    double L = exp(-lambda), p = 1.0;
    unsigned k = 0;
    do { k++; p *= unit(st); } while (p > L);
    return k - 1;
}

static void build_model(model_t *m, uint64_t seed, double s)
{
    uint64_t st = seed ^ 0x5851F42D4C957F2Dull;
    double cum[26], tot = 0;
    unsigned r, i;
    for (i = 0; i < 26; i++) { tot += letter_w[i]; cum[i] = tot; }
    for (r = 0; r < VOCAB; r++) {
        /* frequent words are shorter: mean length grows with log rank */
        double lambda = 1.2 + 4.2 * (log2((double)r + 2.0) / (double)VOCAB_BITS);
        unsigned len = 1 + poisson(&st, lambda);
        if (len > MAXW) len = MAXW;
        m->wlen[r] = (uint8_t)len;
        for (i = 0; i < len; i++) {
            double u = unit(&st) * tot;
            unsigned c = 0;
            while (c < 25 && cum[c] < u) c++;
            m->words[r][i] = (uint8_t)letters[c];
        }
        if (r % 23 == 7) m->words[r][0] = (uint8_t)(m->words[r][0] - 32);   /* some capitalised words */
    }
    /* alias table for Zipf(s) over VOCAB ranks (Vose) */
    {
        double *p = (double *)malloc(sizeof(double) * VOCAB);
        uint32_t *small = (uint32_t *)malloc(sizeof(uint32_t) * VOCAB);
        uint32_t *large = (uint32_t *)malloc(sizeof(uint32_t) * VOCAB);
        unsigned ns = 0, nl = 0;
        double z = 0;
        for (r = 0; r < VOCAB; r++) { p[r] = pow((double)r + 1.0, -s); z += p[r]; }
        for (r = 0; r < VOCAB; r++) { p[r] = p[r] / z * VOCAB; if (p[r] < 1.0) small[ns++] = r; else large[nl++] = r; }
        while (ns && nl) {
            uint32_t a = small[--ns], b = large[--nl];
            m->cut[a] = (uint32_t)(p[a] * 4294967295.0);
            m->alias[a] = b;
            p[b] = p[b] + p[a] - 1.0;
            if (p[b] < 1.0) small[ns++] = b; else large[nl++] = b;
        }
        while (nl) { uint32_t b = large[--nl]; m->cut[b] = 0xFFFFFFFFu; m->alias[b] = b; }
        while (ns) { uint32_t a = small[--ns]; m->cut[a] = 0xFFFFFFFFu; m->alias[a] = a; }
        free(p); free(small); free(large);
    }
    m->s = s; m->seed = seed; m->ready = 1;
}

static const char *markup[] = { "[[", "]]", "&quot;", "&lt;", "&gt;", "\n", "==", "'''", "{{", "}}", "|", "\n\n",
                                "</text>\n    </revision>\n  </page>\n  <page>\n    <title>", "</title>\n    <id>",
                                "</id>\n    <revision>\n      <id>", "</id>\n      <timestamp>2006-0", "</timestamp>\n      <contributor>\n        <username>",
                                "</username>\n        <id>", "</id>\n      </contributor>\n      <text xml:space=\"preserve\">", "[[Category:", "* ", "# ", ": ", "&amp;" };
#define N_MARKUP (sizeof(markup) / sizeof(markup[0]))

static void fill_chunk(const model_t *m, uint8_t *out, size_t n, uint64_t seed, uint64_t chunk_index)
{
    uint64_t st = seed * 0xD1342543DE82EF95ull + chunk_index * 0x9E3779B97F4A7C15ull + 1;
    uint8_t tmp[160];
    size_t at = 0;
    while (at < n) {
        uint64_t r = splitmix(&st);
        unsigned kind = (unsigned)(r & 1023u);
        size_t len = 0;
        if (kind < 41) {                                   /* ~4 % markup */
            const char *t = markup[(r >> 10) % N_MARKUP];
            len = strlen(t);
            memcpy(tmp, t, len);
        } else if (kind < 51) {                            /* ~1 % numbers */
            unsigned digits = 1 + (unsigned)((r >> 10) % 6), d;
            uint64_t v = r >> 20;
            for (d = 0; d < digits; d++) { tmp[len++] = (uint8_t)('0' + v % 10); v /= 10; }
            tmp[len++] = ' ';
        } else {
            uint32_t slot = (uint32_t)(r >> 10) & (VOCAB - 1u);
            uint32_t u = (uint32_t)(r >> 32);
            uint32_t w = u <= m->cut[slot] ? slot : m->alias[slot];
            len = m->wlen[w];
            memcpy(tmp, m->words[w], len);
            if ((r >> 27 & 31u) == 0) tmp[len++] = (r >> 9 & 1) ? ',' : '.';
            tmp[len++] = ' ';
        }
        if (len > n - at) len = n - at;
        memcpy(out + at, tmp, len);
        at += len;
    }
}

/* enwik-shaped text.  s <= 0 selects the calibrated default. */
void tsq_synth_text(uint8_t *out, size_t n, uint64_t seed, double s)
{
    size_t chunks = (n + CHUNK - 1) / CHUNK;
    long c;
    if (s <= 0) s = 1.12;  /* calibrated: oracle no-ext ratio 0.62 */
    if (!g_model) g_model = (model_t *)calloc(1, sizeof(model_t));
    if (!g_model->ready || g_model->s != s || g_model->seed != 0x7453517Aull) build_model(g_model, 0x7453517Aull, s);
#pragma omp parallel for schedule(dynamic, 4)
    for (c = 0; c < (long)chunks; c++) {
        size_t at = (size_t)c * CHUNK;
        fill_chunk(g_model, out + at, n - at < CHUNK ? n - at : CHUNK, seed, (uint64_t)c);
    }
}

/* uniform random bytes */
void tsq_synth_random(uint8_t *out, size_t n, uint64_t seed)
{
    size_t chunks = (n + CHUNK - 1) / CHUNK;
    long c;
#pragma omp parallel for schedule(static)
    for (c = 0; c < (long)chunks; c++) {
        uint64_t st = seed * 0xA24BAED4963EE407ull + (uint64_t)c;
        size_t at = (size_t)c * CHUNK, len = n - at < CHUNK ? n - at : CHUNK, k;
        for (k = 0; k + 8 <= len; k += 8) { uint64_t v = splitmix(&st); memcpy(out + at + k, &v, 8); }
        if (k < len) { uint64_t v = splitmix(&st); memcpy(out + at + k, &v, len - k); }
    }
}

/* alternating 64 KiB pieces of random bytes and text ("50 % compressible mix", BASELINE.json config 5) */
void tsq_synth_mix(uint8_t *out, size_t n, uint64_t seed, double s)
{
    size_t piece = 65536, at, k = 0;
    tsq_synth_text(out, n, seed, s);
    for (at = 0; at < n; at += piece, k++) {
        if ((k & 1) == 0) tsq_synth_random(out + at, n - at < piece ? n - at : piece, seed + k);
    }
}

/* scalar xorshift32 stream of SURVEY.md 8c (K5/K6): b = x >> 24 after each step */
void tsq_synth_xorshift32(uint8_t *out, size_t n, uint32_t seed)
{
    uint32_t x = seed;
    size_t i;
    for (i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; out[i] = (uint8_t)(x >> 24); }
}
