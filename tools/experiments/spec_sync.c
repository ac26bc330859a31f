/* spec_sync.c -- experiment (no product code): does a greedy parse of a turbosqueeze block that starts W bytes early from an
 * EMPTY position table fall in step with the true parse (same visited positions, same decisions, same symbol parity) before it
 * reaches the segment boundary b?  That is the precondition for encoding the segments of one 4 MiB block in parallel
 * (DESIGN.md section 7, item 2).  The parse below is the decision logic of tsq_encode.cpp:48-189 / 192-342 without emission.
 *   gcc -O2 -o spec_sync spec_sync.c ../../turbosqueeze_amd/csrc/tsq_synth.c -lm -fopenmp && ./spec_sync
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void tsq_synth_text(uint8_t *out, size_t n, uint64_t seed, double s);
void tsq_synth_mix(uint8_t *out, size_t n, uint64_t seed, double s);

#define BLOCK (1u << 22)
#define HASHN (1u << 17)
static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t prefix(const uint8_t *in, uint32_t a, uint32_t b, uint32_t cap)
{
    uint32_t k = 0;
    while (k < cap) { uint64_t x = ld64(in + a + k) ^ ld64(in + b + k); if (x) return k + (__builtin_ctzll(x) >> 3); k += 8; }
    return k;
}
static uint32_t nibble(uint32_t k) { return k >= 64 ? 2 : k >= 48 ? 1 : k >= 32 ? 0 : k >= 17 ? 15 : k - 1; }
static uint32_t span(uint32_t m) { return m < 3 ? (m + 2) << 4 : m + 1; }

/* state byte per position: 0 not visited; else 1 | parity << 1 | is_match_start << 2 | nibble << 3 */
static void parse(const uint8_t *in, uint32_t n, uint32_t from, uint32_t to, uint32_t ext, uint16_t *table, uint8_t *state)
{
    const uint32_t cap = ext ? 64 : 16;
    uint32_t i = from, pending, pos, word, offset, origin = from, nsym = 0;
    memset(table, 0, HASHN * 2);
#define PROBE() do { word = ld32(in + i); uint32_t h = (word ^ (word >> 12)) & (HASHN - 1), lo = table[h]; pos = (i & 0xFFFF0000u) + lo; if (lo >= (i & 0xFFFFu)) pos -= 65536u; table[h] = (uint16_t)i; } while (0)
#define SYMBOL(next_origin) do { nsym++; if (!(nsym & 1)) origin = (next_origin); } while (0)
    do {
        pending = i;
        do {
            i++;
            PROBE();
            if (i < to) state[i] = (uint8_t)(1 | ((nsym & 1) << 1));
            offset = origin - pos;
            if (i - pending > 31) while (pending < i) { uint32_t len = i - pending > 16 ? 16 : i - pending; pending += len; SYMBOL(pending); }
        } while (i < n && i < to && !(word == ld32(in + pos) && (offset - 4u) < 0xFFFBu));
        while (pending < i) { uint32_t len = i - pending > 16 ? 16 : i - pending; pending += len; SYMBOL(pending); }
        if (!(i < n) || !(i < to)) break;
        do {
            uint32_t k = prefix(in, i, pos, cap), room = origin - pos, m;
            if (k > room) k = room - 1;
            if (k < 4) break;
            offset = origin - pos;
            if (!((offset - 4u) < 0xFFFBu)) break;
            m = nibble(k);
            state[i] = (uint8_t)(state[i] | 4 | (m << 3));
            i += span(m);
            SYMBOL(i);
            PROBE();
            if (i < to) state[i] = (uint8_t)(1 | ((nsym & 1) << 1));
            offset = origin - pos;
        } while (i < n - 5u && i < to && word == ld32(in + pos) && (offset - 4u) < 0xFFFBu);
    } while (i < n && i < to);
}

int main(int argc, char **argv)
{
    const uint32_t n = BLOCK, seg = 512u << 10;
    uint8_t *in = malloc(n + 256), *truth = calloc(n + 256, 1), *spec = calloc(n + 256, 1);
    uint16_t *table = malloc(HASHN * 2);
    const char *kinds[] = {"text", "mix"};
    for (int kind = 0; kind < 2; ++kind) for (uint32_t ext = 0; ext < 2; ++ext) for (int seed = 1; seed <= 3; ++seed) {
        memset(in, 0, n + 256);
        if (kind == 0) tsq_synth_text(in, n, (uint64_t)seed, 1.12); else tsq_synth_mix(in, n, (uint64_t)seed, 1.12);
        memset(truth, 0, n); parse(in, n, 0, n, ext, table, truth);
        for (uint32_t W = 32u << 10; W <= 256u << 10; W *= 2) {
            int ok = 0, okv = 0, total = 0; uint32_t worst = 0, nflips = 0; double mean = 0, meanv = 0;
            for (uint32_t b = seg; b < n; b += seg) {
                uint32_t s = b - W < b ? b - W : 0, last_bad = s;
                if (W > b) continue;
                memset(spec + s, 0, b + seg - s);
                parse(in, n, s, b + seg, ext, table, spec);
                uint32_t last_vis = s, flips = 0, prev_rel = 2;
                for (uint32_t q = s + 1; q < b + seg; ++q) {
                    if (spec[q] != truth[q]) last_bad = q;
                    if ((spec[q] & ~2u) != (truth[q] & ~2u)) last_vis = q;            /* visited set and decisions, parity aside */
                    else if (spec[q]) { uint32_t rel = ((spec[q] ^ truth[q]) >> 1) & 1u; if (prev_rel != 2 && rel != prev_rel) flips++; prev_rel = rel; }
                }
                total++; ok += last_bad < b; okv += last_vis < b; mean += (double)(last_bad - s); meanv += (double)(last_vis - s); nflips += flips;
                if (last_bad - s > worst) worst = last_bad - s;
            }
            printf("%-4s ext=%u seed=%d  warm-up %3u KiB: fully in step before b: %d of %d; visited set + decisions (parity aside) in step before b: %d of %d "
                   "(last such disagreement after %.0f bytes on average); parity relation changed %.1f times per segment\n",
                   kinds[kind], ext, seed, W >> 10, ok, total, okv, total, meanv / total, (double)nflips / total);
        }
    }
    return 0;
}
