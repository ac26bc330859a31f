// twin_hypothesis.c -- experiment (not product, not oracle): how often would the encoder's WALK stage hit a hazard under two
// classification rules for lanes that have a same-hash position ("twin") in the lag window (their own tile and the two before)?
//
//   H0 (what enc_stage_kernel does): a lane is classified with the candidate gathered from the lagging table; a visited lane
//       with ANY visited twin in the window is a hazard.
//   H1: a lane with twins is classified ahead of time as if its NEAREST twin were visited (candidate = nearest twin); a visited
//       lane whose nearest twin turns out NOT visited is a hazard.
//
// The parse is the reference's greedy parse (tsq_encode.cpp:70-170, no-ext), restated only as far as the visited set needs.
//   gcc -O2 -o /tmp/twin_hypothesis tools/experiments/twin_hypothesis.c turbosqueeze_amd/csrc/tsq_synth.c -lm -fopenmp
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void tsq_synth_text(uint8_t *out, size_t n, uint64_t seed, double s);

static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint32_t hash4(uint32_t c) { return (c ^ (c >> 12)) & 0x1FFFFu; }
static uint32_t prefix(const uint8_t *a, const uint8_t *b, uint32_t max) { uint32_t k = 0; while (k < max && a[k] == b[k]) ++k; return k; }

int main(int argc, char **argv)
{
    const uint32_t n = 1u << 22;
    uint8_t *in = calloc(n + 256, 1);
    tsq_synth_text(in, n, argc > 1 ? strtoull(argv[1], 0, 0) : 1, 0.0);
    uint16_t *table = calloc(1u << 17, 2);
    uint8_t *visited = calloc(n + 256, 1);
    uint32_t *hs = malloc((n + 256) * 4);
    for (uint32_t p = 0; p < n + 200; ++p) hs[p] = hash4(ld32(in + p));

    // the greedy parse (symbol accounting reduced to the pair origin rule)
    uint32_t i = 0, last_i = 0, nsym = 0, origin = 0;
    for (;;) {
        uint32_t pos = 0, cur = 0;
        int hit = 0;
        do {
            ++i;
            cur = ld32(in + i);
            const uint32_t h = hs[i], p16 = table[h];
            pos = p16 + (i & 0xFFFF0000u) - (p16 >= (i & 0xFFFFu) ? 65536u : 0u);
            table[h] = (uint16_t)i;
            visited[i] = 1;
            if (i - last_i > 31) while (last_i != i) { last_i += 16; if (last_i > i) last_i = i; ++nsym; if (!(nsym & 1)) origin = last_i; }
            hit = cur == ld32(in + pos) && (uint32_t)(origin - pos - 4) < 0xFFFBu;
        } while (i < n && !hit);
        while (last_i < i) { uint32_t len = i - last_i > 16 ? 16 : i - last_i; last_i += len; ++nsym; if (!(nsym & 1)) origin = last_i; }
        if (i >= n) break;
        for (;;) {
            uint32_t k = prefix(in + i, in + pos, 16);
            if (k > origin - pos) k = origin - pos - 1;
            if (k < 4) break;
            if ((uint32_t)(origin - pos - 4) >= 0xFFFBu) break;
            i += k;
            ++nsym; if (!(nsym & 1)) origin = i;
            cur = ld32(in + i);
            const uint32_t h = hs[i], p16 = table[h];
            pos = p16 + (i & 0xFFFF0000u) - (p16 >= (i & 0xFFFFu) ? 65536u : 0u);
            table[h] = (uint16_t)i;
            visited[i] = 1;
            if (!(i < n - 5 && cur == ld32(in + pos) && (uint32_t)(origin - pos - 4) < 0xFFFBu)) break;
        }
        last_i = i;
    }

    // per visited lane: its twins in the window
    uint64_t tiles = n >> 6, n_vis = 0, with_twin = 0, h0 = 0, h1 = 0, h1_none = 0, h1_other = 0, h0_near_is_nearest = 0, both = 0;
    uint64_t h1b = 0, h0w1 = 0, h1w1 = 0, agree = 0, agree_nearest = 0, agree_far = 0, agree_prev = 0, agree_lit = 0;
    for (uint32_t p = 128; p < n; ++p) {
        if (!visited[p]) continue;
        ++n_vis;
        const uint32_t lo = (p & ~63u) - 128u;           // the window: tiles t-2 .. t
        int nearest = -1, nearest_vis = -1;
        for (uint32_t q = p - 1; q >= lo && q != 0xFFFFFFFFu; --q)
            if (hs[q] == hs[p]) { if (nearest < 0) nearest = (int)q; if (visited[q]) { nearest_vis = (int)q; break; } }
        int nearest1 = -1, nearest_vis1 = -1;            // the same with a window of tiles t-1 .. t
        for (uint32_t q = p - 1; q >= lo + 64u && q != 0xFFFFFFFFu; --q)
            if (hs[q] == hs[p]) { if (nearest1 < 0) nearest1 = (int)q; if (visited[q]) { nearest_vis1 = (int)q; break; } }
        if (nearest < 0) continue;
        ++with_twin;
        if (nearest_vis >= 0) { ++h0; if (nearest_vis == nearest) ++h0_near_is_nearest; }
        if (nearest_vis != nearest) { ++h1; if (nearest_vis < 0) ++h1_none; else ++h1_other; }
        if (nearest_vis >= 0 && nearest_vis != nearest) ++both;
        // H1b: the hypothesis only for lanes whose nearest twin starts a word-like context (previous byte differs in class)? -- a static
        // predictor: assume visited iff the twin's predecessor byte is a separator
        {
            const uint8_t c = in[nearest - 1];
            const int guess = c == ' ' || c == '\n' || c == '[' || c == ']' || c == '|' || c == '=' || c == '\'';
            const int truth = visited[nearest];
            if (guess != truth) ++h1b;
        }
        if (nearest_vis >= 0) {
            // would the orbit change at all?  span under the gathered candidate (most recent visited same-hash position before the
            // window) against the span under the visited twin
            uint32_t span0 = 1, span_t = 1;
            for (uint32_t q = lo - 1; q + 65536u > p && q != 0xFFFFFFFFu; --q)
                if (hs[q] == hs[p] && visited[q]) { const uint32_t k = prefix(in + p, in + q, 16); if (k >= 4) span0 = k; break; }
            if (p - (uint32_t)nearest_vis >= 4) { const uint32_t k = prefix(in + p, in + nearest_vis, 16); if (k >= 4) span_t = k; }
            if (span0 == span_t) { ++agree; if (nearest_vis == nearest) { ++agree_nearest; if (p - (uint32_t)nearest >= 64u) ++agree_far; if ((uint32_t)nearest < (p & ~63u)) ++agree_prev; if (span0 == 1) ++agree_lit; } }
        }
        if (nearest_vis1 >= 0) ++h0w1;
        if (nearest1 >= 0 && nearest_vis1 != nearest1) ++h1w1;
    }
    printf("tiles %llu  visited lanes per tile %.2f  visited lanes with a twin in the window per tile %.3f\n",
           (unsigned long long)tiles, (double)n_vis / tiles, (double)with_twin / tiles);
    printf("H0 hazards per tile (some twin visited)               %.3f   (of which the visited one is the nearest twin %.3f)\n", (double)h0 / tiles, (double)h0_near_is_nearest / tiles);
    printf("H1 hazards per tile (nearest twin not visited)        %.3f   (no twin visited %.3f, another twin visited %.3f)\n", (double)h1 / tiles, (double)h1_none / tiles, (double)h1_other / tiles);
    printf("static separator predictor wrong per tile             %.3f\n", (double)h1b / tiles);
    printf("H0 hazards whose span does not change with the twin  %.3f (and the twin is the nearest %.3f)\n", (double)agree / tiles, (double)agree_nearest / tiles);
    printf("   of those: twin at least 64 back %.3f, twin in an earlier tile %.3f, both literal %.3f\n", (double)agree_far / tiles, (double)agree_prev / tiles, (double)agree_lit / tiles);
    printf("window t-1..t only: H0 %.3f  H1 %.3f\n", (double)h0w1 / tiles, (double)h1w1 / tiles);

    // ORBIT's late classification (lanes whose only twins are in tile t-2): in how many tiles does it change any lane's span?
    {
        uint64_t tiles_with_only2 = 0, tiles_changed = 0, lanes_changed = 0, tiles_with_hit = 0;
        for (uint32_t t = 3; t < (n >> 6); ++t) {
            int any_only2 = 0, any_change = 0, any_hit = 0;
            for (uint32_t p = t << 6; p < (t << 6) + 64; ++p) {
                const uint32_t lo2 = (t << 6) - 128u, lo1 = (t << 6) - 64u;
                int nearer = 0, vis2 = -1, has2 = 0;
                for (uint32_t q = p - 1; q >= lo1; --q) if (hs[q] == hs[p]) { nearer = 1; break; }
                if (nearer) continue;
                for (uint32_t q = lo1 - 1; q >= lo2; --q) if (hs[q] == hs[p]) { has2 = 1; if (visited[q]) { vis2 = (int)q; break; } }
                if (!has2) continue;
                any_only2 = 1;
                if (vis2 < 0) continue;
                any_hit = 1;
                uint32_t span0 = 1, span_t = 1;
                for (uint32_t q = lo2 - 1; q + 65536u > p && q != 0xFFFFFFFFu; --q)
                    if (hs[q] == hs[p] && visited[q]) { const uint32_t k = prefix(in + p, in + q, 16); if (k >= 4) span0 = k; break; }
                { const uint32_t k = prefix(in + p, in + vis2, 16); if (k >= 4) span_t = k; }
                if (span0 != span_t) { any_change = 1; ++lanes_changed; }
            }
            tiles_with_only2 += any_only2; tiles_changed += any_change; tiles_with_hit += any_hit;
        }
        printf("late classification: tiles with a lane whose only twins are in t-2 %.3f, with a visited one %.3f, where a span changes %.3f (%.3f lanes per tile)\n",
               (double)tiles_with_only2 / tiles, (double)tiles_with_hit / tiles, (double)tiles_changed / tiles, (double)lanes_changed / tiles);
    }
    return 0;
}
