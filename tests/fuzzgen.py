"""Seeded structured-random inputs that exercise literals, short/long matches, chains,
period-N runs, far offsets and block tails.  Shared by the oracle and GPU parity tests."""
from __future__ import annotations

import numpy as np


def structured(rng: np.random.Generator, n: int) -> np.ndarray:
    kind = int(rng.integers(0, 8))
    if kind == 0:      # uniform random
        return rng.integers(0, 256, size=n, dtype=np.uint8)
    if kind == 1:      # tiny alphabet: many hash hits, short matches
        k = int(rng.integers(1, 5))
        return rng.integers(97, 97 + k, size=n, dtype=np.uint8)
    if kind == 2:      # periodic with random period
        p = int(rng.integers(1, 300))
        base = rng.integers(0, 256, size=p, dtype=np.uint8)
        return np.resize(base, n)
    if kind == 3:      # copy-from-history model (LZ-friendly)
        out = np.empty(n, dtype=np.uint8)
        i = 0
        maxd = int(rng.choice([16, 256, 4096, 70000]))
        while i < n:
            if i > 8 and rng.random() < 0.7:
                d = int(rng.integers(1, min(i, maxd) + 1))
                ln = int(rng.integers(3, 80))
                for _ in range(min(ln, n - i)):
                    out[i] = out[i - d]
                    i += 1
            else:
                ln = int(rng.integers(1, 40))
                m = min(ln, n - i)
                out[i:i + m] = rng.integers(0, 256, size=m, dtype=np.uint8)
                i += m
        return out
    if kind == 4:      # word soup
        nw = int(rng.integers(4, 200))
        words = [rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8) for _ in range(nw)]
        parts, tot = [], 0
        while tot < n:
            w = words[int(rng.zipf(1.4)) % nw]
            parts.append(w); parts.append(np.array([32], dtype=np.uint8)); tot += w.size + 1
        return np.concatenate(parts)[:n]
    if kind == 5:      # zeros with sparse noise
        out = np.zeros(n, dtype=np.uint8)
        k = max(1, n // int(rng.integers(8, 200)))
        out[rng.integers(0, n, size=k)] = rng.integers(1, 256, size=k, dtype=np.uint8)
        return out
    if kind == 6:      # blocks of repeated random chunks (long matches, ext codes)
        chunk = rng.integers(0, 256, size=int(rng.integers(20, 400)), dtype=np.uint8)
        parts, tot = [], 0
        while tot < n:
            if rng.random() < 0.6:
                parts.append(chunk)
            else:
                parts.append(rng.integers(0, 256, size=int(rng.integers(1, 50)), dtype=np.uint8))
            tot += parts[-1].size
        return np.concatenate(parts)[:n]
    # kind 7: mix
    a = structured(rng, n // 2 + 1)
    b = structured(rng, n - n // 2 + 1)
    return np.concatenate([a, b])[:n]
