#!/usr/bin/env python3
"""Debug helper: parse two block streams into symbol lists and show the first difference."""
import sys


def parse(stream: bytes, ext: int):
    size = int.from_bytes(stream[:3], "little")
    i, j, syms = 3, 0, []
    while j < size and i < len(stream):
        ctl = stream[i]; i += 1
        for p in range(4):
            if j >= size or i >= len(stream):
                break
            sb = stream[i]; i += 1
            origin = j
            for s in range(2):
                if j >= size:
                    break
                nib = sb >> 4 if s == 0 else sb & 15
                lit = (ctl >> (7 - (2 * p + s))) & 1
                if lit:
                    ln = nib + 1
                    syms.append(("L", j, ln, bytes(stream[i:i + ln])))
                    i += ln
                else:
                    off = stream[i] | stream[i + 1] << 8
                    i += 2
                    ln = (nib + 2) << 4 if (ext and nib < 3) else nib + 1
                    syms.append(("M", j, ln, off, origin))
                j += ln
    return syms


def diff(a: bytes, b: bytes, ext: int, names=("got", "want")):
    sa, sb = parse(a, ext), parse(b, ext)
    for k, (x, y) in enumerate(zip(sa, sb)):
        if x != y:
            print(f"first differing symbol #{k}:")
            for q in range(max(0, k - 4), min(len(sa), len(sb), k + 3)):
                mark = "=>" if q == k else "  "
                print(f" {mark} {q}: {names[0]}={sa[q]}  {names[1]}={sb[q]}")
            return
    print(f"symbol lists agree on {min(len(sa), len(sb))} symbols; lengths {len(sa)} vs {len(sb)}; bytes {len(a)} vs {len(b)}")
    if a != b:
        n = next(q for q in range(min(len(a), len(b))) if a[q] != b[q]) if a[:min(len(a), len(b))] != b[:min(len(a), len(b))] else min(len(a), len(b))
        print("first differing byte", n, a[max(0, n - 4):n + 4].hex(), b[max(0, n - 4):n + 4].hex())


if __name__ == "__main__":
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import turbosqueeze_amd as tsq
    from oracle.pyoracle import Oracle
    name, ext = sys.argv[1], int(sys.argv[2])
    data = open(name, "rb").read()
    got = tsq.tsq_encode(data, ext)
    want = Oracle().encode_block(data, ext)
    print("equal" if got == want else "DIFFERENT", len(got), len(want))
    diff(got, want, ext)
