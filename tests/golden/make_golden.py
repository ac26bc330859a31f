#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/ (run in the BUILD container only, where
/root/reference exists).  Fixtures are DATA: inputs and the byte streams the compiled
reference (oracle/_ref/libtsq_ref.so, see oracle/Makefile) produces for them under the
canonical conditions of SURVEY.md 8c.  Nothing here travels as reference source.

  k1_input.bin            the 699-byte input string of the reference's tests (test/test.cpp:26)
  <name>.in / .noext / .ext   small inputs and the reference's output streams
  manifest.json           sizes + FNV-1a64 of every fixture and of the larger known-answer
                          vectors (K3, K6) that are too big to commit
"""
import ast
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

from oracle.pyoracle import Oracle, Reference, build  # noqa: E402
import kat  # noqa: E402


def k1_from_reference_test() -> bytes:
    src = open("/root/reference/test/test.cpp", "r", encoding="utf-8").read()
    m = re.search(r'const char \*testinput = ("(?:[^"\\]|\\.)*");', src)
    return ast.literal_eval("b" + m.group(1))


def main() -> None:
    build()
    ref, orc = Reference(), Oracle()
    k1 = k1_from_reference_test()
    assert len(k1) == 699
    open(os.path.join(HERE, "k1_input.bin"), "wb").write(k1)

    rng = np.random.default_rng(20260929)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 10)), dtype=np.uint8)) for _ in range(400)]
    wordy = b" ".join(words[int(i)] for i in rng.zipf(1.3, size=6000) % 400)[:24000]
    small = {
        "k0": bytes(kat.KATS["K0"][0]()),
        "k1": k1,
        "k1b": b"A",
        "len2": b"ab",
        "len5": b"aaaaa",
        "k2_zeros4096": bytes(4096),
        "k4_pattern": bytes(kat.k4_pattern(100000)[:20000]),
        "period4": (b"abcd" * 2000),
        "period5": (b"abcde" * 1000),
        "period64": bytes(range(64)) * 100,
        "wordy": wordy,
        "k7_head": bytes(kat.k7_textlike(30000)),
        "random4k": bytes(rng.integers(0, 256, size=4096, dtype=np.uint8)),
    }
    manifest = {}
    for name, data in small.items():
        open(os.path.join(HERE, name + ".in"), "wb").write(data)
        entry = {"in_bytes": len(data), "in_fnv": "%016x" % orc.fnv(data)}
        for ext, tag in ((0, "noext"), (1, "ext")):
            out = ref.encode_block(data, ext)
            assert ref.decode_block(out, ext) == data, name
            open(os.path.join(HERE, f"{name}.{tag}"), "wb").write(out)
            entry[tag] = {"bytes": len(out), "fnv": "%016x" % orc.fnv(out)}
        manifest[name] = entry
    # big known-answer vectors: hashes only
    for name in ("K3", "K5", "K6", "K7"):
        data = bytes(kat.KATS[name][0]())
        entry = {"in_bytes": len(data), "in_fnv": "%016x" % orc.fnv(data)}
        for ext, tag in ((0, "noext"), (1, "ext")):
            out = ref.encode_block(data, ext)
            entry[tag] = {"bytes": len(out), "fnv": "%016x" % orc.fnv(out)}
        manifest[name] = entry
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(small), "fixtures +", "manifest.json")


if __name__ == "__main__":
    main()
