#!/usr/bin/env python3
"""Times the regions of the staged encoder's parser wave one at a time (make -C turbosqueeze_amd/csrc regions):
each library times ONE region with two s_memtime per pass; region 0 is empty and calibrates the timer cost.
Experiment tool, not product."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "(empty: timer cost)", 1: "W wait+loads+prologue", 2: "W orbit+check", 3: "W masks, last match, tile-end test", 4: "W hazard: candidate (twins, LDS words)", 5: "W hazard: decide + event (+ answer)", 6: "W tile end: SEG event", 7: "W visited mask + publish",
         10: "M wait scan + record loads", 11: "M wait parser + commit", 12: "M table gather", 13: "M candidate bytes + prefix", 14: "M classify + write + publish"}

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    k = int(sys.argv[2])
    sys.path.insert(0, ROOT)
    import torch
    import turbosqueeze_amd as tsq
    from turbosqueeze_amd import api
    api.lib_path = lambda ab=False, k=k: os.path.join(ROOT, "turbosqueeze_amd", f"libturbosqueeze_amd_stats_r{k}.so")
    api._libs.clear()
    L = api.lib()
    L.tsqa_debug_stats.argtypes = [C.c_void_p, C.c_void_p]
    n = 64 * (1 << 22)
    kind = sys.argv[3] if len(sys.argv) > 3 else "text"
    ext = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    host = {"text": tsq.synth.text, "random": tsq.synth.random_bytes, "mix": tsq.synth.mix}[kind](n, 1)
    src = torch.from_numpy(host).cuda()
    codec = tsq.DeviceCodec(0)
    blob = codec.compress(src, ext)
    assert torch.equal(codec.decompress(blob), src)
    enc = (C.c_ulonglong * 48)()
    L.tsqa_debug_stats(enc, None)
    e = list(enc)
    T = max(e[15], 1)
    a, b = (13, 14) if 10 <= k <= 14 else (11, 12)
    print(f"{k} {e[a] / T:.1f} {e[b] / T:.3f} {(e[4] if 10 <= k <= 14 else e[10]) / T:.1f}")
    sys.exit(0)

rows = []
for k in [int(x) for x in os.environ.get('TSQ_REGIONS', '0 1 2 3 4 5 6 7 10 11 12 13 14').split()]:
    out = subprocess.run([sys.executable, __file__, "--one", str(k)] + sys.argv[1:], capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l and l[0].isdigit()]
    if not line:
        print(out.stdout[-500:], out.stderr[-1500:]); sys.exit(1)
    rows.append([float(x) for x in line[-1].split()])
timer = rows[0][1] / max(rows[0][2], 1e-9)
print(f"timer cost per pass: {timer:.0f} cycles")
for r in rows[1:]:
    k, cyc, passes, total = int(r[0]), r[1], r[2], r[3]
    print(f"  {NAMES[k]:30s} {cyc - passes * timer:7.0f} cycles/tile  ({passes:.2f} passes/tile, {(cyc / max(passes, 1e-9)) - timer:6.0f} per pass; wave total in this build {total:.0f})")
