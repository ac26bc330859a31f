#!/usr/bin/env python3
"""When each encoder stage handed over each of 256 consecutive tiles of block 0 (cycles, relative to WALK's hand-over of the tile
three back), and the averages.  Experiment tool.
  python tools/tile_trace.py [rows] [ext]
Library: TSQ_TRACE_LIB (default libturbosqueeze_amd_trace.so: -DTSQ_TRACEONLY, the timeline alone at production timing -- `make trace`;
libturbosqueeze_amd_stats.so has the timeline too, but its counters slow every stage down)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
LIB = os.environ.get("TSQ_TRACE_LIB", "libturbosqueeze_amd_trace.so")
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", LIB)
api._libs.clear()
L = api.lib()
L.tsqa_debug_trace.argtypes = [C.c_void_p]
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ext = int(sys.argv[2]) if len(sys.argv) > 2 else 0
codec = tsq.DeviceCodec(0)
kind = os.environ.get("TSQ_TRACE_KIND", "text")          # text | zeros | random | mix
n_in = int(float(os.environ.get("TSQ_TRACE_BYTES", 10 ** 9 if kind == "text" else 1 << 30)))      # (more than 256 blocks: the lean layout, two blocks per CU)
src = torch.from_numpy({"text": lambda: tsq.synth.text(n_in, 1), "zeros": lambda: np.zeros(n_in, dtype=np.uint8), "random": lambda: tsq.synth.random_bytes(n_in, 3),
                        "mix": lambda: tsq.synth.mix(n_in, 3)}[kind]()).cuda()
out = torch.empty(api.container_bound(src.numel()), dtype=torch.uint8, device="cuda")
codec.compress(src, ext, out)
codec.profile(True)
codec.compress_async(src, ext, out)
torch.cuda.synchronize()
em, en, _, _ = codec.profile_read(); codec.profile(False)
print("%s, %s input, ext=%d: encode kernel %.2f ms" % (LIB, kind, ext, em / max(en, 1)))
tr = np.zeros(4096, dtype=np.uint32)
assert L.tsqa_debug_trace(tr.ctypes.data) == 0
tr = tr.reshape(16, 256).astype(np.int64)
if tr[15].any():
    print("HW_ID of waves 0..15 (simd = bits 4..5):", " ".join("%d:simd%d" % (w, (int(tr[15, w]) >> 4) & 3) for w in range(16)))
names = ["HASH go", "HASH pub", "TWINS pub", "MATCH go", "MATCH ver", "MATCH pub", "COMMIT pub", "ORBIT pub", "WALK go", "WALK pub", "gather", "ORBIT go", "COMMIT go", "ACCT pub"]
ix = {n: k for k, n in enumerate(names)}
W = tr[ix["WALK pub"]]
def rel(stage, t, ref_t):   # cycles from WALK's hand-over of tile ref_t to `stage` of tile t (mod 2^32)
    return int((tr[ix[stage], t] - W[ref_t] + (1 << 31)) % (1 << 32) - (1 << 31))
print("cycles of tile t's hand-overs relative to WALK pub of tile t-5 (the tile whose table entries t's gather needs); period = WALK pub(t) - WALK pub(t-1)")
cols = ["COMMIT go", "COMMIT pub", "HASH go", "HASH pub", "TWINS pub", "MATCH go", "gather", "MATCH ver", "MATCH pub", "ORBIT go", "ORBIT pub", "WALK go", "WALK pub", "ACCT pub"]
lag = {"COMMIT go": 5, "COMMIT pub": 5}
hdr = ["tile", "period"] + [c + ("(t-5)" if c in lag else "") for c in cols]
print(" ".join("%12s" % h for h in hdr))
acc = np.zeros(len(hdr) - 1); cnt = 0
rows_ = []
for t in range(8, 256):
    vals = [rel("WALK pub", t, t - 1)] + [rel(c, t - lag.get(c, 0), t - 5) for c in cols]
    rows_.append(vals)
    if t < 8 + rows:
        print("%12d " % t + " ".join("%12d" % v for v in vals))
a = np.array(rows_)
print("%12s " % "mean" + " ".join("%12d" % v for v in a.mean(axis=0)))
print("%12s " % "median" + " ".join("%12d" % v for v in np.median(a, axis=0)))
# the chain, link by link (medians): each link is the time from the left hand-over to the right one, for the same dependency
def link(a_, ta, b_, tb):
    d = np.array([rel(b_, t - tb, t - 5) - rel(a_, t - ta, t - 5) for t in range(8, 256)])
    return "%-34s median %6d  mean %6d" % ("%s(t-%d) -> %s(t-%d)" % (a_, ta, b_, tb), np.median(d), d.mean())
print("links of the lag loop (tile t's gather needs COMMIT(t-5)):")
for l in [("WALK pub", 5, "COMMIT go", 5), ("COMMIT go", 5, "COMMIT pub", 5), ("COMMIT pub", 5, "gather", 0), ("gather", 0, "MATCH ver", 0), ("MATCH ver", 0, "MATCH pub", 0),
          ("MATCH pub", 0, "ORBIT go", 0), ("ORBIT go", 0, "ORBIT pub", 0), ("ORBIT pub", 0, "WALK go", 0), ("WALK go", 0, "WALK pub", 0), ("WALK pub", 0, "ACCT pub", 0),
          ("WALK pub", 1, "WALK go", 0), ("HASH go", 0, "HASH pub", 0), ("HASH pub", 0, "TWINS pub", 0), ("TWINS pub", 0, "MATCH go", 0), ("MATCH go", 0, "gather", 0)]:
    print("  " + link(*l))
# distributions: the serial stage's tile, the period, and how long each consumer takes to pick up an input that is there
def col(name, back=0):
    return np.array([rel(name, t - back, t - 5) for t in range(8, 256)])
def pct(name, d):
    q = np.percentile(d, [10, 25, 50, 75, 90, 99])
    print("  %-46s mean %6d | p10 %6d p25 %6d p50 %6d p75 %6d p90 %6d p99 %6d" % ((name, d.mean()) + tuple(q)))
print("distributions:")
pct("period", np.array([rel("WALK pub", t, t - 1) for t in range(8, 256)]))
pct("WALK go -> WALK pub", col("WALK pub") - col("WALK go"))
pct("WALK: max(ORBIT pub(t), WALK pub(t-1)) -> go", col("WALK go") - np.maximum(col("ORBIT pub"), col("WALK pub", 1)))
pct("WALK waits for ORBIT: ORBIT pub(t) - WALK pub(t-1)", col("ORBIT pub") - col("WALK pub", 1))
pct("COMMIT: max(WALK pub(t), COMMIT pub(t-1)) -> go", col("COMMIT go") - np.maximum(col("WALK pub"), col("COMMIT pub", 1)))
pct("COMMIT behind WALK: COMMIT pub(t-1) - WALK pub(t)", col("COMMIT pub", 1) - col("WALK pub"))
pct("COMMIT go -> pub", col("COMMIT pub") - col("COMMIT go"))
pct("MATCH waits for COMMIT: COMMIT pub(t-5) - MATCH go(t)", col("COMMIT pub", 5) - col("MATCH go"))
pct("MATCH: max(COMMIT pub(t-5), MATCH go) -> gather", col("gather") - np.maximum(col("COMMIT pub", 5), col("MATCH go")))
pct("MATCH ver waits for WALK(t-4): WALK pub(t-4) - gather", col("WALK pub", 4) - col("gather"))
pct("MATCH: max(WALK pub(t-4), gather) -> ver", col("MATCH ver") - np.maximum(col("WALK pub", 4), col("gather")))
pct("MATCH ver -> pub", col("MATCH pub") - col("MATCH ver"))
pct("ORBIT: MATCH pub -> ORBIT go", col("ORBIT go") - col("MATCH pub"))
pct("ORBIT go -> pub", col("ORBIT pub") - col("ORBIT go"))
pct("ACCOUNT: max(WALK pub(t), ACCT pub(t-1)) -> pub", col("ACCT pub") - np.maximum(col("WALK pub"), col("ACCT pub", 1)))
pct("ACCOUNT behind WALK: ACCT pub(t) - WALK pub(t)", col("ACCT pub") - col("WALK pub"))
pct("HASH ahead: WALK go(t) - HASH pub(t)", col("WALK go") - col("HASH pub"))
