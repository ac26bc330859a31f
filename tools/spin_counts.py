#!/usr/bin/env python3
"""Light instrumentation (make -C turbosqueeze_amd/csrc spins): unsuccessful polls per tile of every encoder wavefront of block 0, at
production timing (one ds_add per spin).  A tight poll is ~100 cycles, a sleeping one ~150-200.  Experiment tool.
  python tools/spin_counts.py [ext]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", os.environ.get("TSQ_LIB", "libturbosqueeze_amd_spins.so"))
api._libs.clear()
L = api.lib()
L.tsqa_debug_spins.argtypes = [C.c_void_p]
ext = int(sys.argv[1]) if len(sys.argv) > 1 else 0
codec = tsq.DeviceCodec(0)
src = torch.from_numpy(tsq.synth.text(10 ** 9, 1)).cuda()
out = torch.empty(api.container_bound(src.numel()), dtype=torch.uint8, device="cuda")
codec.profile(True)
codec.compress(src, ext, out); codec.profile_read()
for _ in range(3): codec.compress(src, ext, out)
torch.cuda.synchronize()
em, en, dm, dn = codec.profile_read()
sp = np.zeros(20, dtype=np.uint32)
assert L.tsqa_debug_spins(sp.ctypes.data) == 0
tiles = (1 << 22) / 64
import re
src_ = open(os.path.join(ROOT, "turbosqueeze_amd", "csrc", "tsq_enc_stage.cuh")).read()
mm = re.search(r"#define TSQ_X_MAP (\d+)", src_)
pat = r"constexpr uint32_t role_map\[16\] = \{([^}]*)\}"
names = [x.strip().replace("kRole", "") for x in re.search(pat, src_).group(1).split(",")]
extra = ["WALK waits for answers", "ACCOUNT waits for BUILDER (queue full)", "WALK waits for ACCOUNT (events full)", "BUILDER waits for EMIT (ring full)"]
print(f"encode kernel {em / max(en, 1):.2f} ms; unsuccessful polls per tile, block 0:")
for w in range(16):
    if names[w] != "None": print(f"  wave {w:2d} {names[w]:16s} {sp[w] / tiles:8.3f}")
for q in range(4): print(f"  {extra[q]:44s} {sp[16 + q] / tiles:8.3f}")
