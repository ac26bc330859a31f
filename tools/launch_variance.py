#!/usr/bin/env python3
"""Launch-to-launch spread of the encode kernel on the headline job (60 launches per level, the library's HIP events): a delayed-stage
build took 10 ms longer once in round 3 and once in round 4 without reproducing; the production build does not show it
(41.77 .. 41.98 ms without extensions, 44.25 .. 45.57 with).  Experiment tool.  Usage (GPU box, repo root): python tools/launch_variance.py"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api
codec = tsq.DeviceCodec(0)
src = torch.from_numpy(tsq.synth.text(10**9, 1)).cuda()
out = torch.empty(api.container_bound(src.numel()), dtype=torch.uint8, device="cuda")
for ext in (0, 1):
    codec.compress(src, ext, out)
    ts = []
    for i in range(60):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        codec.profile(True); codec.profile_read()
        blob = codec.compress(src, ext, out)
        torch.cuda.synchronize()
        em, en, dm, dn = codec.profile_read()
        ts.append(em / max(en, 1))
    ts = np.array(ts)
    print(f"ext={ext}: encode kernel over 60 launches: min {ts.min():.2f} median {np.median(ts):.2f} mean {ts.mean():.2f} max {ts.max():.2f} ms; >1.05x median: {(ts > 1.05*np.median(ts)).sum()}")
