#!/usr/bin/env python3
"""Does the encoder's position-table traffic cost time?  

The 256 KiB position table of a block is written and read at random 2-byte places; an XCD works on n_blocks / 8 blocks at
a time, so its live tables are n_blocks / 8 x 256 KiB against 4 MiB of L2: 30 tables (7.5 MB) at 239 blocks -- they spill to
the Infinity Cache and show up as FETCH_SIZE / WRITE_SIZE -- but 15 tables (3.75 MB) at 120 blocks and 7.5 at 60, which stay
in L2.  The per-block work is identical (one workgroup per block, blocks <= CUs), so if the encode kernel takes the same time
at 60, 120 and 239 blocks, the spill costs no time.  This prints the kernel time per launch for those block counts (same
text, same seed); run it under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` to see the traffic per launch next to it.
  python tools/traffic_experiment.py [ext]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from turbosqueeze_amd import api

ext = int(sys.argv[1]) if len(sys.argv) > 1 else 0
codec = tsq.DeviceCodec(0)
codec.profile(True)
B = 1 << 22
host = tsq.synth.text(239 * B, 1)
for nb in (30, 60, 120, 180, 239):
    src = torch.from_numpy(host[: nb * B]).cuda()
    out = torch.empty(api.container_bound(nb * B), dtype=torch.uint8, device="cuda")
    blob = codec.compress(src, ext, out)
    codec.profile_read()
    for _ in range(5):
        blob = codec.compress(src, ext, out)
    torch.cuda.synchronize()
    em, en, _, _ = codec.profile_read()
    ms = em / max(en, 1)
    print(json.dumps({"blocks": nb, "blocks_per_xcd": round(nb / 8, 1), "live_tables_per_xcd_MB": round(nb / 8 * 0.25, 2), "l2_per_xcd_MB": 4,
                      "ext": ext, "encode_kernel_ms": round(ms, 3), "ratio": round(blob.numel() / (nb * B), 4)}), flush=True)
