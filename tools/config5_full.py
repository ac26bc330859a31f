#!/usr/bin/env python3
"""BASELINE.json config 5 at its full size on ONE GPU: 10 GiB of zeros / random / 50 % mix, with extensions.  For each input: compress on
the device, compare the WHOLE container with the oracle's, decompress, compare with the input; ratio and GB/s (kernel time from the
library's HIP events).  (The 8-GPU form of this config shards these 2 560 blocks b % 8; one GPU simply queues them: 10 rounds of blocks.)"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turbosqueeze_amd as tsq
from oracle.pyoracle import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10 << 30
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ("zeros", "random", "mix")
lim = open("/sys/fs/cgroup/memory.max").read().strip() if os.path.exists("/sys/fs/cgroup/memory.max") else "max"
if lim != "max" and int(lim) < 6 * n:
    sys.exit(f"host memory limit {lim} is too small for a {n}-byte run")
codec, orc = tsq.DeviceCodec(0), Oracle()
nb = (n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ
for kind in kinds:
    host = {"zeros": lambda: np.zeros(n, dtype=np.uint8), "random": lambda: tsq.synth.random_bytes(n, 3),
            "mix": lambda: tsq.synth.mix(n, 3), "text": lambda: tsq.synth.text(n, 3)}[kind]()
    src = torch.from_numpy(host).cuda()
    out = torch.empty(tsq.container_bound(n), dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    codec.compress_async(src, 1, out); codec.decompress_async(out, nb, back); torch.cuda.synchronize()
    codec.profile(True)
    codec.compress_async(src, 1, out); torch.cuda.synchronize()
    csz, st = codec.last_size_status(); assert st == 0
    codec.decompress_async(out, nb, back); torch.cuda.synchronize()
    em, en, dm, dn = codec.profile_read(); codec.profile(False)
    round_trip = bool(torch.equal(back, src))
    del back
    t0 = time.perf_counter()
    want = np.frombuffer(orc.compress(host, 1, threads=32), dtype=np.uint8)
    t_oracle = time.perf_counter() - t0
    got = out[:csz].cpu().numpy()
    same = got.size == want.size and bool(np.array_equal(got, want))
    print(json.dumps({"input": kind, "bytes": n, "blocks": nb, "ext": 1, "ratio": round(csz / n, 4), "container_equals_oracle": same, "round_trip_exact": round_trip,
                      "encode_GBps": round(n / (em / en * 1e-3) / 1e9, 2), "decode_GBps": round(n / (dm / dn * 1e-3) / 1e9, 2),
                      "oracle_compress_s_on_host": round(t_oracle, 1)}), flush=True)
    assert same and round_trip
    del src, out, host, want, got
    torch.cuda.empty_cache()
