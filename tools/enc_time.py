#!/usr/bin/env python3
"""Timing only: encode kernel ms of every experiment library (tools/xbuild.sh) at the given sizes, no decode and no comparison
(for timing-only builds whose streams are wrong on purpose).  python tools/enc_time.py [bytes ...]   (default 1e9 and 4 GiB)"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import turbosqueeze_amd as tsq
    from turbosqueeze_amd import api
    lib = sys.argv[2]
    api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", lib)
    api._libs.clear()
    codec = tsq.DeviceCodec(0)
    out = {}
    for n in [int(float(x)) for x in sys.argv[3:]]:
        kind = os.environ.get("QC_KIND", "text")
        import numpy as np
        host = {"text": lambda: tsq.synth.text(n, 1), "zeros": lambda: np.zeros(n, dtype=np.uint8), "random": lambda: tsq.synth.random_bytes(n, 3), "mix": lambda: tsq.synth.mix(n, 3)}[kind]()
        src = torch.from_numpy(host).cuda()
        dst = torch.empty(api.container_bound(n), dtype=torch.uint8, device="cuda")
        ext = int(os.environ.get("QC_EXT", "0"))
        codec.compress_async(src, ext, dst); torch.cuda.synchronize()
        codec.profile(True)
        for _ in range(3):
            codec.compress_async(src, ext, dst)
        torch.cuda.synchronize()
        em, en, _, _ = codec.profile_read(); codec.profile(False)
        out[str(n)] = round(em / max(en, 1), 3)
        del src, dst
        torch.cuda.empty_cache()
    print(json.dumps(out))
    sys.exit(0)
sizes = sys.argv[1:] or ["1e9", str(4 << 30)]
for f in sorted(glob.glob(os.path.join(ROOT, "turbosqueeze_amd", "libturbosqueeze_amd_x_*.so"))):
    name = os.path.basename(f)[len("libturbosqueeze_amd_x_"):-3]
    try:      # (a timing-only build may hang: bounded, and the next library still runs)
        r = subprocess.run([sys.executable, __file__, "--one", os.path.basename(f)] + sizes, capture_output=True, text=True, timeout=int(os.environ.get("ENC_TIME_LIMIT", "120")))
    except subprocess.TimeoutExpired:
        print(name, "TIMED OUT", flush=True); continue
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    name = os.path.basename(f)[len("libturbosqueeze_amd_x_"):-3]
    if not line:
        print(name, "FAILED", r.stderr[-400:]); continue
    d = json.loads(line[-1])
    print(f"{name:16s} " + "  ".join(f"{int(float(k)) / 1e9:.2f} GB: {v:8.2f} ms = {int(float(k)) / v / 1e6:6.2f} GB/s" for k, v in d.items()), flush=True)
