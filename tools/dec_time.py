#!/usr/bin/env python3
"""Timing only: decode kernel ms of every experiment library (tools/xbuild.sh) at 239 and 1 024 blocks; the output is compared and the
verdict printed, but a wrong output does not stop the run (timing-only builds)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    import turbosqueeze_amd as tsq
    from turbosqueeze_amd import api
    base = tsq.DeviceCodec(0)                      # the container is made by the product library
    out = {}
    blobs = {}
    for n in [int(float(x)) for x in sys.argv[3:]]:
        src = torch.from_numpy(tsq.synth.text(n, 1)).cuda()
        blobs[n] = (base.compress(src, 0).clone(), src)
    base.close()
    api.lib_path = lambda ab=False: os.path.join(ROOT, "turbosqueeze_amd", sys.argv[2])
    api._libs.clear()
    codec = tsq.DeviceCodec(0)
    for n, (blob, src) in blobs.items():
        nb = (n + tsq.BLOCK_SZ - 1) // tsq.BLOCK_SZ
        back = torch.empty(n, dtype=torch.uint8, device="cuda")
        codec.decompress_async(blob, nb, back); torch.cuda.synchronize()
        codec.profile(True)
        for _ in range(5):
            codec.decompress_async(blob, nb, back)
        torch.cuda.synchronize()
        _, _, dm, dn = codec.profile_read(); codec.profile(False)
        out[str(n)] = [round(dm / max(dn, 1), 3), bool(torch.equal(back, src))]
    print(json.dumps(out))
    sys.exit(0)
sizes = sys.argv[1:] or ["1e9", str(4 << 30)]
for f in sorted(glob.glob(os.path.join(ROOT, "turbosqueeze_amd", "libturbosqueeze_amd_x_*.so"))):
    r = subprocess.run([sys.executable, __file__, "--one", os.path.basename(f)] + sizes, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    name = os.path.basename(f)[len("libturbosqueeze_amd_x_"):-3]
    if not line:
        print(name, "FAILED", r.stderr[-600:]); continue
    d = json.loads(line[-1])
    print(f"{name:16s} " + "  ".join(f"{int(float(k)) / 1e9:.2f} GB: {v[0]:7.3f} ms = {int(float(k)) / v[0] / 1e6:6.1f} GB/s ({'ok' if v[1] else 'WRONG OUTPUT'})" for k, v in d.items()), flush=True)
