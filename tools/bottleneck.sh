#!/bin/bash
# Bottleneck finder for the staged encoder.  One library per stage in which THAT stage sleeps 128 cycles per tile (per item / per
# batch for BUILDER / EMIT); the share of the delay that shows in the kernel time says how much of the time the stage is on the
# critical path.  Usage: tools/bottleneck.sh build   (build container)      tools/bottleneck.sh run   (GPU box)
set -u
cd "$(dirname "$0")/.."
NAMES=(hash twins near match commit orbit walk account builder emit)
if [ "${1:-}" = build ]; then
  args=(s_none "")
  for k in 0 1 2 3 4; do args+=("s${k}_${NAMES[$k]}" "-DTSQ_X_DELAY_STAGE=$k"); done
  tools/xbuild.sh "${args[@]}" > /dev/null
  args=()
  for k in 5 6 7 8 9; do args+=("s${k}_${NAMES[$k]}" "-DTSQ_X_DELAY_STAGE=$k"); done
  tools/xbuild.sh "${args[@]}" > /dev/null
  ls turbosqueeze_amd/libturbosqueeze_amd_x_s*.so
else
  bash tools/xrun.sh --enc-only > gpurun_out/x/bottleneck_raw.txt 2>&1
  python - <<'PY'
import re
rows = {}
for line in open("gpurun_out/x/bottleneck_raw.txt"):
    m = re.match(r"== (s\w+):.*PARITY (\w+).*ext=\d: encode kernel ([\d.]+) ms", line)
    if m: rows[m.group(1)] = (float(m.group(3)), m.group(2))
base = rows["s_none"][0]
tiles = 65537
cyc = base * 1e-3 * 2.4e9 / tiles
full = 128.0 / cyc * base
print(f"encode kernel without delay {base:.2f} ms = {cyc:.0f} cycles per tile at 2.4 GHz; 128 cycles per tile on the critical path would add {full:.2f} ms")
print("stage      kernel ms   added ms   share of the delay that shows")
for k, (ms, par) in sorted(rows.items()):
    if k == "s_none": continue
    note = "  (two wavefronts: every other tile)" if ("match" in k or "orbit" in k) else ("  (per item)" if "builder" in k else "  (per batch of 64 symbols)" if "emit" in k else "")
    print(f"{k[3:]:10s} {ms:8.2f}   {ms - base:+7.2f}    {(ms - base) / full:5.2f}{note}   parity {par}")
PY
fi
