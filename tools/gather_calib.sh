#!/bin/bash
# FETCH_SIZE calibration for the encoder's kind of access: bytes the counter reports per random 2-byte gather and per
# streamed byte.  On the GPU box: bash tools/gather_calib.sh [outdir]
OUT=${1:-gpurun_out/gather_calib}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- tools/micro/gather_calib > $OUT/run.log 2>&1
python - <<PY
import csv, glob
g = 4096 * 256 * 64
for fn in glob.glob("$OUT/f/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]; v = float(r["Counter_Value"])
        if "gather2" in k:
            print(f"gather2 : FETCH_SIZE = {v:.0f} KiB for {g} random 2-byte gathers -> {v * 1024 / g:.1f} B per gather as reported (x2 correction: {2 * v * 1024 / g:.1f} B)")
        if "stream16" in k:
            print(f"stream16: FETCH_SIZE = {v:.0f} KiB for {2 << 30} streamed bytes -> {v * 1024 / (2 << 30):.3f} B per byte as reported (the guide's x2 correction makes it {2 * v * 1024 / (2 << 30):.3f})")
PY
