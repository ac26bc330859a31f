#!/bin/bash
# SQ instruction counters of the two dominant kernels (separate rocprofv3 passes, two counters each; kernel-trace only).
# Usage (on the GPU box, from the repo root): bash tools/sq_counters.sh r02
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG/sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for pair in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pair | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $pair -d $OUT/$name -o c --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/$name.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if "enc_stage" in k or "dec_sym" in k:
            a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-24s %.4g per launch (%d launches)" % (c, v / max(n, 1), n))
PY
