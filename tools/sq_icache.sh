#!/bin/bash
# Instruction-cache and instruction-fetch counters of the dominant kernels: separate rocprofv3 passes, kernel-trace only.
# Usage (GPU box, repo root): bash tools/sq_icache.sh
OUT=gpurun_out/sq_icache; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQC_INST" | head -40 > $OUT/avail.txt
for pair in "SQC_ICACHE_REQ SQC_ICACHE_HITS" "SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS"; do
  name=$(echo $pair | tr ' ' '_')
  timeout 120 rocprofv3 --kernel-trace --pmc $pair -d $OUT/$name -o c --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/$name.log 2>&1
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
        print("   %-28s %.4g per launch (%d launches)" % (c, v / max(n, 1), n))
PY
cat $OUT/avail.txt | head -30
