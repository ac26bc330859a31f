#!/bin/bash
# Collects the round's evidence on the GPU box: kernel-trace stats and (separate passes) HBM PMCs.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch_bench.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write_bench.log 2>&1
python - <<PY
import csv, glob, collections, json
out = {}
for name in ("fetch", "write"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for fn in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    out[name] = {k: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for k, v in agg.items() if "tsq" in k}
import sys
sys.path.insert(0, ".")
import turbosqueeze_amd
out["kernel_fingerprint"] = turbosqueeze_amd.source_fingerprint()
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv
