#!/bin/bash
# Per-unit busy counters of the dominant kernels (VALU / scalar / LDS active cycles, LDS conflicts and misaligned accesses): separate
# rocprofv3 passes of two counters each, kernel-trace only.  Usage (GPU box, repo root): bash tools/sq_units.sh
OUT=gpurun_out/sq_units; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for pair in "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"; do
  name=$(echo $pair | tr ' ' '_')
  timeout 120 rocprofv3 --kernel-trace --pmc $pair -d $OUT/$name -o c --output-format csv -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extras > $OUT/$name.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for fn in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        if "enc_stage" in k or "dec_sym" in k:
            # keyed by kernel AND launch shape (blocks = grid size / workgroup size), like tools/pmc_summary.py: the profiled command also
            # launches both kernels at 1 024 blocks (the throughput job)
            try:
                blocks = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            except Exception:
                blocks = 0
            k = "%s@%d" % (k, blocks)
            a = agg[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-24s %.4g per launch (%d launches)" % (c, v / max(n, 1), n))
PY
