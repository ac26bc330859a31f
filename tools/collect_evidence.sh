#!/bin/bash
# Everything the round's numbers come from, collected on the GPU box into gpurun_out/$TAG (copy what is to be judged into profiles/).
# Usage (on the GPU box, from the repo root): bash tools/collect_evidence.sh r02
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
(lscpu | head -25; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; nproc; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -12) > $OUT/gpu_box_host.txt 2>&1
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5) > $OUT/pytest_gpu.log
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
# (bench.py takes roofline.traffic from the newest PMC summary under profiles/ that carries the running kernels' fingerprint: the one just made)
cp $OUT/pmc_summary.json profiles/${TAG}_pmc_fetch_write.json
(timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -1) > $OUT/bench.json
python tools/phase_stats.py > $OUT/phase_stats.txt 2>&1
(timeout 300 python tools/tile_trace.py 2>&1 | tail -60) > $OUT/tile_trace.txt
python tools/traffic_experiment.py > $OUT/traffic_experiment.jsonl 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
 for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c -d $OUT/traffic_$c -o t --output-format csv -- python tools/traffic_experiment.py > /dev/null 2>&1; done
 python - <<PY
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for fn in glob.glob("$OUT/traffic_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "enc_stage_kernel" in r["Kernel_Name"]:
                rows.append((int(r["Grid_Size"]) // int(r["Workgroup_Size"]), float(r["Counter_Value"])))
    agg = collections.defaultdict(list)
    for nb, v in rows:
        agg[nb].append(v)
    out[c] = {str(nb): sum(v) / len(v) for nb, v in sorted(agg.items())}
json.dump(out, open("$OUT/traffic_experiment_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
) > $OUT/traffic_experiment_pmc.log 2>&1
timeout 300 python tools/quick_check.py > $OUT/quick_check.txt 2>&1
for n in 1073741824 4294967296; do timeout 300 python tools/config5_sweep.py $n 1 2>/dev/null; done > $OUT/config5.jsonl
timeout 900 python tools/config5_full.py > $OUT/config5_full_10gib.jsonl 2>/dev/null
# more blocks than CUs with a partial last round: 1.25 GiB + 4 KiB of text = 321 blocks (256 + 65), and one GPU's share of config 5 at N = 8:
# 320 blocks of the 50 % mix with extensions (256 + 64)
timeout 300 python tools/config5_sweep.py 1342181376 0 0 text 2>/dev/null >> $OUT/config5.jsonl
timeout 300 python tools/config5_sweep.py 1342177280 1 0 mix 2>/dev/null >> $OUT/config5.jsonl
timeout 300 python tools/config5_sweep.py 4294967296 0 0 text 2>/dev/null >> $OUT/config5.jsonl
for args in "--synthetic 1000000000 --reps 5 --no-ext" "--synthetic 1000000000 --reps 5" "--synthetic 4000000000 --reps 3"; do timeout 300 tools/tsq_cli b $args 2>/dev/null | tail -1; done > $OUT/cli_host_buffers.json
tools/micro/lds_unaligned > $OUT/lds_access_costs.txt 2>&1
# the N > 1 line on this 1-GPU box: `python bench.py --gpus N` typed plainly launches its N ranks itself; here they share GPU 0 over gloo (a dry run of the
# code path, not a scaling measurement: N processes time-slice one GPU and one PCIe link)
TSQ_BENCH_BACKEND=gloo TSQ_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_2ranks_1gpu.json
TSQ_BENCH_BACKEND=gloo TSQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --size 2000000000 --ext 1 --kind mix 2>/dev/null | tail -1 > $OUT/bench_8ranks_1gpu.json
(TSQ_AMD_DEBUG=1 timeout 200 tools/tsq_cli b --synthetic 1000000000 --reps 2 --no-ext 2>&1 | tail -45) > $OUT/cli_timeline.txt
mkdir -p gpurun_out/x; bash tools/bottleneck.sh run > $OUT/bottleneck.txt 2>&1
python tools/spin_counts.py > $OUT/spin_counts.txt 2>&1
bash tools/sq_counters.sh $TAG > gpurun_out/${TAG}_sq.log 2>&1
bash tools/sq_units.sh > $OUT/sq_units.txt 2>&1
timeout 200 python tools/duo_xcd.py 2>&1 | grep -v amdgpu.ids > $OUT/duo_xcd.txt
tools/micro/hostcopy > $OUT/hostcopy.txt 2>&1
ls -la $OUT
