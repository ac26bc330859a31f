#!/bin/bash
# Collects the round's evidence on the GPU box: kernel-trace stats and (separate passes) HBM PMCs.
# Usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01
# The summaries are keyed by kernel AND launch shape (tools/pmc_summary.py): the profiled command launches the codec kernels at
# 239 blocks (the headline job) and at 1 024 blocks (the throughput job).
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
# (the extra bench sections -- ext1, config5, host_gather -- launch the same kernels at other sizes: left out of the profiled command)
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
PMC="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats_bench.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $PMC > $OUT/fetch_bench.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $PMC > $OUT/write_bench.log 2>&1
python tools/pmc_summary.py pmc $OUT/pmc_summary.json fetch=$OUT/fetch write=$OUT/write > /dev/null
python tools/pmc_summary.py trace $OUT/kernel_by_shape.json $OUT/stats
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv
