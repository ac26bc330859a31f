#!/bin/bash
# Copies what is to be judged from gpurun_out/$TAG (scratch, written on the GPU box by tools/collect_evidence.sh and
# tools/sq_counters.sh) into profiles/ (tracked).  Usage: bash tools/publish_profiles.sh r03
set -eu
TAG=${1:?tag}
S=gpurun_out/$TAG
cp $S/bench.json profiles/${TAG}_bench.json
cp $S/kernel_stats.csv profiles/${TAG}_rocprofv3_kernel_stats.csv
tail -1 $S/stats_bench.log > profiles/${TAG}_bench_under_rocprof.json
cp $S/pmc_summary.json profiles/${TAG}_pmc_fetch_write.json
cp $S/phase_stats.txt profiles/${TAG}_phase_stats.txt
cp $S/config5.jsonl profiles/${TAG}_config5.jsonl
cp $S/cli_host_buffers.json profiles/${TAG}_cli_host_buffers.json
cp $S/cli_timeline.txt profiles/${TAG}_cli_timeline.txt
cp $S/bench_2ranks_1gpu.json profiles/${TAG}_bench_2ranks_1gpu.json
cp $S/pytest_gpu.log profiles/${TAG}_pytest_gpu.log
cp $S/gpu_box_host.txt profiles/${TAG}_gpu_box_host.txt
cp $S/traffic_experiment.jsonl profiles/${TAG}_traffic_experiment.jsonl
cp $S/traffic_experiment_pmc.json profiles/${TAG}_traffic_experiment_pmc.json
cp $S/quick_check.txt profiles/${TAG}_quick_check.txt
cp $S/bottleneck.txt profiles/${TAG}_encoder_bottleneck.txt
cp $S/spin_counts.txt profiles/${TAG}_encoder_spin_counts.txt
[ -f gpurun_out/${TAG}_sq.log ] && grep -v amdgpu.ids gpurun_out/${TAG}_sq.log > profiles/${TAG}_sq_counters.txt
[ -f $S/regions.txt ] && cp $S/regions.txt profiles/${TAG}_walk_regions.txt
[ -f $S/sq_units.txt ] && grep -v amdgpu.ids $S/sq_units.txt > profiles/${TAG}_sq_units.txt
[ -f $S/kernel_by_shape.json ] && cp $S/kernel_by_shape.json profiles/${TAG}_kernel_by_shape.json
[ -f $S/config5_full_10gib.jsonl ] && cp $S/config5_full_10gib.jsonl profiles/${TAG}_config5_full_10gib.jsonl
[ -f $S/duo_xcd.txt ] && cp $S/duo_xcd.txt profiles/${TAG}_duo_xcd.txt
[ -f $S/hostcopy.txt ] && cp $S/hostcopy.txt profiles/${TAG}_hostcopy.txt
ls -la profiles/${TAG}_*
[ -f $S/tile_trace.txt ] && grep -v amdgpu.ids $S/tile_trace.txt > profiles/${TAG}_tile_trace.txt
[ -f $S/sq_icache.txt ] && grep -v amdgpu.ids $S/sq_icache.txt > profiles/${TAG}_sq_icache.txt
for f in tile_trace_zeros_ext1 tile_trace_lean_random_ext1 fuzz_variants; do [ -f $S/$f.txt ] && grep -v amdgpu.ids $S/$f.txt > profiles/${TAG}_$f.txt; done
true
cp $S/bench_8ranks_1gpu.json profiles/${TAG}_bench_8ranks_1gpu.json 2>/dev/null || true
