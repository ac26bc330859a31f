#!/bin/bash
# On the GPU box: the host layer (scheduler thread, per-device feeders, drainer, sinks, prefault / allocator threads) under
# ThreadSanitizer and AddressSanitizer.  Needs `make -C turbosqueeze_amd/csrc sanitize` and `make -C oracle san_programs` (build
# container).  Runs, per sanitizer: the reference's ten test programs (test/test.cpp:334-362; test_tsq_queue_mt and the 1 000 chained
# async jobs of test_tsq_massive_async_mt among them), tsq_cli file -> file in three sink modes (collected in memory / mapped +
# fallocate / positional writes) with one GPU listed twice (two feeder threads), and the memory benchmark.
# Usage: tools/sanitize.sh [outdir]      Reports land in <outdir>/<sanitizer>_<what>.log; the summary counts reports.
OUT=${1:-gpurun_out/san}
mkdir -p $OUT
R=oracle/_ref
W=$(mktemp -d /tmp/tsqsan.XXXX)
python - <<PY
import sys; sys.path.insert(0, ".")
import turbosqueeze_amd as tsq
tsq.synth.text(9 * (1 << 22) + 12345, seed=7).tofile("$W/in.bin")
PY
export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4 report_signal_unsafe=0 suppressions=$PWD/tools/tsan.supp"
# (protect_shadow_gap=0: the HIP runtime maps fixed addresses; detect_leaks=0: LeakSanitizer's stop-the-world scan at exit never returns with the
#  HIP runtime's threads alive -- the first attempt of this script sat in it until its timeout)
export ASAN_OPTIONS="detect_leaks=0 halt_on_error=0 protect_shadow_gap=0"
for s in ${SANITIZERS:-tsan asan}; do
  for t in test_tsq_context test_tsq_compress test_tsq_context_mt test_tsq_compress_mt test_tsq_queue_mt test_tsq_context_mt2 \
           test_tsq_decompress_mt test_tsq_compress_async_mt test_tsq_decompress_async_mt test_tsq_massive_async_mt; do
    timeout ${SAN_TIMEOUT:-300} $R/ref_test_$s $t > $OUT/${s}_$t.log 2>&1; echo "exit $?" >> $OUT/${s}_$t.log
  done
  k=0
  for env in "" "TSQ_AMD_FILE_MAP_MIN=1 TSQ_AMD_FILE_INMEM_MAX=1 TSQ_AMD_FILE_BATCH_BLOCKS=2" "TSQ_AMD_FILE_MAP_MIN=1 TSQ_AMD_FILE_NO_MMAP=1 TSQ_AMD_FILE_INMEM_MAX=1 TSQ_AMD_FILE_BATCH_BLOCKS=2"; do
    k=$((k+1))
    ( env $env TSQ_AMD_DEVICES=0,0 timeout ${SAN_TIMEOUT:-300} $R/tsq_cli_$s c $W/in.bin $W/out$k.tsq; env $env TSQ_AMD_DEVICES=0,0 timeout ${SAN_TIMEOUT:-300} $R/tsq_cli_$s d $W/out$k.tsq $W/back$k.bin; cmp $W/in.bin $W/back$k.bin && echo "round trip ok" ) > $OUT/${s}_cli_files_$k.log 2>&1; echo "exit $?" >> $OUT/${s}_cli_files_$k.log
  done
  ( TSQ_AMD_DEVICES=0,0 timeout ${SAN_TIMEOUT:-300} $R/tsq_cli_$s b --synthetic 100000000 --reps 2 ) > $OUT/${s}_cli_bench.log 2>&1; echo "exit $?" >> $OUT/${s}_cli_bench.log
  # one device, 72 blocks, results above 64 MiB: the ramped batches, the striped first touch of the result and its wait-for-my-pages
  ( timeout ${SAN_TIMEOUT:-300} $R/tsq_cli_$s b --synthetic 300000000 --reps 1 ) > $OUT/${s}_cli_bench_ramped.log 2>&1; echo "exit $?" >> $OUT/${s}_cli_bench_ramped.log
done
rm -rf $W
{
  echo "sanitizer runs of the host layer ($(date -u +%F)); one line per program: exit code, sanitizer reports"
  for f in $(ls $OUT/tsan_*.log $OUT/asan_*.log 2>/dev/null); do
    printf "%-44s %-8s reports: %s%s\n" "$(basename $f .log)" "$(tail -1 $f)" "$(grep -c -E 'WARNING: ThreadSanitizer|ERROR: AddressSanitizer|ERROR: LeakSanitizer' $f)" \
      "$(grep -q 'dev_runtime_unloaded_' $f && echo '   (AddressSanitizer CHECK dev_runtime_unloaded_ inside libamdhip64 __cxa_finalize at process exit: the runtime frees after the sanitizer device allocator is gone)')"
  done
} | tee $OUT/summary.txt
