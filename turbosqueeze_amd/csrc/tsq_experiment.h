// tsq_experiment.h -- the guard around build switches that change what the kernels compute or how they are timed.
//
// Three kinds of switches exist in the kernel sources:
//   tunables          TSQ_LM / TSQ_LF / TSQ_RECORDS / TSQ_OWNBITS / TSQ_WIN / TSQ_HASH_AHEAD: every value gives the oracle's streams
//                     (tools/xbuild.sh builds them, tools/quick_check.py checks them); the product builds with the defaults.
//   instruments       TSQ_STATS, TSQ_SPINS, TSQ_TRACEONLY, TSQ_REGION, TSQ_JITTER, TSQ_X_DELAY_STAGE: streams unchanged, timing changed;
//                     never in the product library (separate make targets, separate .so names).
//   timing-only       TSQ_X_NOHAZ, TSQ_X_NOCOMMITWAIT, TSQ_X_NOPATCH, TSQ_X_FAKE_TABLE, TSQ_X_FAKE_CAND, ...: WRONG STREAMS on purpose, to bound
//                     what a part of the pipeline costs.  They compile only together with -DTSQ_EXPERIMENT (tools/xbuild.sh passes it);
//                     a product build that carries one of them by accident does not compile.
// tsqa_build_info() reports which of the three were compiled into a library; tests/test_abi_cpu.py asserts the product has none.
#pragma once

#if defined(TSQ_X_NOHAZ) || defined(TSQ_X_NOCOMMITWAIT) || defined(TSQ_X_NOPATCH) || defined(TSQ_X_FAKE_TABLE) || defined(TSQ_X_FAKE_CAND) || \
    defined(TSQ_X_FREE_QUERY)
#define TSQ_TIMING_ONLY_BUILD 1
#ifndef TSQ_EXPERIMENT
#error "a timing-only switch (TSQ_X_*: wrong streams on purpose) needs -DTSQ_EXPERIMENT; the product library is never built with one (tools/xbuild.sh)"
#endif
#else
#define TSQ_TIMING_ONLY_BUILD 0
#endif

#if defined(TSQ_STATS) || defined(TSQ_SPINS) || defined(TSQ_TRACEONLY) || defined(TSQ_REGION) || defined(TSQ_JITTER) || defined(TSQ_X_DELAY_STAGE)
#define TSQ_INSTRUMENTED_BUILD 1
#else
#define TSQ_INSTRUMENTED_BUILD 0
#endif
