#!/bin/bash
# On the GPU box: quick_check of the product library (or TSQ_LIB), output kept under gpurun_out/q/<name>.txt.  tools/gq.sh <name> [quick_check args]
n=${1:-q}; shift
mkdir -p gpurun_out/q
timeout 900 python tools/quick_check.py "$@" > gpurun_out/q/$n.txt 2>&1
grep -E "PARITY|TIME|BIG|FAIL|Error|error" gpurun_out/q/$n.txt
