#!/bin/bash
# On the GPU box: parity + kernel times for every experiment library (tools/xbuild.sh).  Extra arguments go to quick_check.
mkdir -p gpurun_out/x
for f in turbosqueeze_amd/libturbosqueeze_amd_x_*.so; do
  n=$(basename $f .so); n=${n#libturbosqueeze_amd_x_}
  TSQ_LIB=$(basename $f) timeout 600 python tools/quick_check.py "$@" > gpurun_out/x/$n.txt 2>&1
  echo "== $n: $(grep -c ok gpurun_out/x/$n.txt) ok lines; $(grep -E 'PARITY|TIME|BIG|FAIL|Error' gpurun_out/x/$n.txt | tr '\n' ' ')"
done
