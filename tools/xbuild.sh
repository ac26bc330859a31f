#!/bin/bash
# Experiment builds: tools/xbuild.sh name "-DFLAG=1 ..." [name "flags" ...]  ->  turbosqueeze_amd/libturbosqueeze_amd_x_<name>.so
# and tools/xrun.sh runs quick_check --enc-only over them on the GPU box.
set -e
cd "$(dirname "$0")/../turbosqueeze_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DTSQ_EXPERIMENT $flags -shared -o ../libturbosqueeze_amd_x_$name.so tsq_runtime.hip tsq_compat.hip -lpthread &
done
wait
ls -la ../libturbosqueeze_amd_x_*.so
