#!/bin/bash
# Tuning aid (GPU box): builds the engine with extra compiler flags into a scratch copy and prints the bench workload's kernel times.
#   tools/variant.sh "-DBL_SMALL_WG=768 -DBL_SMALL_EU=6" "-DLANCET_PROF=2" ...     (one variant per argument; "" = the tree as it is)
cd "$(dirname "$0")/.."
n=0
for flags in "$@"; do
  n=$((n+1)); d=/tmp/var_$n; rm -rf $d; mkdir -p $d
  cp -r lancet_amd include oracle tools tests $d/ 2>/dev/null
  if (cd $d/lancet_amd/csrc && for f in engine window_fat; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $flags -c $f.hip -o $f.o || exit 1; done &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC engine.o window_fat.o host_vdb.o host_frontend.o host_trace.o -lz -lpthread -o liblancet_engine.so) > $d/build.log 2>&1; then
    echo "== variant $n: $flags"
    (cd $d && timeout ${VAR_TIMEOUT:-120} python tools/quick_gpu.py ${VAR_CASE:-bench} ${WINDOWS:-32768} 2>&1 | grep -E "${VAR_GREP:-^run 2|total slot|kernel ms|build phase|total workgroup}")
  else echo "== variant $n: $flags: BUILD FAILED"; tail -5 $d/build.log; fi
done
