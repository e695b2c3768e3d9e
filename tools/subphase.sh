#!/bin/bash
# Profiling aid: builds the engine with -DLANCET_PROF=<group> (kernels.h SUBPHASE: the steps inside one coarse phase of the window kernel,
# accounted in the slots of the general build's phases) into csrc/prof<group>/ and prints the per-phase slot time of the bench workload.
# Run on the GPU box: [MODE=bench5 WINDOWS=8192] tools/subphase.sh 1 2 3 4 5     (1 first compress, 2 per-component passes, 3 path search, 4 transcript walk,
# 5 build_gather of a linked-read batch, on top of the small slots 11-13: 5 node order, 11 lane-per-node loop, 12 linked-read replay by the wave, 13 the rest)
set -e
cd "$(dirname "$0")/.."
for g in "$@"; do
  d=/tmp/prof$g; rm -rf $d; mkdir -p $d
  cp -r lancet_amd include oracle tools tests $d/ 2>/dev/null
  (cd $d/lancet_amd/csrc && for f in engine window_fat; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DLANCET_PROF=$g -c $f.hip -o $f.o; done &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC engine.o window_fat.o host_vdb.o host_frontend.o host_trace.o -lz -lpthread -o liblancet_engine.so)
  echo "== LANCET_PROF=$g"
  (cd $d && python tools/quick_gpu.py ${MODE:-bench} ${WINDOWS:-32768} 2>&1 | grep -E "^  phase|total slot|kernel ms")
done
