#!/bin/bash
# Builds the engine library with extra compiler flags into tmp_ab/<name>_engine.so (here, not on the GPU box: compile time there is GPU budget).
#   tools/build_variant.sh name "-DBL_INFLIGHT=8"
name=$1; flags=$2
d=$(mktemp -d); cp -r /root/repo/lancet_amd/csrc/* $d/; cp -r /root/repo/include $d/../include 2>/dev/null
cd $d && for f in engine window_fat; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -I/root/repo/include -I/root/repo/lancet_amd/csrc $flags -c $f.hip -o $f.o || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC engine.o window_fat.o host_vdb.o host_frontend.o host_trace.o -lz -lpthread -o /root/repo/tmp_ab/${name}_engine.so && echo "built tmp_ab/${name}_engine.so"
