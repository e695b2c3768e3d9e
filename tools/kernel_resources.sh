#!/bin/bash
# VGPR / SGPR / scratch / spill figures of the gfx950 kernels (device-only compile, then the code-object notes)
set -e
T=$(mktemp -d); cd "$T"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result --cuda-device-only -c /root/repo/lancet_amd/csrc/engine.hip -o dev.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=dev.co --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=dev2.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes dev2.co | grep -E "\.name:|\.vgpr_count|private_segment_fixed|\.sgpr_count|vgpr_spill"
