# Several builds of the engine library on ONE box (tmp_ab/<name>_engine.so, tools/build_variant.sh), each through tools/quick_gpu.py on the
# bench workload: kernel times and the build kernel's phase profile.   usage: ab_many.sh name1 name2 ...
cd /root/repo
cp lancet_amd/csrc/liblancet_engine.so tmp_ab/keep.so
for which in "$@"; do
  cp tmp_ab/${which}_engine.so lancet_amd/csrc/liblancet_engine.so
  echo "== $which"
  timeout 120 python tools/quick_gpu.py bench ${WINDOWS:-32768} 2>&1 | grep -E "${AB_GREP:-^run 2|LDS build kernel|build phase  (4|8|10|14|15)|total workgroup}"
done
cp tmp_ab/keep.so lancet_amd/csrc/liblancet_engine.so
