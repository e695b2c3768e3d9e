# A/B of two builds of the engine library on ONE box (box-to-box variation is a few percent): tmp_ab/old_engine.so against
# tmp_ab/new_engine.so, each through tools/quick_gpu.py on the bench workload, twice in alternation.
cd /root/repo
cp lancet_amd/csrc/liblancet_engine.so tmp_ab/keep.so
for round in 1; do
  for which in old new; do
    cp tmp_ab/${which}_engine.so lancet_amd/csrc/liblancet_engine.so
    echo "== $which (round $round)"
    timeout 120 python tools/quick_gpu.py bench ${1:-32768} 2>&1 | grep -E "^run 2|LDS build kernel|build phase|total workgroup|first compress|transcript|per-comp|bfs|align|repeats|materialize|other"
  done
done
cp tmp_ab/keep.so lancet_amd/csrc/liblancet_engine.so
