# The GPU suite with a tier-1 node table of 64 entries: practically every window is assembled by the several-wave kernel of the
# re-run tier (window_fat.hip).  The tests skip their asserts on how many windows tier 1 itself served (tests/test_engine_gpu.py TIER1);
# two tests are left out because such counts are all they add.
cd /root/repo; mkdir -p gpurun_out
LANCET_NODE_CAP1=64 python -m pytest tests/test_engine_gpu.py tests/test_engine_gpu_sweep.py -m gpu -q \
  --deselect tests/test_engine_gpu.py::test_graphs_built_ahead_after_the_reference_table_was_trimmed \
  --deselect tests/test_engine_gpu.py::test_coverage_pile_up_window_does_not_size_the_whole_batch 2>&1 | tee gpurun_out/fat_check.log | tail -n 5
