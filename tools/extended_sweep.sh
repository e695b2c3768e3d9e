# The device sweep of tests/test_engine_gpu_sweep.py on other draws than the committed ones, beside the suite (GPU box; ~25 s per set of
# 48 + 12 draws unless a draw holds heavy windows): plain, with --linked-reads on every draw, with the wide hand-off areas forced.
#   tools/extended_sweep.sh [first_offset] [sets_per_kind]
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
o=${1:-1000}; n=${2:-2}
for i in $(seq 1 $n); do
  echo "== plain, offset $o";  LANCET_SWEEP_OFFSET=$o timeout 900 python -m pytest tests/test_engine_gpu_sweep.py -m gpu -q 2>&1 | tail -1; o=$((o + 1000))
  echo "== linked reads, offset $o"; LANCET_SWEEP_OFFSET=$o LANCET_SWEEP_LINKED=1 timeout 900 python -m pytest tests/test_engine_gpu_sweep.py -m gpu -q 2>&1 | tail -1; o=$((o + 1000))
  echo "== wide hand-off areas, offset $o"; LANCET_SWEEP_OFFSET=$o LANCET_PRE_WIDE=1 timeout 900 python -m pytest tests/test_engine_gpu_sweep.py -m gpu -q 2>&1 | tail -1; o=$((o + 1000))
done
