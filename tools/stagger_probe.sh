# Tuning aid (GPU box): the next batch's build kernel started when this batch's build kernel is through (LANCET_STAGGER=1) instead of when its
# window kernel is, with the window kernel on fewer slots so that a build workgroup fits beside them on every CU.
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 40 --settle 50 --cpu-sample 0 --no-configs --no-bam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['roofline'].get('per_kernel_ms'), d['config'].get('records_sha256_rank0_contig'))"; }
run X=1
run LANCET_STAGGER=1
run LANCET_STAGGER=1 LANCET_MAX_SLOTS=2048
run LANCET_STAGGER=1 LANCET_MAX_SLOTS=1536
run LANCET_STAGGER=1 LANCET_MAX_SLOTS=2560
run LANCET_STAGGER=1 LANCET_MAX_SLOTS=2048 LANCET_BUILD_SLOTS=256
run X=1
