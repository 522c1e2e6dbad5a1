#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_lazy_results.py tests/test_native_scan_gpu.py tests/test_dropin_gpu.py tests/test_variants_gpu.py tests/test_cache_invalidation.py -x -q -m gpu 2>&1 | tail -4 > $OUT/r04_c36_tests.txt
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for st in "20 5" "160 8"; do
  set -- $st
  python $R/bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('steps $1', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'], 'one-at-a-time', d['ms_per_step_one_scan_at_a_time'], 'latency', d['latency_ms'], d.get('stages_ms'))" >> $OUT/r04_c36_early.txt
done
done
echo done
