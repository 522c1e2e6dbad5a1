#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_lazy_results.py tests/test_native_scan_gpu.py tests/test_dropin_gpu.py tests/test_variants_gpu.py tests/test_cache_invalidation.py -x -q -m gpu > $OUT/r04_c37_pytest_full.txt 2>&1
grep -E "passed|failed|error" $OUT/r04_c37_pytest_full.txt | tail -3 > $OUT/r04_c37_tests.txt
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for e in 1 0; do
  SG_EARLY_COPY=$e python $R/bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('early $e', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'], 'one-at-a-time', d['ms_per_step_one_scan_at_a_time'], 'latency', d['latency_ms'])" >> $OUT/r04_c37_early.txt
done
done
echo done
