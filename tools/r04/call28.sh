#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_unet_train_gpu.py tests/test_native_scan_gpu.py -q -m gpu 2>&1 | tail -4 > $OUT/r04_c28_tests.txt
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for sw in 0 200 50; do
  python $R/bench.py --steps 20 --warmup 5 --switch-interval-us $sw --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('switch $sw', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'], 'one-at-a-time', d['ms_per_step_one_scan_at_a_time'], 'latency', d['latency_ms'])" >> $OUT/r04_c28_switch.txt
done
done
timeout 300 python $R/tools/train_profile.py 20 plain16 > $OUT/r04_c28_train_plain16.txt 2>&1
echo done
