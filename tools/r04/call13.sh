#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "SG_X=1" "SG_CONV_KSPLIT_WV=8" "SG_CONV_KSPLIT_WV=2"; do
  echo "== $v: $(env $v timeout 120 python $R/tools/conv_only.py 20 2>&1 | tail -1)" >> $OUT/r04_c13_conv.txt
done
for c in 4 6 8; do
  python $R/bench.py --steps 40 --warmup 8 --contexts $c --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('contexts $c', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'], 'one-at-a-time', d['ms_per_step_one_scan_at_a_time'], 'latency', d['latency_ms'])" >> $OUT/r04_c13_contexts.txt
done
for c in 4 6; do
  python $R/bench.py --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('steps20 contexts $c', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'])" >> $OUT/r04_c13_contexts.txt
done
echo done
