#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for c in 4 8 12; do
  python $R/bench.py --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('steps20 contexts $c', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'], 'host', d['host_threads']['scan_threads'])" >> $OUT/r04_c14_contexts.txt
done
done
echo done
