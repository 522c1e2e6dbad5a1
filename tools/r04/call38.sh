#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for c in 3 4 5; do
  python $R/bench.py --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('contexts $c steps 20', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'])" >> $OUT/r04_c38_ctx.txt
done
done
for c in 3 4 5; do
  python $R/bench.py --steps 100 --warmup 8 --contexts $c --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('contexts $c steps 100', 'ms/step', d['ms_per_step'], 'windows', d['ms_per_step_windows'])" >> $OUT/r04_c38_ctx.txt
done
echo done
