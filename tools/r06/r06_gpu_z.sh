#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06z_driver_shape.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
for i in 1 2 3 4 5 6 7 8; do
  $B --steps 20 --warmup 5 2>&1 | python -c "
import json,sys
lines=sys.stdin.read().strip().splitlines()
d=json.loads(lines[-1])
print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency', d.get('latency_ms'))" >> $F
done
echo done >> $F
