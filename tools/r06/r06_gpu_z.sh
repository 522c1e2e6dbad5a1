#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06z_driver_shape.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
run() {
  for i in 1 2 3 4; do
    env "$@" 2>&1 | python -c "
import json,sys
lines=sys.stdin.read().strip().splitlines()
d=json.loads(lines[-1])
print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency', d.get('latency_ms'))" >> $F
  done
}
for p in "" hnnll hllll hhnnl lnnnh; do
  echo "== SG_SCAN_PRIO=$p, 20 steps" >> $F
  run SG_SCAN_PRIO=$p $B --steps 20 --warmup 5
done
for p in "" hnnll hllll; do
  echo "== SG_SCAN_PRIO=$p, default steps" >> $F
  run SG_SCAN_PRIO=$p $B
done
echo done >> $F
