#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06z_contexts.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
run() {
  for i in 1 2 3; do
    "$@" 2>&1 | python -c "
import json,sys
lines=sys.stdin.read().strip().splitlines()
d=json.loads(lines[-1])
print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'in flight', d.get('scans_in_flight'))" >> $F
  done
}
nproc >> $F
for c in 8 12 16; do
  echo "== contexts $c, default steps" >> $F
  run $B --contexts $c
done
for c in 5 8 10 12 16; do
  echo "== contexts $c, 20 steps" >> $F
  run $B --contexts $c --steps 20 --warmup 5
done
echo done >> $F
