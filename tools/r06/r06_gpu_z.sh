#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06z_pinned.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
one() {
  "$@" 2>&1 | python -c "
import json,sys
lines=sys.stdin.read().strip().splitlines()
d=json.loads(lines[-1])
print(d['ms_per_step'], d.get('ms_per_step_windows'))"
}
for i in 1 2 3 4; do
  echo "160 steps pinned 256: $(one $B)" >> $F
  echo "160 steps pinned 512: $(SG_PINNED_RESULTS_MB=512 one $B)" >> $F
done
for i in 1 2 3 4; do
  echo "20 steps pinned 256: $(one $B --steps 20 --warmup 5)" >> $F
  echo "20 steps pinned 512: $(SG_PINNED_RESULTS_MB=512 one $B --steps 20 --warmup 5)" >> $F
done
SG_BENCH_DIAG=1 $B 2>&1 | grep "bench diag" | cut -c1-200 >> $F
echo done >> $F
