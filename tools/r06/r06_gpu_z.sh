#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06z_shared_copy.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
one() {
  "$@" 2>&1 | python -c "
import json,sys
lines=sys.stdin.read().strip().splitlines()
d=json.loads(lines[-1])
print(d['ms_per_step'], d.get('ms_per_step_windows'))"
}
for i in 1 2 3 4 5; do
  for t in 0 1; do echo "160 steps shared copy $t: $(SG_SCAN_SHARED_COPY=$t one $B)" >> $F; done
done
for i in 1 2 3 4; do
  for t in 0 1; do echo "20 steps shared copy $t: $(SG_SCAN_SHARED_COPY=$t one $B --steps 20 --warmup 5)" >> $F; done
done
echo done >> $F
