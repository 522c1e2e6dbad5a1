#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06z_conv_target.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
one() {
  "$@" 2>&1 | python -c "
import json,sys
lines=sys.stdin.read().strip().splitlines()
d=json.loads(lines[-1])
print(d['ms_per_step'], d.get('ms_per_step_windows'))"
}
for i in 1 2 3; do
  echo "160 steps target default: $(one $B)" >> $F
  for t in 128 256 512 2048; do echo "160 steps target $t: $(SG_CONV_TARGET=$t one $B)" >> $F; done
done
for t in 1024 256 1024 256; do echo "latency target $t: $(cd $R; SG_CONV_TARGET=$t python tools/scan_only.py 30 150000 scannet 2>/dev/null | tail -1)" >> $F; done
echo done >> $F
