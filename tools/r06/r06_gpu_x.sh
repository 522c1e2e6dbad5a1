#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06x_hw_queues.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
run() {
  echo "== $*" >> $F
  for i in 1 2 3; do
    env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency', d.get('latency_ms'))" >> $F
  done
}
run A=1 $B --steps 20 --warmup 5
run GPU_MAX_HW_QUEUES=8 $B --steps 20 --warmup 5
run GPU_MAX_HW_QUEUES=16 $B --steps 20 --warmup 5
run GPU_MAX_HW_QUEUES=2 $B --steps 20 --warmup 5
run A=1 $B
run GPU_MAX_HW_QUEUES=8 $B
run GPU_MAX_HW_QUEUES=16 $B
echo done
