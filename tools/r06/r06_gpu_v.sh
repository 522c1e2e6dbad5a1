#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06v_host_side.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
run() {
  echo "== $*" >> $F
  for i in 1 2 3; do
    env "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'))" >> $F
  done
}
run A=1 $B
run SG_BENCH_SKIP_DIGEST=1 $B
run A=1 $B --switch-interval-us 500
run A=1 $B --switch-interval-us 100
run A=1 $B --contexts 8
run SG_BENCH_SKIP_DIGEST=1 $B --contexts 8 --switch-interval-us 200
echo done
