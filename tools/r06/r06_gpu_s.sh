#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06s_gc_order.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline --steps 20 --warmup 5"
for rep in 1 2 3 4 5; do
for V in 1 0; do
  SG_BENCH_GC_AFTER_WARMUP=$V $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gc_after_warmup $V rep $rep ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'))" >> $F
done; done
echo done
