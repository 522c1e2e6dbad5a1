#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06g_token_matrix.txt
: > $F
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline --steps 20 --warmup 5"
for rep in 1 2 3 4; do
for T in 0 1 2; do
for CTX in 3 4 5; do
  SG_SCAN_TOKEN=$T $B --contexts $CTX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('token $T contexts $CTX rep $rep ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'))" >> $F
done; done; done
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
for rep in 1 2; do
for T in 0 1 2; do
for CTX in 4 5; do
  SG_SCAN_TOKEN=$T $B --contexts $CTX 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('160 steps: token $T contexts $CTX rep $rep ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'))" >> $F
done; done; done
echo done
