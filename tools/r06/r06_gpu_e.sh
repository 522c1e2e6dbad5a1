#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06e_token.txt
: > $F
run() {
  echo "== $*" >> $F
  for i in 1 2 3; do
    env "$@" 2>>$F | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency_ms', d.get('latency_ms'))" >> $F
  done
}
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
run SG_CONV_CHAIN=0 SG_SCAN_TOKEN=0 $B --steps 20 --warmup 5
run SG_CONV_CHAIN=0 SG_SCAN_TOKEN=1 $B --steps 20 --warmup 5
run SG_CONV_CHAIN=0 SG_SCAN_TOKEN=0 $B
run SG_CONV_CHAIN=0 SG_SCAN_TOKEN=1 $B
run SG_CONV_CHAIN=0 SG_SCAN_TOKEN=1 $B --contexts 3 --steps 20 --warmup 5
run SG_CONV_CHAIN=0 SG_SCAN_TOKEN=1 $B --contexts 4 --steps 20 --warmup 5
run SG_CONV_CHAIN=0 SG_SCAN_TOKEN=0 $B --contexts 3 --steps 20 --warmup 5
run SG_CONV_CHAIN=1 SG_SCAN_TOKEN=1 $B --steps 20 --warmup 5
run SG_CONV_CHAIN=1 SG_SCAN_TOKEN=1 $B
echo "== diag, token 0 / 1, 20 steps" >> $F
SG_CONV_CHAIN=0 SG_SCAN_TOKEN=0 SG_BENCH_DIAG=1 $B --steps 20 --warmup 5 2>&1 >/dev/null | grep "bench diag" >> $F
SG_CONV_CHAIN=0 SG_SCAN_TOKEN=1 SG_BENCH_DIAG=1 $B --steps 20 --warmup 5 2>&1 >/dev/null | grep "bench diag" >> $F
echo done
