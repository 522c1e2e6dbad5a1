#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_scan_contexts_gpu.py tests/test_conv_chain_gpu.py -m gpu -x -q 2>&1 | tail -4 > $OUT/r06f_tests.txt
SG_SCAN_TOKEN=2 timeout 900 python -m pytest tests/test_scan_contexts_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $OUT/r06f_tests.txt
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06f_token.txt
: > $F
run() {
  echo "== $*" >> $F
  for i in 1 2 3; do
    env "$@" 2>>$F | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency_ms', d.get('latency_ms'))" >> $F
  done
}
B="python $R/bench.py --no-cpu-baseline --no-legs --no-roofline"
run SG_SCAN_TOKEN=0 $B --steps 20 --warmup 5
run SG_SCAN_TOKEN=2 $B --steps 20 --warmup 5
run SG_SCAN_TOKEN=1 $B --steps 20 --warmup 5
run SG_SCAN_TOKEN=0 $B
run SG_SCAN_TOKEN=2 $B
run SG_SCAN_TOKEN=2 $B --contexts 4 --steps 20 --warmup 5
run SG_SCAN_TOKEN=2 $B --contexts 4
run SG_SCAN_TOKEN=2 $B --contexts 6 --steps 20 --warmup 5
echo "== diag, token 2, 20 steps" >> $F
SG_SCAN_TOKEN=2 SG_BENCH_DIAG=1 $B --steps 20 --warmup 5 2>&1 >/dev/null | grep "bench diag" >> $F
echo done
