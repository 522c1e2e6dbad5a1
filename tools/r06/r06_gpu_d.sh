#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT/r06d_chain_throughput.txt
run() {
  echo "== $*" >> $OUT/r06d_chain_throughput.txt
  for i in 1 2; do
    env "$@" timeout 300 python $R/bench.py --no-cpu-baseline --no-legs --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'windows', d.get('ms_per_step_windows'), 'latency_ms', d.get('latency_ms'))" >> $OUT/r06d_chain_throughput.txt
  done
}
run SG_CONV_CHAIN=0
run SG_CONV_CHAIN=1
run SG_CONV_CHAIN=1 SG_CHAIN_SERIAL=0
run SG_CONV_CHAIN=1 SG_CHAIN_GRID=128
run SG_CONV_CHAIN=1 SG_CHAIN_GRID=64
run SG_CONV_CHAIN=1 SG_CONV_CHAIN_ROWS=1000
run SG_CONV_CHAIN=1 SG_CONV_CHAIN_ROWS=1000 SG_CHAIN_GRID=128
echo done
