#!/bin/bash
# Developer A/B of the sparse-conv kernel variants on one box (environment knobs of
# csrc/spconv_conv.hip).  Conv time per scan comes from the in-library HIP events (bench.py roofline).
#   bash tools/conv_ab.sh "SG_CONV_SPLIT=0" "SG_CONV_SPLIT=1" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "$@"; do
  env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'conv ms/scan', r['kernel_ms_per_scan'], 'frac', r['frac'], 'ms/step', d['ms_per_step'], 'latency', d['latency_ms'])"
done
done
