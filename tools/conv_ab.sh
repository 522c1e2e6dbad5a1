#!/bin/bash
# Developer A/B of the sparse-conv kernel on one box: the round-2 kernel (libsoftgroup_hip_r02conv.so,
# built by hand from the previous source), this round's kernel with the static hand-out, and with
# the ticket hand-out.  Conv time per scan comes from the in-library HIP events (bench.py roofline).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for v in "SG_LIB_NAME=libsoftgroup_hip_r02conv.so" "SG_CONV_STATIC=1" "SG_CONV_STATIC=0"; do
  env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'conv ms/scan', r['kernel_ms_per_scan'], 'frac', r['frac'], 'ms/step', d['ms_per_step'], 'latency', d['latency_ms'])"
done
done
