#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_data_gpu.py -m gpu -x -q 2>&1 | tail -2 > $OUT/r06u_summary.txt
cd /tmp && export TMPDIR=/tmp
for LD in 1 2; do
SG_BENCH_LOADERS=$LD python $R/bench.py --no-cpu-baseline > $OUT/r06u_bench.json 2> $OUT/r06u_bench.err
python -c "
import json
d=json.loads(open('$OUT/r06u_bench.json').read().strip().splitlines()[-1])
print('loaders $LD: ms_per_step', d['ms_per_step'], d['ms_per_step_windows'], 'latency', d['latency_ms'])
print(d['legs']['with_h2d'])
" >> $OUT/r06u_summary.txt 2>&1
done
echo done
