#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R


cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/r04_bench.json 2> $OUT/r04_bench.err
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_driver_shape.json 2> $OUT/r04_bench_driver_shape.err
echo done
