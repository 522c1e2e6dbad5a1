#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/r04_bench.json 2> $OUT/r04_bench.err
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_driver_shape.json 2> $OUT/r04_bench_driver_shape.err
cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/r04_c39_smoke.txt 2>&1
echo done
