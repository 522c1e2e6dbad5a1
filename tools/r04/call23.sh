#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 600 python tools/train_exec_diag.py > $OUT/r04_c23_diag.txt 2>&1
timeout 900 python -m pytest tests/test_unet_train_gpu.py -q -m gpu -s 2>&1 | tail -30 > $OUT/r04_c23_tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/train_step_bench.py 150000 > $OUT/r04_c23_train_step_bench.txt 2>&1
SG_TRAIN_EXEC=0 timeout 600 python $R/tools/train_step_bench.py 150000 > $OUT/r04_c23_train_step_bench_modules.txt 2>&1
echo done
