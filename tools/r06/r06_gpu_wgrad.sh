#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
F=$OUT/r06_wgrad.txt
: > $F
timeout 1200 python -m pytest tests/test_train_gpu.py tests/test_spconv_gpu.py -x -q 2>&1 | tail -2 >> $F
cd /tmp && export TMPDIR=/tmp
python $R/tools/train_conv_bench.py 2>/dev/null | tail -10 >> $F
python $R/tools/train_step_profile.py 100000 fp32 2>/dev/null | tail -1 >> $F
python $R/tools/train_step_profile.py 100000 fp32 2>/dev/null | tail -1 >> $F
echo done >> $F
