#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_unet_train_gpu.py tests/test_train_gpu.py tests/test_dropin_gpu.py -q -m gpu 2>&1 | tail -12 > $OUT/r04_c25_tests.txt
echo done
