#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_unet_train_gpu.py -q -m gpu > $OUT/r04_c40_full.txt 2>&1
grep -E "passed|failed|Error|assert" $OUT/r04_c40_full.txt | tail -12 > $OUT/r04_c40_tests.txt
echo done
