#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_unet_train_gpu.py -q -m gpu -s -k "forward_train" 2>&1 | grep -E "Error|assert|^E|passed|failed|relative L2" | head -40 > $OUT/r04_c24_tests.txt
echo done
