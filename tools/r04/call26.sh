#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_native_scan_gpu.py tests/test_train_gpu.py tests/test_unet_train_gpu.py -q -m gpu 2>&1 | tail -12 > $OUT/r04_c26_tests.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/tools/train_profile.py 20 plain > $OUT/r04_c26_train_plain.txt 2>&1
timeout 300 python $R/tools/train_profile.py 20 plain16 > $OUT/r04_c26_train_plain16.txt 2>&1
timeout 300 python $R/tools/train_profile.py 10 stages16 > $OUT/r04_c26_train_stages16.txt 2>&1
timeout 300 python $R/tools/train_profile.py 10 host16 > $OUT/r04_c26_train_host16.txt 2>&1
echo done
