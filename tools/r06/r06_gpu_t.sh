#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1200 python -X faulthandler -m pytest tests/test_train_gpu.py -m gpu -x -q -s -k "follows_fp32" > /tmp/t.log 2>&1
grep "^step " /tmp/t.log | cut -c1-500 > $OUT/r06t_tests.txt
grep -n "Error\|assert " /tmp/t.log | head -8 >> $OUT/r06t_tests.txt
tail -2 /tmp/t.log | cut -c1-300 >> $OUT/r06t_tests.txt
echo done
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/train_step_bench.py 2>/dev/null | grep "ms/step" > $OUT/r06_train_step.txt
cat $OUT/r06_train_step.txt >> $OUT/r06t_tests.txt
