#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $OUT/r04_c41_full.txt 2>&1
grep -E "passed|failed" $OUT/r04_c41_full.txt | tail -3 > $OUT/r04_c41_tests.txt
echo done
