#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/r06k_tests.txt
echo done
