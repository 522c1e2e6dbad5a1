#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1200 python -X faulthandler -m pytest tests/test_native_scan_gpu.py -m gpu -x -q -k "variants" > /tmp/t.log 2>&1
grep -n "Error\|assert \|Fatal" /tmp/t.log | head -12 > $OUT/r06w_tests.txt
tail -3 /tmp/t.log | cut -c1-300 >> $OUT/r06w_tests.txt
echo done
