#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_native_scan_gpu.py -m gpu -x -q -k "softgroup_pp" 2>&1 | tail -30 > $OUT/r06h_tests.txt
timeout 900 python -m pytest tests/test_parity_at_size.py tests/test_variants_gpu.py tests/test_native_scan_gpu.py -m gpu -x -q 2>&1 | tail -8 >> $OUT/r06h_tests.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT/r06h_pp_ab.txt
for T in 1 0 1 0; do
  echo "== SG_NATIVE_GROUPING_PP=$T" >> $OUT/r06h_pp_ab.txt
  SG_NATIVE_GROUPING_PP=$T timeout 300 python $R/tools/scan_only.py 30 150000 stpls3d_pp 2>/dev/null | tail -1 >> $OUT/r06h_pp_ab.txt
done
rm -rf /tmp/prof_scan
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 stpls3d_pp > /dev/null 2>&1
python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/r06h_scan_stpls3d_pp pointwise_heads_kernel
echo done
