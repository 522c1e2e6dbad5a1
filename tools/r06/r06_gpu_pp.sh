#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
F=$OUT/r06_pp_defer.txt
: > $F
timeout 1500 python -m pytest tests/test_native_scan_gpu.py tests/test_parity_at_size.py -x -q -k "pp or config4 or stpls" 2>&1 | tail -4 >> $F
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
  echo "== SG_PP_DEFER=$v" >> $F
  SG_PP_DEFER=$v timeout 300 python $R/tools/scan_only.py 30 150000 stpls3d_pp 2>&1 | tail -1 >> $F
done
echo done >> $F
