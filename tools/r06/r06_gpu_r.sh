#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06r_readback.txt
: > $F
for V in 0 1 0 1 0 1; do
  echo "== SG_READBACK_KERNEL=$V" >> $F
  SG_READBACK_KERNEL=$V timeout 300 python $R/tools/scan_only.py 30 150000 scannet 2>/dev/null | tail -1 >> $F
done
cd $R
SG_READBACK_KERNEL=1 timeout 600 python -m pytest tests/test_native_scan_gpu.py tests/test_scan_forward_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $F
echo done
