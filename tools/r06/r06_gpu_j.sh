#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/r06j_tests.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT/r06j_side_ab.txt
for T in 1 0 1 0; do
  echo "== SG_BFS_BIG_SIDE=$T" >> $OUT/r06j_side_ab.txt
  for CFG in kitti stpls3d_pp scannet; do
    SG_BFS_BIG_SIDE=$T timeout 300 python $R/tools/scan_only.py 30 150000 $CFG 2>/dev/null | tail -1 >> $OUT/r06j_side_ab.txt
  done
done
rm -rf /tmp/prof_scan
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_scan -o r -- python $R/tools/scan_only.py 16 150000 kitti > /dev/null 2>&1
python $R/tools/scan_sequence.py /tmp/prof_scan $OUT/r06j_scan_kitti pointwise_heads_kernel
echo done
