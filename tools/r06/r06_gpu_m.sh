#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
SG_CONV_STATIC=0 timeout 900 python -m pytest tests/test_spconv_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -4 > $OUT/r06m_tests.txt
cd /tmp && export TMPDIR=/tmp
: > $OUT/r06m_dyn_ab.txt
for rep in 1 2; do
for V in "SG_CONV_STATIC=1" "SG_CONV_STATIC=0 SG_CONV_DYN_ROUNDS=2" "SG_CONV_STATIC=0 SG_CONV_DYN_ROUNDS=1"; do
  echo "== $V" >> $OUT/r06m_dyn_ab.txt
  env $V timeout 300 python $R/tools/conv_exec_layers.py 150000 10 2>&1 | grep -v amdgpu | grep "^ 27\|total" | head -14 >> $OUT/r06m_dyn_ab.txt
  env $V timeout 300 python $R/tools/scan_only.py 30 150000 scannet 2>/dev/null | tail -1 >> $OUT/r06m_dyn_ab.txt
done; done
SG_BFS_STATS=1 timeout 300 python $R/tools/scan_only.py 6 150000 kitti 2>&1 | grep "giant clusters" | tail -2 > $OUT/r06m_bfs_big_phases.txt
timeout 300 python $R/tools/scan_only.py 30 150000 kitti 2>/dev/null | tail -1 >> $OUT/r06m_bfs_big_phases.txt
echo done
