#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_native_scan_gpu.py tests/test_variants_gpu.py -m gpu -q > $OUT/r04_c12_pytest.txt 2>&1
echo "pytest rc $?" >> $OUT/r04_c12_pytest.txt
cd /tmp && export TMPDIR=/tmp
for v in "SG_BFS_BIG_WGS=32" "SG_BFS_BIG_WGS=16" "SG_BFS_BIG_WGS=64" "SG_BFS_BIG_WGS=128" "SG_BFS_BIG_WGS=8"; do
  echo "== $v dense: $(env $v python $R/tools/dense_profile.py 300000 10 2>&1 | tail -1)" >> $OUT/r04_c12_bfs_wgs.txt
  echo "== $v kitti: $(env $v python $R/tools/host_profile.py 10 kitti 2>&1 | grep 'ms/scan under')" >> $OUT/r04_c12_bfs_wgs.txt
  echo "== $v stpls3d: $(env $v python $R/tools/host_profile.py 10 stpls3d_pp 2>&1 | grep 'ms/scan under')" >> $OUT/r04_c12_bfs_wgs.txt
done
echo done
