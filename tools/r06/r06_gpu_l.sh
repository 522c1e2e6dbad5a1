#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -x -q -s -k "losses_explained" 2>&1 | grep -v "Warning\|warnings.warn\|^$" | tail -40 > $OUT/r06l_tests.txt
timeout 600 python -m pytest tests/test_data_gpu.py tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -4 >> $OUT/r06l_tests.txt
cd /tmp && export TMPDIR=/tmp
SG_BFS_STATS=1 timeout 300 python $R/tools/scan_only.py 6 150000 kitti 2>&1 | grep "giant clusters" | tail -3 > $OUT/r06l_bfs_big_phases.txt
SG_BFS_STATS=1 timeout 300 python $R/tools/scan_only.py 6 150000 stpls3d_pp 2>&1 | grep "giant clusters" | tail -3 >> $OUT/r06l_bfs_big_phases.txt
echo done
