#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SG_BFS_STATS=1 python $R/tools/host_profile.py 1 scannet 2>&1 | grep "bfs cluster" | tail -25 > $OUT/r04_c33_bfs_stats.txt
echo done
