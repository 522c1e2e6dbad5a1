#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
F=$OUT/r06y_bfs_local.txt
: > $F
SG_BFS_BIG_LOCAL=1 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "bfs" 2>&1 | tail -8 >> $F
cd /tmp && export TMPDIR=/tmp
for cfg in kitti stpls3d_pp; do
for v in "1 16 4" "1 24 4" "1 32 4"; do
set -- $v
export SG_BFS_BIG_LOCAL=$1 SG_BFS_BIG_LOCAL_WGS=$2 SG_BFS_BIG_LOCAL_EVERY=$3
rm -rf /tmp/prof_y
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -o y -- python $R/tools/scan_only.py 12 150000 $cfg > /dev/null 2>&1
python - <<PY >> $F
import csv,glob
f=glob.glob('/tmp/prof_y/**/*kernel_stats.csv',recursive=True)
print('== $cfg local=$1 wgs=$2 every=$3')
for r in csv.DictReader(open(f[0])):
    if 'bfs_emit_big' in r['Name'] and float(r['AverageNs'])>20000: print('  ', r['Name'][:40], r['Calls'], r['AverageNs'])
PY
[ $1 = 1 ] && SG_BFS_STATS=1 timeout 300 python $R/tools/scan_only.py 2 150000 $cfg 2>&1 | grep -E "local form|thin levels" | tail -1 >> $F
for i in 1 2; do timeout 300 python $R/tools/scan_only.py 30 150000 $cfg 2>&1 | tail -1 >> $F; done
done
done
echo done >> $F
