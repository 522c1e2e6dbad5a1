#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06y_bfs_local.txt
: > $F
for cfg in kitti stpls3d_pp; do
for v in "8 4" "12 4" "16 4" "12 3" "16 3" "20 4"; do
set -- $v
export SG_BFS_BIG_LOCAL_WGS=$1 SG_BFS_BIG_LOCAL_EVERY=$2
rm -rf /tmp/prof_y
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -o y -- python $R/tools/scan_only.py 12 150000 $cfg > /dev/null 2>&1
python - <<PY >> $F
import csv,glob
f=glob.glob('/tmp/prof_y/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'bfs_emit_big_local' in r['Name']: print('$cfg wgs=$1 every=$2', r['Calls'], r['AverageNs'])
PY
done
done
echo done >> $F
