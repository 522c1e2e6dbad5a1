#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
F=$OUT/r06_select_scan.txt
: > $F
timeout 1500 python -m pytest tests/test_native_scan_gpu.py tests/test_model_gpu.py tests/test_parity_at_size.py -x -q 2>&1 | tail -2 >> $F
cd /tmp && export TMPDIR=/tmp
for cfg in scannet; do
rm -rf /tmp/prof_y
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_y -o y -- python $R/tools/scan_only.py 12 150000 $cfg > /dev/null 2>&1
python - <<PY >> $F
import csv,glob
f=glob.glob('/tmp/prof_y/**/*kernel_stats.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'select_' in r['Name']: print('$cfg', r['Name'][:44], r['Calls'], r['AverageNs'])
PY
for i in 1 2; do timeout 300 python $R/tools/scan_only.py 30 150000 $cfg 2>&1 | tail -1 >> $F; done
done
echo done >> $F
