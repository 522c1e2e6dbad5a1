#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
F=$OUT/r06_wgrad.txt
: > $F
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -3 >> $F
cd /tmp && export TMPDIR=/tmp
python $R/tools/train_step_profile.py 100000 fp32 2>/dev/null | tail -1 >> $F
python $R/tools/train_step_profile.py 100000 bf16 2>/dev/null | tail -1 >> $F
rm -rf /tmp/prof_t
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $R/tools/train_step_profile.py 100000 fp32 > /dev/null 2>&1
python - <<PY >> $F
import csv,glob
f=glob.glob('/tmp/prof_t/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('kernel time per step %.2f ms, %d launches per step (11 steps traced)' % (tot/11/1e6, calls/11))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:26]:
    print('  %-70s %6d calls/step %8.1f us each %7.3f ms/step' % (r['Name'][:70], int(r['Calls'])/11, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/11/1e6))
PY
echo done >> $F
