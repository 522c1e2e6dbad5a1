#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06_train3_profile.txt
: > $F
python $R/tools/train_profile.py 12 plain 2>/dev/null | tail -3 >> $F
rm -rf /tmp/prof_t
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python $R/tools/train_profile.py 12 plain > /dev/null 2>&1
python - <<PY >> $F
import csv,glob
f=glob.glob('/tmp/prof_t/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('kernel time %.2f ms total, %d launches (whole process: warm-up + 12 steps)' % (tot/1e6, calls))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:32]:
    print('  %-70s %7d calls %8.1f us each %8.3f ms total' % (r['Name'][:70], int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
echo done >> $F
