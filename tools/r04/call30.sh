#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
rocprofv3 --memory-copy-trace --output-format csv -d /tmp/prof -o r -- python $R/bench.py --contexts 1 --steps 10 --warmup 8 --no-cpu-baseline --no-legs --no-roofline > /dev/null 2> /tmp/prof.err
f=$(find /tmp/prof -name "*memory_copy_trace.csv" | head -1)
head -3 $f > $OUT/r04_c30_copies.txt
python - $f >> $OUT/r04_c30_copies.txt <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows),'copies total; columns',list(rows[0].keys()))
# take the last 1/3 of the trace (steady state), group by (direction,size)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)
acc=collections.Counter(); tim=collections.Counter()
for r in rows[n//2:]:
    k=(r.get('Direction'), r.get('Size') or r.get('Bytes'))
    acc[k]+=1; tim[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,c in sorted(acc.items(), key=lambda kv:-tim[kv[0]])[:60]:
    print(k, 'count', c, 'avg_us', round(tim[k]/c/1e3,1))
P
echo done
