#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
  SG_CONV_SPLIT=$m timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/c19_$m -o t -- python $R/tools/conv_only.py 9 > /dev/null 2>&1
  f=$(find $OUT/c19_$m -name '*kernel_trace.csv' | head -1)
  python $R/tools/conv_seq.py $f 10 > $OUT/r04_c19_seq_split$m.txt 2>&1
  python - $f > $OUT/r04_c19_other_split$m.txt <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda:[0,0])
for r in rows:
    n=r['Kernel_Name'].split('(')[0][:70]
    acc[n][0]+=int(r['End_Timestamp'])-int(r['Start_Timestamp']); acc[n][1]+=1
for n,(t,c) in sorted(acc.items(), key=lambda kv:-kv[1][0])[:25]:
    print(f'{n:70s} calls/fwd {c/10:6.1f} ms/fwd {t/1e7:7.4f}')
print('total', sum(t for t,c in acc.values())/1e7)
P
  rm -rf $OUT/c19_$m
done
echo done
