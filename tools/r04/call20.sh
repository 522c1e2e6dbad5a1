#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in plain plain16; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c20_$m -o train -- python $R/tools/train_profile.py 10 $m > $OUT/r04_c20_rocprof_$m.log 2>&1
python $R/tools/kernel_stats.py $(find $OUT/c20_$m -name '*kernel_stats.csv' | head -1) 13 60 > $OUT/r04_c20_train_kernel_top_$m.txt 2>&1
rm -rf $OUT/c20_$m
done
echo done
