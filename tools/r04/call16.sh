#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/tools/train_profile.py 10 stages > $OUT/r04_c16_train_stages.txt 2>&1
timeout 300 python $R/tools/train_profile.py 10 host > $OUT/r04_c16_train_host.txt 2>&1
timeout 300 python $R/tools/train_profile.py 20 plain > $OUT/r04_c16_train_plain.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_c16_prof -o train -- python $R/tools/train_profile.py 10 plain > $OUT/r04_c16_rocprof.log 2>&1
python $R/tools/kernel_stats.py $(find $OUT/r04_c16_prof -name '*kernel_stats.csv' | head -1) 13 40 > $OUT/r04_c16_train_kernel_top.txt 2>&1
rm -rf $OUT/r04_c16_prof
echo done
