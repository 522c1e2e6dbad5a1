#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
F=$OUT/r06_wgrad.txt
: > $F
cd /tmp && export TMPDIR=/tmp
for lib in libsoftgroup_hip.so libsg_alt_wd3.so; do
  echo "== SG_LIB_NAME=$lib" >> $F
  SG_LIB_NAME=$lib python $R/tools/train_conv_bench.py 2>/dev/null | tail -9 >> $F
  SG_LIB_NAME=$lib python $R/tools/train_step_profile.py 100000 fp32 2>/dev/null | tail -1 >> $F
done
echo done >> $F
