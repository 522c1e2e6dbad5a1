#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for lib in libsoftgroup_hip.so libsg_alt_d3.so libsg_alt_d4.so libsg_alt_d3w1.so libsg_alt_d2w1.so; do
  echo "== $lib: $(env SG_LIB_NAME=$lib SG_CONV_SPLIT=2 timeout 120 python $R/tools/conv_only.py 20 2>&1 | tail -1)" >> $OUT/r04_c18_conv.txt
done
echo done
