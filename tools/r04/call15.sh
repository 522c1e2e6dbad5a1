#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "SG_CONV_WIDE=0" "SG_CONV_WIDE=1" "SG_CONV_WIDE_WV=2" "SG_CONV_WIDE_WV=4"; do
  echo "== $v: $(env $v timeout 120 python $R/tools/conv_only.py 20 2>&1 | tail -1)" >> $OUT/r04_c15_conv.txt
  env $v timeout 200 python $R/tools/conv_seq.py > $OUT/r04_c15_seq_${v//=/_}.txt 2>&1
done
cd $R && timeout 900 python -m pytest tests/test_spconv_gpu.py tests/test_unet_exec_gpu.py -x -q -m gpu 2>&1 | tail -5 > $OUT/r04_c15_tests.txt
echo done
