#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "SG_CONV_SPLIT=1" "SG_CONV_SPLIT=2"; do
  echo "== $v: $(env $v timeout 120 python $R/tools/conv_only.py 20 2>&1 | tail -1)" >> $OUT/r04_c17_conv.txt
done
timeout 300 python $R/tools/train_profile.py 20 plain > $OUT/r04_c17_train_plain.txt 2>&1
timeout 300 python $R/tools/train_profile.py 20 plain16 > $OUT/r04_c17_train_plain16.txt 2>&1
timeout 300 python $R/tools/train_profile.py 10 stages16 > $OUT/r04_c17_train_stages16.txt 2>&1
cd $R && timeout 900 python -m pytest tests/test_spconv_gpu.py -x -q -m gpu -k "bf16_operand or split_precision" 2>&1 | tail -5 > $OUT/r04_c17_tests.txt
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $OUT/r04_c17_tests.txt
echo done
