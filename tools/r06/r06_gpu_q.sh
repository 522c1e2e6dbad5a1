#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_spconv_gpu.py -m gpu -x -q -s -k "bf16_rows" 2>&1 | grep -v "Warning\|warn" | tail -12 > $OUT/r06q_tests.txt
timeout 1500 python -m pytest tests/test_spconv_gpu.py tests/test_train_gpu.py tests/test_unet_train_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -6 >> $OUT/r06q_tests.txt
cd /tmp && export TMPDIR=/tmp
F=$OUT/r06q_rows16.txt
: > $F
for V in 0 1 0 1; do
  echo "== SG_UNET_ROWS16=$V train_step_bench (100 k points)" >> $F
  SG_UNET_ROWS16=$V timeout 600 python $R/tools/train_step_bench.py 2>/dev/null | grep "frozen" >> $F
done
for V in 0 1; do
  echo "== AUTOCAST=1 SG_UNET_ROWS16=$V conv_exec_layers (150 k points, backbone forward)" >> $F
  AUTOCAST=1 SG_UNET_ROWS16=$V timeout 300 python $R/tools/conv_exec_layers.py 150000 10 2>&1 | grep -v amdgpu | grep "^ 27\|total" | head -14 >> $F
done
echo done
