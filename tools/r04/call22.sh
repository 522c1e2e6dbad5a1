#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 600 python tools/train_exec_diag.py > $OUT/r04_c22_diag.txt 2>&1
echo "---- COMBINE=0" >> $OUT/r04_c22_diag.txt
SG_CONV_COMBINE=0 timeout 600 python tools/train_exec_diag.py >> $OUT/r04_c22_diag.txt 2>&1
echo done
