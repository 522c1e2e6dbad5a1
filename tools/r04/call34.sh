#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/tools/window_diag.py 20 4 > $OUT/r04_c34_window.txt 2>&1
python $R/tools/window_diag.py 20 3 >> $OUT/r04_c34_window.txt 2>&1
python $R/tools/window_diag.py 20 6 >> $OUT/r04_c34_window.txt 2>&1
echo done
