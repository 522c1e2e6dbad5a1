#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $OUT/r04_c27_pytest.txt
PMC=1 bash tools/profile_round.sh r04 > $OUT/r04_c27_profile.log 2>&1
echo done
