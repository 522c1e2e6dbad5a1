#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $OUT/r04_c9_pytest.txt 2>&1
echo "pytest rc $?" >> $OUT/r04_c9_pytest.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $OUT/r04_c9_bench.json 2> $OUT/r04_c9_bench.err
echo done
