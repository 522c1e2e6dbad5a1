#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd $R
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/r06_final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r06_final_smoke.txt 2>&1
PMC=1 bash tools/profile_round.sh r06
python bench.py --steps 20 --warmup 5 > $OUT/r06_bench_driver_shape.json 2> /dev/null
echo done
