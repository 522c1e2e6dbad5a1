cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py $ARGS --no-legs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$ARGS $*', d['ms_per_step'], d['ms_per_step_windows'])"; }
for rep in 1 2; do
ARGS="--contexts 5" run SG_BENCH_DIGEST=main
ARGS="--contexts 4" run SG_BENCH_DIGEST=main
ARGS="--contexts 4" run SG_BENCH_DIGEST=worker
ARGS="--contexts 4 --switch-interval-us 200" run SG_BENCH_DIGEST=main
ARGS="--contexts 5 --switch-interval-us 200" run SG_BENCH_DIGEST=main
ARGS="--contexts 3" run SG_BENCH_DIGEST=main
done
