cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', d['ms_per_step'], d['ms_per_step_windows'], d['timed_results_identical'])"; }
run A=1; run A=1; run A=1; run A=1; run A=1; run A=1
