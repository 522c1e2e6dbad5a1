cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for c in 2 3 4 5; do python bench.py --steps 20 --warmup 5 --contexts $c --no-legs --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ctx $c', d['ms_per_step'], d['ms_per_step_windows'])"; done; done
