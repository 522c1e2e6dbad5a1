"""`python bench.py --gpus N` launches N ranks by itself (the reference's tools/dist_test.sh:1-9
pattern: torchrun, one process per GPU, OMP_NUM_THREADS=1) and reports n_gpus = N; under an
external torchrun the same code path runs.  Exercised here on CPU: gloo backend, the scan replaced
by a host delay (--stub), everything else -- launch, device-identity gather, barrier-bracketed
timed region, MAX over ranks, the JSON line -- is the real code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ, **(env or {}))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        e.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_self_spawns_n_ranks():
    out = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '6', '--warmup', '1', '--backend', 'gloo',
                '--stub'])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['steps'] == 6
    assert sorted(out['devices']) == ['cpu:0', 'cpu:1']
    # 2 ranks x 6 steps of >= 2 ms: whole-job rate below 1000 steps/s, above what one rank alone gives
    assert 0 < out['value'] < 1000 and out['ms_per_step'] >= 2.0


def test_bench_under_external_torchrun():
    out = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                '--master-addr', '127.0.0.1', '--master-port', '29561', 'bench.py', '--gpus', '2', '--steps',
                '4', '--warmup', '1', '--backend', 'gloo', '--stub'])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2


def test_bench_refuses_rank_count_mismatch():
    e = dict(os.environ)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29562', 'bench.py', '--gpus', '4',
                        '--steps', '2', '--warmup', '0', '--backend', 'gloo', '--stub'],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'launcher started 2 ranks' in (r.stdout + r.stderr)
