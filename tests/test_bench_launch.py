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


def test_bench_eight_ranks_on_gloo():
    """the driver's 8-GPU launch shape on CPU: 8 ranks, 8 distinct devices, ONE JSON line, the host
    thread plan of all ranks within half the cores, the single-GPU legs reported as skipped"""
    out = _run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '4', '--warmup', '1', '--backend', 'gloo',
                '--stub'])
    assert out['n_gpus'] == 8 and out['ranks_seen'] == 8 and len(set(out['devices'])) == 8
    ht = out['host_threads']
    assert ht['scan_threads'] >= 1 and ht['per_rank'] == ht['scan_threads'] + 2
    assert ht['all_ranks'] == 8 * ht['per_rank']
    assert ht['all_ranks'] <= max(ht['host_cores'] // 2, 8 * 3)
    assert out['legs'] == 'skipped (N>1)'


def test_bench_exits_nonzero_when_a_rank_fails_before_the_first_barrier():
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        e.pop(k, None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '4', '--steps', '2', '--warmup', '0', '--backend',
                        'gloo', '--stub', '--stub-fail-rank', '2'], cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0, r.stdout + r.stderr
    assert 'injected failure before the first barrier' in (r.stdout + r.stderr)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')], 'no result line from a failed job'


def test_host_thread_plan_caps_contexts():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cores = os.cpu_count() or 1
    used, plan = bench.host_thread_plan(4, 1)
    assert used == min(4, max(3, cores // 2) - 2) and plan['all_ranks'] == used + 2
    used8, plan8 = bench.host_thread_plan(4, 8)
    assert 1 <= used8 <= 4 and plan8['all_ranks'] == 8 * (used8 + 2)
    assert used8 == max(1, min(4, max(3, (cores // 2) // 8) - 2))


def test_n_gt_1_line_carries_the_ddp_leg_and_the_rank_core_slices():
    """N > 1: the line reports the gradient all-reduce of the trainable heads (2 919 208 bytes, what
    DistributedDataParallel exchanges per step with the fine-tune configs' frozen backbone; reference
    tools/train.py:172-174) measured on the job's own process group, per-rank DDP training step times,
    and the slice of host cores every rank bound itself to (disjoint slices; pinned result staging of
    a rank bounded by 12 MB x scan threads x 2)."""
    out = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1', '--backend', 'gloo',
                '--stub'])
    d = out['ddp']
    assert d['allreduce_bytes'] == 2919208 and d['backend'] == 'gloo'
    assert 0 < d['allreduce_ms_min'] <= d['allreduce_ms_median']
    assert len(d['train_ms_per_step_per_rank']) == 2 and d['train_ms_per_step_min'] <= d['train_ms_per_step_max']
    assert d['trainable_bytes'] == 2919208
    ht = out['host_threads']
    assert ht['pinned_staging_mb_per_rank_bound'] == 256 + 16        # the cap of util/cast.py + the block being filled
    slices = out['rank_core_slices']
    assert len(slices) == 2
    if hasattr(os, 'sched_setaffinity') and all(s is not None for s in slices):
        (a0, a1), (b0, b1) = slices
        assert a1 < b0 or b1 < a0, slices              # disjoint


def test_n_gt_1_line_survives_a_rank_that_never_joins_the_ddp_leg():
    """the headline is complete before the DDP leg starts; a leg whose collectives never complete (here:
    rank 1 never joins) must not cost the job its line: after the deadline rank 0 prints the line with the
    reason in `ddp.error`, every rank leaves, the launcher returns 0"""
    import time
    t0 = time.time()
    out = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1', '--backend', 'gloo',
                '--stub', '--stub-ddp-hang-rank', '1'], env={'SG_BENCH_DDP_DEADLINE_S': '8'})
    assert time.time() - t0 < 120
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['value'] > 0
    assert 'did not finish within 8 s' in out['ddp']['error']
    assert out['legs'] == 'skipped (N>1)'
