"""The mode bench.py's headline is measured in: ``model.scan_contexts = K`` keeps K scans in flight
(worker threads, one HIP stream each, shared module caches, per-(device, stream) arenas and conv
counter pools; softgroup_amd/model/softgroup.py `_submit_scan`).  The reference's loop runs one scan
at a time and appends the results in order (tools/test.py:145-150): every result of a pipelined run
must be BIT-identical to the same scene run alone -- dense predictions, every instance's label,
confidence and RLE string -- whatever is in flight next to it."""
import numpy as np
import pytest
import torch

from softgroup_amd import synthetic
from softgroup_amd.util.digest import result_digest, results_equal

pytestmark = pytest.mark.gpu


def _cuda(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def _scenes():
    """six distinct S2 scenes, 30 k and 150 k points alternating (the workers' arenas grow in the
    middle of a run), plus a uniform cloud in which no cluster reaches the size threshold (the
    no-proposal branch of forward_test)"""
    out = []
    for seed in range(1, 7):
        if seed % 2:
            xyz, rgb, inst = synthetic.scene_s2(seed=seed, n=30000, room_scale=0.45)
        else:
            xyz, rgb, inst = synthetic.scene_s2(seed=seed, n=150000)
        out.append(_cuda(synthetic.make_batch(xyz, rgb, instance_labels=inst, scan_id=f'scene_{seed:02d}')))
    xyz, rgb = synthetic.scene_s1(seed=9, n=20000)
    out.append(_cuda(synthetic.make_batch(xyz, rgb, scan_id='scene_empty')))
    return out


@pytest.fixture(scope='module')
def setup():
    model = synthetic.build_model(seed=0)
    scenes = _scenes()
    model.scan_contexts = 1
    with torch.no_grad():
        refs = [dict(model(b)) for b in scenes]
    assert sum(len(r['pred_instances']) > 0 for r in refs) >= 6, 'scenes must exercise grouping + refinement'
    assert len(refs[-1]['pred_instances']) == 0, 'the uniform cloud must yield no instance'
    return model, scenes, refs


def _check(results, order, refs):
    for r, i in zip(results, order):
        same, why = results_equal(dict(r), refs[i])
        assert same, f'scene {i}: {why}'
        assert result_digest(r) == result_digest(refs[i])


@pytest.mark.parametrize('contexts', [3, 5])
def test_scans_in_flight_equal_one_at_a_time(setup, contexts):
    model, scenes, refs = setup
    rng = np.random.default_rng(contexts)
    model.scan_contexts = contexts
    try:
        with torch.no_grad():
            for rep in range(5):
                order = list(rng.permutation(len(scenes))) + list(rng.permutation(len(scenes)))
                rets = [model(scenes[i]) for i in order]       # submitted back to back
                _check(rets, order, refs)
    finally:
        model.scan_contexts = 1


def test_sequential_runs_are_reproducible(setup):
    model, scenes, refs = setup
    with torch.no_grad():
        again = [dict(model(b)) for b in scenes]
    _check(again, range(len(scenes)), refs)


def test_cache_invalidation_and_weight_update_between_submissions(setup):
    """invalidate_caches() retires the scan pool (its workers' streams and arenas are released) and
    drops every derived tensor; an in-place weight update bumps the parameter versions.  Scans
    submitted before and after must see consistent state."""
    model, scenes, refs = setup
    model.scan_contexts = 3
    try:
        with torch.no_grad():
            first = [model(scenes[i]) for i in (1, 0, 3)]
            model.invalidate_caches()                       # waits for the scans in flight
            second = [model(scenes[i]) for i in (2, 6, 1)]
            _check(first, (1, 0, 3), refs)
            _check(second, (2, 6, 1), refs)
            # a weight update between submissions: results change with the weights, and equal the
            # one-at-a-time results for the SAME weights
            w = model.unet.blocks.block0.conv_branch[2].weight
            keep = w.detach().clone()
            pending = [model(scenes[i]) for i in (1, 3)]
            for r in pending:
                r.resolve()          # (an optimizer step comes after the forward it belongs to)
            w.mul_(1.25)
            third = [model(scenes[i]) for i in (1, 3, 0)]
            for r in third:
                r.resolve()
            model.scan_contexts = 1
            alone = [dict(model(scenes[i])) for i in (1, 3, 0)]
            for r, a, i in zip(third, alone, (1, 3, 0)):
                same, why = results_equal(dict(r), a)
                assert same, f'after the weight update, scene {i}: {why}'
            assert any(not results_equal(a, refs[i])[0] for a, i in zip(alone, (1, 3, 0))), \
                'the weight update must change the results'
            w.copy_(keep)
            _check(pending, (1, 3), refs)
            back = [dict(model(scenes[i])) for i in (1, 3)]
            _check(back, (1, 3), refs)
    finally:
        model.scan_contexts = 1


def test_scan_result_hook_runs_on_the_worker(setup):
    """model.scan_result_hook(result) is called by the scan worker before the result is handed back"""
    import threading
    model, scenes, refs = setup
    seen = []
    model.scan_contexts = 3
    model.scan_result_hook = lambda res: (seen.append(threading.current_thread().name),
                                          res.__setitem__('digest', result_digest(res)))
    try:
        with torch.no_grad():
            rets = [model(scenes[i]) for i in (1, 3, 5)]
            for r, i in zip(rets, (1, 3, 5)):
                assert r['digest'] == result_digest(refs[i])
    finally:
        model.scan_result_hook = None
        model.scan_contexts = 1
    assert len(seen) == 3 and all(n.startswith('softgroup-scan') for n in seen)


def test_pool_retirement_releases_stream_arenas(setup):
    from softgroup_amd.model import native_scan as NS
    from softgroup_amd.spconv import unet_exec as UE
    model, scenes, refs = setup
    model.scan_contexts = 3
    with torch.no_grad():
        for r in [model(scenes[i]) for i in (1, 3, 5, 1, 3, 5)]:
            r.resolve()
    pool = model.__dict__['_scan_pool']
    raws = [st.cuda_stream for _, st in pool._sg_streams]
    from softgroup_amd.model import scan_forward as SF
    # the one-call scan keeps ONE arena per stream (scan_forward._arenas); the staged path two kinds
    assert raws and any(k[1] in raws for k in SF._arenas)
    model.use_scan_forward = False
    with torch.no_grad():
        for r in [model(scenes[i]) for i in (1, 3, 5)]:
            r.resolve()
    model.use_scan_forward = True
    assert any(k[2] in raws for k in NS._arenas) and any(k[1] in raws for k in UE._arena)
    model.invalidate_caches()
    assert not any(k[1] in raws for k in SF._arenas)
    assert not any(k[2] in raws for k in NS._arenas) and not any(k[1] in raws for k in UE._arena)
    model.scan_contexts = 1
    with torch.no_grad():
        _check([model(scenes[1])], (1, ), refs)


def test_accumulated_results_do_not_accumulate_pinned_memory(setup):
    """result arrays are views of a pinned staging block only while few such blocks are alive"""
    from softgroup_amd.util import cast
    model, scenes, refs = setup
    keep_cap = cast._PINNED_CAP
    cast._PINNED_CAP = 3 << 20
    try:
        with torch.no_grad():
            base = cast.pinned_result_bytes()
            held = [dict(model(scenes[0])) for _ in range(8)]       # ~2.4 MB of dense results each
            assert cast.pinned_result_bytes() - base <= 3 << 20
            _check(held, [0] * 8, refs)
            del held
            import gc
            gc.collect()
            assert cast.pinned_result_bytes() <= base
    finally:
        cast._PINNED_CAP = keep_cap
