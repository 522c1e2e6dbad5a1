"""LazyResults: the dict forward_test returns while result formatting may still be running."""
import pickle
from concurrent.futures import ThreadPoolExecutor

import pytest

from softgroup_amd.util.lazy import LazyResults, worker


def test_resolves_on_any_read_and_behaves_like_a_dict():
    with ThreadPoolExecutor(1) as pool:
        r = LazyResults(scan_id='s')
        r.defer(pool.submit(lambda: dict(pred_instances=[1, 2], semantic_preds='x')))
        assert dict.__len__(r) == 1                      # nothing merged yet
        assert r['pred_instances'] == [1, 2]             # first read waits and merges
        assert set(r) == {'scan_id', 'pred_instances', 'semantic_preds'}
        assert 'semantic_preds' in r and r.get('nope', 7) == 7 and len(r) == 3
        assert isinstance(r, dict) and r == dict(scan_id='s', pred_instances=[1, 2], semantic_preds='x')
        assert pickle.loads(pickle.dumps(r)) == dict(r)


def test_worker_exception_surfaces_on_access():
    def boom():
        raise ValueError('formatting failed')
    r = LazyResults(scan_id='s')
    r.defer(worker().submit(boom))
    with pytest.raises(ValueError, match='formatting failed'):
        r['pred_instances']


def test_plain_use_without_deferral():
    r = LazyResults(a=1)
    r.update(b=2)
    assert r.resolve() is r and r == {'a': 1, 'b': 2} and r.pop('a') == 1
