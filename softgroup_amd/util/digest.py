"""Digest of what ``forward_test`` returns for one scan (reference softgroup/model/softgroup.py:323-360:
``semantic_preds``, ``offset_preds``, ``pred_instances`` with label / confidence / RLE mask).  Two runs
of a scene have equal digests iff those results are bit-identical; bench.py compares the digest of every
scan of its timed region with the digest of the same scene run one scan at a time, and
tests/test_scan_contexts_gpu.py uses ``results_equal`` for the field-by-field statement."""
import struct

import numpy as np

try:
    import xxhash

    def _hasher():
        return xxhash.xxh3_128()
except ImportError:      # same digest semantics, slower
    import hashlib

    def _hasher():
        return hashlib.blake2b(digest_size=16)

DENSE_KEYS = ('semantic_preds', 'offset_preds')


def result_digest(ret):
    h = _hasher()
    for k in DENSE_KEYS:
        if k in ret:
            a = np.ascontiguousarray(ret[k])
            h.update(k.encode() + str(a.dtype).encode() + struct.pack('<q', a.size))
            h.update(memoryview(a).cast('B'))
    insts = ret.get('pred_instances')
    if insts is not None:
        h.update(struct.pack('<q', len(insts)))
        for p in insts:
            h.update(struct.pack('<qd', int(p['label_id']), float(p['conf'])))
            m = p['pred_mask']
            h.update(m['counts'].encode() if isinstance(m, dict) else memoryview(np.ascontiguousarray(m)).cast('B'))
    pan = ret.get('panoptic_preds')
    if pan is not None:
        h.update(memoryview(np.ascontiguousarray(pan)).cast('B'))
    return h.hexdigest()


def results_equal(a, b):
    """-> (equal, first difference as text).  Every key of the two result dicts: arrays bit-identical
    (dtype, shape, bytes), instances identical in order, label, confidence and RLE mask."""
    if set(a.keys()) != set(b.keys()):
        return False, f'keys differ: {sorted(a.keys())} vs {sorted(b.keys())}'
    for k in a.keys():
        x, y = a[k], b[k]
        if k in ('pred_instances', ):
            if len(x) != len(y):
                return False, f'{k}: {len(x)} vs {len(y)} instances'
            for i, (p, q) in enumerate(zip(x, y)):
                if p['label_id'] != q['label_id'] or p['conf'] != q['conf'] or p['scan_id'] != q['scan_id']:
                    return False, f'{k}[{i}]: label/conf/scan_id {p["label_id"]},{p["conf"]} vs {q["label_id"]},{q["conf"]}'
                if p['pred_mask'] != q['pred_mask']:
                    return False, f'{k}[{i}]: RLE mask differs'
        elif isinstance(x, np.ndarray):
            if x.dtype != y.dtype or x.shape != y.shape or not np.array_equal(x, y, equal_nan=x.dtype.kind == 'f'):
                return False, f'{k}: arrays differ'
        elif x != y:
            return False, f'{k}: {x!r} vs {y!r}'
    return True, ''
