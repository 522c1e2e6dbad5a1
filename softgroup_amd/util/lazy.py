"""Result dict of ``forward_test`` whose host-side post-processing (instance extraction, RLE
strings, device-to-host copies) may still be running on a worker thread.  It behaves like the
plain dict the reference returns (softgroup/model/softgroup.py:299-361): the first access of any
kind waits for the worker and merges what it produced; a worker exception is raised there."""
from concurrent.futures import ThreadPoolExecutor


class LazyResults(dict):

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._pending = None

    def defer(self, future):
        self._pending = future

    def resolve(self):
        fut, self._pending = self._pending, None
        if fut is not None:
            dict.update(self, fut.result())
        return self

    # every read goes through resolve()
    def __getitem__(self, k): return dict.__getitem__(self.resolve(), k)
    def __contains__(self, k): return dict.__contains__(self.resolve(), k)
    def __iter__(self): return dict.__iter__(self.resolve())
    def __len__(self): return dict.__len__(self.resolve())
    def __eq__(self, o): return dict.__eq__(self.resolve(), o)
    def __ne__(self, o): return dict.__ne__(self.resolve(), o)
    def __repr__(self): return dict.__repr__(self.resolve())
    def __reduce__(self): return (dict, (dict(self.resolve()), ))
    def get(self, k, d=None): return dict.get(self.resolve(), k, d)
    def keys(self): return dict.keys(self.resolve())
    def values(self): return dict.values(self.resolve())
    def items(self): return dict.items(self.resolve())
    def copy(self): return dict(self.resolve())
    def pop(self, *a): return dict.pop(self.resolve(), *a)
    def setdefault(self, *a): return dict.setdefault(self.resolve(), *a)

    __hash__ = None


_pool = None


def worker():
    """one background thread: results are finished in submission order"""
    global _pool
    if _pool is None:
        _pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='softgroup-results')
    return _pool
