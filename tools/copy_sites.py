"""Developer tool: which Python lines issue copies / fills / casts in one forward (TorchDispatchMode
+ the Python stack).  `python tools/copy_sites.py all`: every aten op that launches something."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402

WATCH = ('copy_', '_to_copy', 'fill_', 'zero_', 'zeros', 'full', 'item', '_local_scalar_dense', 'nonzero',
         'clone', 'contiguous', 'cat', 'index_put_', 'empty_like', 'ones', 'arange')


ALL = len(sys.argv) > 1 and sys.argv[1] == 'all'
VIEWS = ('view', 'reshape', 't', 'expand', 'select', 'slice', 'unsqueeze', 'squeeze', 'as_strided', 'detach', 'alias',
         '_unsafe_view', 'permute', 'transpose', 'empty', 'empty_like', 'empty_strided', 'new_empty', 'sym_size',
         'size', 'stride', 'numel', 'dim', 'is_pinned', 'lift_fresh', 'unbind', 'split', 'narrow', 'unfold',
         'resize_', 'set_', '_reshape_alias', 'view_as', 'expand_as')


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.cnt = collections.Counter()
        self.bytes = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        if (ALL and name not in VIEWS) or name in WATCH:
            where = '?'
            for f in reversed(traceback.extract_stack()):
                if 'softgroup_amd/' in f.filename and 'tools/' not in f.filename:
                    where = f'{f.filename.split("softgroup_amd/")[-1]}:{f.lineno}'
                    break
            self.cnt[(where, name)] += 1
            out = func(*args, **(kwargs or {}))
            if isinstance(out, torch.Tensor):
                self.bytes[(where, name)] += out.numel() * out.element_size()
            return out
        return func(*args, **(kwargs or {}))


def main():
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    model.async_results = False
    with torch.no_grad():
        for _ in range(3):
            model(batch)
        torch.cuda.synchronize()
        spy = Spy()
        with spy:
            model(batch)
            torch.cuda.synchronize()
    print(sum(spy.cnt.values()), 'ops')
    for (where, n), c in sorted(spy.cnt.items()):
        print(f'{c:3d} {n:22s} {spy.bytes[(where, n)] / 1e6:9.2f} MB  {where}')


if __name__ == '__main__':
    main()
