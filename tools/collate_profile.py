"""Developer tool: host profile of collate_device + forward on the bench scene."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402
from softgroup_amd.data import collate_device, make_item  # noqa: E402


def main():
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    item = make_item(xyz, rgb, 50, None, inst, 's')
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        for _ in range(3):
            model(collate_device([item])).resolve()
        torch.cuda.synchronize()
        for name, fn in (('collate only', lambda: collate_device([item])),
                         ('collate + forward', lambda: model(collate_device([item])).resolve())):
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            print(f'{name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms')
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            collate_device([item])
        torch.cuda.synchronize()
        pr.disable()
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in pstats.Stats(pr).stats.items():
        rows.append((ct / 10 * 1e3, tt / 10 * 1e3, nc / 10, f'{os.path.basename(fn)}:{line}:{name}'))
    rows.sort(reverse=True)
    for ct, tt, nc, nm in rows[:22]:
        print(f'{ct:9.3f} {tt:9.3f} {nc:7.1f}  {nm}')


if __name__ == '__main__':
    main()
