"""Developer tool: where the HOST time of one forward goes (cProfile, no extra syncs)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        for _ in range(3):
            model(batch)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        n = 10
        pr.enable()
        for _ in range(n):
            model(batch)
        torch.cuda.synchronize()
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative')
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        rows.append((ct / n * 1e3, tt / n * 1e3, nc / n, f'{os.path.basename(fn)}:{line}:{name}'))
    rows.sort(reverse=True)
    print('cum ms/scan  self ms/scan  calls/scan  function')
    for ct, tt, nc, nm in rows[:70]:
        print(f'{ct:10.3f} {tt:12.3f} {nc:10.1f}  {nm}')


if __name__ == '__main__':
    main()
