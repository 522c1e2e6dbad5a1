"""Developer tool: where does the HOST time of one scan go?  cProfile over N forward_test calls
(one scan at a time, results in line), top functions by cumulative and by own time.
Usage (GPU box): python tools/host_profile.py [scans]"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    model.async_results = False
    with torch.no_grad():
        for _ in range(5):
            model(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            model(batch)
        pr.disable()
        torch.cuda.synchronize()
        print(f'{(time.perf_counter() - t0) / n * 1e3:.3f} ms/scan under the profiler')
    for key in ('cumulative', 'tottime'):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])


if __name__ == '__main__':
    main()
