"""Developer tool: host profile (cProfile) of SoftGroup.get_instances on the bench scene.
Usage (GPU box): python tools/instances_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import parity  # noqa: E402  (gpu_stages helper only)
from softgroup_amd import synthetic  # noqa: E402


def main():
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    model = synthetic.build_model(seed=0)
    g = parity.gpu_stages(model, batch)
    args = ('s', g['pidx'], g['sem'], g['cls'], g['iou'], g['mask'])
    with torch.no_grad():
        for _ in range(3):
            model.get_instances(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(n):
            model.get_instances(*args)
        pr.disable()
        print(f'get_instances: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call')
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in pstats.Stats(pr).stats.items():
        rows.append((ct / n * 1e3, tt / n * 1e3, nc / n, f'{os.path.basename(fn)}:{line}:{name}'))
    rows.sort(reverse=True)
    for ct, tt, nc, nm in rows[:28]:
        print(f'{ct:9.3f} {tt:9.3f} {nc:7.1f}  {nm}')


if __name__ == '__main__':
    main()
