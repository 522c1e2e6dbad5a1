"""Developer tool: where does the HOST time of one scan go?  cProfile over N forward_test calls
(one scan at a time, results in line), top functions by cumulative and by own time.
Usage (GPU box): python tools/host_profile.py [scans] [scannet|stpls3d_pp|kitti]"""
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
    import copy
    import numpy as np
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    which = sys.argv[2] if len(sys.argv) > 2 else 'scannet'
    if which == 'stpls3d_pp':
        xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
        xyz = (xyz * np.float32(40)).astype(np.float32)
        batch = synthetic.make_batch(xyz, rgb, scale=3, instance_labels=inst)
        cfg = copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG)
    elif which == 'kitti':
        xyz, rgb, inst = synthetic.scene_lidar(seed=3, n=120000)
        batch = synthetic.make_batch(xyz, rgb, scale=20, instance_labels=inst)
        cfg = copy.deepcopy(synthetic.KITTI_MODEL_CFG)
    else:
        xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
        batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
        cfg = None
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(cfg, seed=0)
    model.async_results = False
    with torch.no_grad():
        for _ in range(2):
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
