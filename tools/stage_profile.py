"""Host+device time of sub-stages (cuda-synchronised wall clock) for one forward.
Usage (GPU box): python tools/stage_profile.py"""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import ops, synthetic  # noqa: E402
from softgroup_amd.model import softgroup as sgm  # noqa: E402
from softgroup_amd.ops import functions as F  # noqa: E402
from softgroup_amd.spconv import core  # noqa: E402
from softgroup_amd.util import rle  # noqa: E402

ACC = defaultdict(float)
CNT = defaultdict(int)


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        ACC[name] += time.perf_counter() - t0
        CNT[name] += 1
        return r
    return wrapper


def main():
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        for _ in range(3):
            model(batch)
    # wrap
    core._Plan.__init__ = timed('plan_build', core._Plan.__init__)
    core.SubMRule.__init__ = timed('subm_rule(incl plan)', core.SubMRule.__init__)
    core.DownRule.__init__ = timed('down_rule(incl plan)', core.DownRule.__init__)
    core.gather_conv = timed('gather_conv', core.gather_conv)
    F.BallQueryBatchP.forward = staticmethod(timed('ballquery', F.BallQueryBatchP.forward))
    ops.bfs_cluster_segments = timed('bfs_cluster_segments', ops.bfs_cluster_segments)
    sgm.ops.bfs_cluster_segments = ops.bfs_cluster_segments
    sgm._runs_of_pairs = timed('runs_of_pairs', sgm._runs_of_pairs)
    sgm.rle_encode_many = timed('rle_encode_many', sgm.rle_encode_many)
    m = model
    for name in ['forward_backbone', 'forward_grouping', 'clusters_voxelization', 'forward_instance',
                 'get_instances', 'get_point_wise_results', 'get_gt_instances']:
        setattr(m, name, timed(name, getattr(m, name)))
    reps = 5
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model(batch)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) / reps
    print(f'total (with sync instrumentation) {tot * 1e3:.2f} ms/scan')
    for k, v in sorted(ACC.items(), key=lambda kv: -kv[1]):
        print(f'  {k:28s} {v / reps * 1e3:8.3f} ms/scan   calls/scan {CNT[k] / reps:.0f}')


if __name__ == '__main__':
    main()
