"""N whole scans (model(batch), one at a time, results in line) of the bench scene and nothing else: the
workload for rocprofv3 traces of ONE scan (tools/scan_sequence.py post-processes the kernel trace).
Usage (GPU box): python tools/scan_only.py [scans] [points] [config: scannet|stpls3d_pp|kitti]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
    which = sys.argv[3] if len(sys.argv) > 3 else 'scannet'
    import copy
    import numpy as np
    if which == 'stpls3d_pp':         # BASELINE config 4 (bench.py config_legs)
        xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
        xyz = (xyz * np.float32(40)).astype(np.float32)
        batch = synthetic.make_batch(xyz, rgb, scale=3, instance_labels=inst)
        cfg = copy.deepcopy(synthetic.STPLS3D_PP_MODEL_CFG)
    elif which == 'kitti':            # BASELINE config 5
        xyz, rgb, inst = synthetic.scene_lidar(seed=3, n=min(n, 120000))
        batch = synthetic.make_batch(xyz, rgb, scale=20, instance_labels=inst)
        cfg = copy.deepcopy(synthetic.KITTI_MODEL_CFG)
    else:
        xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
        batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
        cfg = None
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(cfg, seed=0)
    model.async_results = os.environ.get('SCAN_ASYNC', '0') == '1'
    with torch.no_grad():
        for _ in range(3):
            dict(model(batch))
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = dict(model(batch))
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
    ctx = int(os.environ.get('SCAN_CONTEXTS', '0'))
    if ctx > 1:      # the same scans in flight: throughput, and every result against the one-at-a-time result
        from softgroup_amd.util.digest import result_digest
        with torch.no_grad():
            want = result_digest(dict(model(batch)))
            model.scan_contexts = ctx
            for r_ in [model(batch) for _ in range(2 * ctx)]:
                r_.resolve()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rets = [model(batch) for _ in range(reps)]
            same = [result_digest(dict(r_)) == want for r_ in rets]
            torch.cuda.synchronize()
            print(f'{ctx} scans in flight: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms/scan over {reps} scans, '
                  f'{sum(same)} of {len(same)} identical to the scan run alone ({which})')
    ts.sort()
    print(f'{reps} scans, latency ms min {ts[0]:.3f} median {ts[len(ts) // 2]:.3f} max {ts[-1]:.3f}; '
          f'{len(r.get("pred_instances", []))} instances ({which})')


if __name__ == '__main__':
    main()
