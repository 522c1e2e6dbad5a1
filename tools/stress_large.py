"""Developer check: one forward on a scene well beyond the bench size (default 400 k points) --
arena sizing, 32-bit offsets, queue sizes.  Usage (GPU box): python tools/stress_large.py [points]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    xyz, rgb, inst = synthetic.scene_s2(seed=4, n=n)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        r = model(batch).resolve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = model(batch).resolve()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        model.use_executor = False
        r2 = model(batch).resolve()
    assert len(r['pred_instances']) == len(r2['pred_instances'])
    same = sum(a['pred_mask'] == b['pred_mask'] for a, b in zip(r['pred_instances'], r2['pred_instances']))
    print(f'{n} points, {batch["voxel_coords"].shape[0]} voxels: {ms:.1f} ms/scan, '
          f'{len(r["pred_instances"])} instances, {same} identical masks executor vs modules, '
          f'peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')


if __name__ == '__main__':
    main()
