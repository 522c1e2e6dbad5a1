"""Developer tool: the 300 k-point dense scene (bench leg `dense_scene`), N forwards one at a time;
run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        for _ in range(3):
            model(batch).resolve()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            r = model(batch)
            r.resolve()
        torch.cuda.synchronize()
        print(f'{n} points: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms/scan, '
              f'{len(r["pred_instances"])} instances', flush=True)


if __name__ == '__main__':
    main()
