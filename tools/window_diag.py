"""Developer tool: completion time of every scan of a bench-shaped timed region (20 scans submitted
at once to 4 contexts), to see where the first third of the region loses time.
Usage (GPU box): python tools/window_diag.py [steps] [contexts]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    model.scan_contexts = ctx
    with torch.no_grad():
        for r in [model(batch) for _ in range(5)]:
            r.resolve()
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rets = [model(batch) for _ in range(steps)]
            t_sub = time.perf_counter() - t0
            done = []
            for r in rets:
                r.resolve()
                done.append((time.perf_counter() - t0) * 1e3)
            torch.cuda.synchronize()
            tot = (time.perf_counter() - t0) * 1e3
            print(f'rep {rep}: submit {t_sub * 1e3:.2f} ms, total {tot:.1f} ms = {tot / steps:.2f} ms/scan; '
                  f'resolved at ' + ' '.join(f'{d:.1f}' for d in done))


if __name__ == '__main__':
    main()
