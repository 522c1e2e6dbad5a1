"""Developer tool: scans/s against the number of scans in flight (model.scan_contexts).
Usage (GPU box): python tools/throughput_contexts.py [points] [contexts ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    ctxs = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    K = 60
    if os.environ.get('SG_SWITCH'):
        sys.setswitchinterval(float(os.environ['SG_SWITCH']))
        print('switch interval', sys.getswitchinterval())
    with torch.no_grad():
        model(batch).resolve()
        ref = model(batch)
        ref_masks = [p['pred_mask']['counts'] for p in ref['pred_instances']]
        for rep in range(3):            # interleaved repetitions: run-to-run noise shows
            for c in ctxs:
                model.scan_contexts = c
                for r in [model(batch) for _ in range(2 * c)]:
                    r.resolve()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rets = [model(batch) for _ in range(K)]
                for r in rets:
                    r.resolve()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                same = all([p['pred_mask']['counts'] for p in r['pred_instances']] == ref_masks for r in rets)
                print(f'rep {rep} contexts {c}: {dt / K * 1e3:.2f} ms/scan, {K / dt:.1f} scans/s, '
                      f'results identical: {same}', flush=True)


if __name__ == '__main__':
    main()
