"""Per-layer-shape timing of the sparse-conv kernel inside one forward (HIP events per launch).
Usage (GPU box): python tools/conv_layers.py [points]"""
import os
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402
from softgroup_amd.spconv import core as spcore  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=n)
    batch = synthetic.make_batch(xyz, rgb, instance_labels=inst)
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    model = synthetic.build_model(seed=0)
    with torch.no_grad():
        for _ in range(int(os.environ.get('CONV_LAYERS_WARM', '3'))):
            model(batch)
        prof = spcore.ConvProfiler()
        spcore.PROFILER = prof
        model.use_executor = False      # per-launch events are recorded on the module path
        reps = int(os.environ.get('CONV_LAYERS_REPS', '5'))
        for _ in range(reps):
            model(batch)
        torch.cuda.synchronize()
        spcore.PROFILER = None
    agg = defaultdict(lambda: [0, 0.0, 0, 0])
    for s, e, b, f, tag in prof.records:
        a = agg[tag]
        a[0] += 1
        a[1] += s.elapsed_time(e)
        a[2] += b
        a[3] += f
    print(f'{"K":>3} {"Cin":>4} {"Cout":>4} {"M_out":>7} {"n/scan":>6} {"us/launch":>9} {"ms/scan":>8} {"TF/s":>6} {"GB/s":>7}')
    tot = 0.0
    for tag, (cnt, ms, b, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tot += ms / reps
        print(f'{tag[0]:>3} {tag[1]:>4} {tag[2]:>4} {tag[3]:>7} {cnt // reps:>6} {ms / cnt * 1e3:>9.1f} '
              f'{ms / reps:>8.3f} {f / ms / 1e9:>6.2f} {b / ms / 1e6:>7.1f}')
    print('total conv ms/scan', round(tot, 3))


if __name__ == '__main__':
    main()
