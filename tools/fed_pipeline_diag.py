"""Developer tool: five scans in flight fed from pinned host scenes by data.prefetch_device -- where does a scan's
3.6 ms go?  Per batch: the loader's collate wall time (host part / until its stream is done), the time the consumer
waited for the batch, the submit time.  Usage (GPU box): python tools/fed_pipeline_diag.py [loaders] [scans]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from softgroup_amd import synthetic  # noqa: E402
from softgroup_amd import data as D  # noqa: E402


def main():
    loaders = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    if len(sys.argv) > 3:
        sys.setswitchinterval(float(sys.argv[3]))
    xyz, rgb, inst = synthetic.scene_s2(seed=1, n=150000)
    item = D.make_item(xyz, rgb, 50, None, inst, 's')
    model = synthetic.build_model(seed=0)
    model.scan_contexts = 5
    stats = []
    inner = D.collate_device

    def timed_collate(batch, **kw):
        t0 = time.perf_counter()
        out = inner(batch, **kw)
        t1 = time.perf_counter()
        torch.cuda.current_stream().synchronize()
        stats.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
        return out

    with torch.no_grad():
        for r in [model(b) for b in D.prefetch_device([[item]] * 10, depth=2, workers=loaders)]:
            r.resolve()
        torch.cuda.synchronize()
        for timed in (False, True):
            stats.clear()
            waits, submits = [], []
            t0 = time.perf_counter()
            rets = []
            it = iter(D.prefetch_device([[item]] * n, collate=timed_collate if timed else None, depth=2, workers=loaders))
            while True:
                a = time.perf_counter()
                try:
                    b = next(it)
                except StopIteration:
                    break
                c = time.perf_counter()
                rets.append(model(b))
                waits.append((c - a) * 1e3)
                submits.append((time.perf_counter() - c) * 1e3)
                if len(rets) > 10:          # (a serving loop takes and releases results as it goes)
                    rets.pop(0).resolve()
            for r in rets:
                r.resolve()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            med = lambda v: sorted(v)[len(v) // 2]
            line = f'loaders {loaders}: {ms:.2f} ms/scan; consumer waited {med(waits):.2f} ms per batch (median), submit {med(submits):.2f}'
            if timed:
                line += f'; collate under load: host part {med([s[0] for s in stats]):.2f} ms, until its stream is done {med([s[1] for s in stats]):.2f} ms'
            print(line, flush=True)


if __name__ == '__main__':
    main()
