"""Reads the stamps SG_CHAIN_TRACE=<file> makes conv_chain_kernel write (per step and workgroup: step start,
body end, arrival at the grid barrier, barrier exit; 100 MHz real-time counter) and prints, per step, where
the time goes.  Usage (GPU box):
  SG_CHAIN_TRACE=/tmp/chain.bin python tools/conv_only.py 1; python tools/chain_trace.py /tmp/chain.bin [launch index]"""
import sys

import numpy as np


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.int64)
    want = int(sys.argv[2]) if len(sys.argv) > 2 else None
    pos, li = 0, 0
    launches = []
    while pos < len(raw):
        n, grid = int(raw[pos]), int(raw[pos + 1])
        pos += 4
        steps = raw[pos:pos + 8 * n].reshape(n, 8)
        pos += 8 * n
        st = raw[pos:pos + n * grid * 4].reshape(n, grid, 4).astype(np.float64) / 100.0    # us
        pos += n * grid * 4
        launches.append((steps, st))
    print(f'{len(launches)} chain launches in the file')
    kinds = {0: 'conv', 1: 'concat', 2: 'bnrelu'}
    for li, (steps, st) in enumerate(launches):
        if want is not None and li != want:
            continue
        n, grid = st.shape[:2]
        t0 = st[0, :, 0].min()
        print(f'--- launch {li}: {n} steps, {grid} workgroups, span {st[n - 1, :, 1].max() - t0:.1f} us')
        print(f'{"step":>4} {"kind":>6} {"M":>6} {"K":>3} {"Cin":>4} {"Cout":>4} {"units":>6} {"ks":>3} | {"start":>7} '
              f'{"body mean":>9} {"body max":>8} {"last body end":>13} {"drain":>6} {"barrier after last arrival":>26} {"step total":>10}')
        for i in range(n):
            k, M, K, ci, co, units, ks, cu = [int(v) for v in steps[i]]
            s0 = st[i, :, 0]
            body = st[i, :, 1] - s0
            start = s0.min() - t0
            last_body = st[i, :, 1].max() - t0
            if i + 1 < n:
                drain = (st[i, :, 2] - st[i, :, 1]).mean()
                last_arr = st[i, :, 2].max()
                bar = st[i, :, 3].max() - last_arr
                total = st[i, :, 3].max() - s0.min()
            else:
                drain = bar = float('nan')
                total = st[i, :, 1].max() - s0.min()
            print(f'{i:>4} {kinds.get(k, "?"):>6} {M:>6} {K:>3} {ci:>4} {co:>4} {units:>6} {ks:>3} | {start:>7.1f} '
                  f'{body.mean():>9.2f} {body.max():>8.2f} {last_body:>13.1f} {drain:>6.2f} {bar:>26.2f} {total:>10.2f}')


if __name__ == '__main__':
    main()
