"""Developer tool: idle gaps of the GPU between consecutive kernels of a rocprofv3 kernel trace.
Usage: python tools/gap_profile.py <kernel_trace.csv> <n_scans> [top]
Aggregates the idle time by (kernel before the gap -> kernel after it)."""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
    # skip the warm-up part: keep the last 60 % of the trace
    ev = ev[int(len(ev) * 0.4):]
    gaps = defaultdict(lambda: [0, 0.0])
    busy = 0.0
    end = ev[0][1]
    prev = ev[0][2]
    for s, e, name in ev[1:]:
        if s > end:
            g = gaps[(prev[:48], name[:48])]
            g[0] += 1
            g[1] += (s - end) / 1e3
        busy += (e - max(s, end)) / 1e3 if e > end else 0.0
        if e > end:
            end, prev = e, name
    span = (ev[-1][1] - ev[0][0]) / 1e3
    tot_gap = sum(g[1] for g in gaps.values())
    frac = 0.6
    print(f'span {span / 1e3:.2f} ms, busy {busy / 1e3:.2f} ms, idle {tot_gap / 1e3:.2f} ms '
          f'(last 60% of the trace ~ {n * frac:.1f} scans -> idle {tot_gap / 1e3 / (n * frac):.2f} ms/scan)')
    for (a, b), (cnt, us) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f'{us / (n * frac) :8.1f} us/scan  n/scan {cnt / (n * frac):5.1f}  avg {us / cnt:7.1f} us   {a}  ->  {b}')


if __name__ == '__main__':
    main()
