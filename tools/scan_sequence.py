"""Post-process a rocprofv3 kernel trace (+ memory-copy trace) of tools/scan_only.py: the launch sequence of
ONE scan from the middle of the run (start, duration, gap to the previous operation) and the per-kernel table
averaged over the middle scans.  A scan is delimited by its first kernel (voxelize_fp_kernel, launched once
per scan; another marker kernel as third argument).  Usage: python tools/scan_sequence.py <trace dir> <out prefix> [marker]"""
import collections
import csv
import glob
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    marker = sys.argv[3] if len(sys.argv) > 3 else 'voxelize_fp_kernel'
    ops = []
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                        f"{r['Grid_Size_X']}x{r['Workgroup_Size_X']}"))
    for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            ops.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '?'), ''))
    ops.sort()
    marks = [i for i, o in enumerate(ops) if marker in o[2]]
    n = len(marks)
    a, b = marks[n // 2], marks[n // 2 + 1]
    with open(out + '_sequence.txt', 'w') as fo:
        t0, prev = ops[a][0], None
        busy = 0
        for s, e, name, grid in ops[a:b]:
            gap = (s - prev) / 1e3 if prev else 0.0
            fo.write(f'{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {grid:>12}  {name[:100]}\n')
            prev = max(prev or 0, e)
            busy += e - s
        fo.write(f'# {b - a} operations, busy {busy / 1e6:.3f} ms, span {(ops[b][0] - t0) / 1e6:.3f} ms\n')
    lo, hi = marks[n // 4], marks[3 * n // 4]
    scans = 3 * n // 4 - n // 4
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, name, grid in ops[lo:hi]:
        k = name.split('(')[0][:80]
        agg[k][0] += 1
        agg[k][1] += e - s
    tot = sum(v[1] for v in agg.values())
    with open(out + '_top.txt', 'w') as fo:
        fo.write(f'GPU busy {tot / 1e6 / scans:.3f} ms/scan, {sum(v[0] for v in agg.values()) / scans:.1f} operations/scan '
                 f'(scans {n // 4}..{3 * n // 4} of {n})\n')
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            fo.write(f'{k:80s} calls/scan {c / scans:6.1f} avg_us {t / c / 1e3:8.1f} ms/scan {t / 1e6 / scans:6.3f}\n')


if __name__ == '__main__':
    main()
