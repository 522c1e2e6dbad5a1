"""Developer tool: kernel durations of a rocprofv3 kernel trace grouped by (kernel, grid size) --
tells the small-level conv launches (few workgroups) from the big ones.
Usage: python tools/conv_by_grid.py <kernel_trace.csv> <n_scans|auto> [name substring]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if sys.argv[2] == 'auto':      # scans in the trace = launches of a once-per-scan kernel
        n = float(sum(1 for r in rows if 'bfs_union_kernel' in r['Kernel_Name'])) or 1.0
    else:
        n = float(sys.argv[2])
    sub = sys.argv[3] if len(sys.argv) > 3 else 'gather_conv'
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = r['Kernel_Name']
        if sub not in name:
            continue
        g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))), 1)
        a = agg[(name.split('(')[0][-48:], g)]
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f'{"kernel":48s} {"wgs":>6s} {"n/scan":>7s} {"avg us":>8s} {"ms/scan":>8s}')
    for (name, g), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{name:48s} {g:6d} {c / n:7.1f} {us / c:8.1f} {us / n / 1e3:8.3f}')


if __name__ == '__main__':
    main()
