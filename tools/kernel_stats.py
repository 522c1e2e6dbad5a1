"""Print a rocprofv3 kernel_stats.csv per scan.  Usage: python tools/kernel_stats.py <csv> <n_scans|auto> [top]
(auto: the number of scans in the trace = the calls of bfs_union_kernel, launched once per scan)"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if sys.argv[2] == 'auto':
        n = float(next(int(r['Calls']) for r in rows if 'bfs_union_kernel' in r['Name']))
    else:
        n = float(sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'GPU busy {tot / 1e6 / n:.3f} ms/scan over {len(rows)} kernels')
    for r in rows[:top]:
        print(f"{r['Name'][:66]:66s} calls/scan {int(r['Calls']) / n:6.1f} avg_us "
              f"{float(r['AverageNs']) / 1e3:8.1f} ms/scan {float(r['TotalDurationNs']) / 1e6 / n:6.3f}")


if __name__ == '__main__':
    main()
