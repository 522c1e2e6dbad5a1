"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name.
Usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if sub not in k:
                continue
            k = k[:60] + ' grid=' + row['Grid_Size']
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            cnt[(k, row['Counter_Name'])] += 1
    for k, c in agg.items():
        print(k)
        for name, v in sorted(c.items()):
            print(f'   {name:32s} total {v:16.0f}  per-dispatch {v / max(cnt[(k, name)], 1):14.1f}  n={cnt[(k, name)]}')


if __name__ == '__main__':
    main()
