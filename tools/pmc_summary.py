"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name.
Usage: python tools/pmc_summary.py <dir> [kernel-substring] [--json out.json --scans N]
With --json the per-counter totals of the matching kernels are written as one JSON object
(per scan and per dispatch), which bench.py reads for roofline.traffic."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    d = args[0]
    sub = args[1] if len(args) > 1 else ''
    out_json = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
    scans = float(sys.argv[sys.argv.index('--scans') + 1]) if '--scans' in sys.argv else 1.0
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    tot = defaultdict(float)
    disp = defaultdict(int)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name']
            if sub not in k:
                continue
            name = row['Counter_Name']
            tot[name] += float(row['Counter_Value'])
            disp[name] += 1
            k = k[:60] + ' grid=' + row['Grid_Size']
            agg[k][name] += float(row['Counter_Value'])
            cnt[(k, name)] += 1
    for k, c in agg.items():
        print(k)
        for name, v in sorted(c.items()):
            print(f'   {name:32s} total {v:16.0f}  per-dispatch {v / max(cnt[(k, name)], 1):14.1f}  n={cnt[(k, name)]}')
    if out_json:
        obj = {'kernel_substring': sub, 'scans': scans,
               'counters': {n: {'total': tot[n], 'dispatches': disp[n], 'per_scan': tot[n] / scans,
                                'per_dispatch': tot[n] / max(disp[n], 1)} for n in tot}}
        try:
            old = json.load(open(out_json))
            old['counters'].update(obj['counters'])
            obj = old
        except (OSError, ValueError, KeyError):
            pass
        json.dump(obj, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main()
