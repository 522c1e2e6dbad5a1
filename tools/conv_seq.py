"""Developer tool: the conv launches of one backbone forward in launch order, averaged over the
forwards of a rocprofv3 kernel trace of tools/conv_only.py (every forward issues the same sequence).
Usage: python tools/conv_seq.py <kernel_trace.csv> <forwards incl. warm-up>"""
import csv
import sys


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gather_conv' in r['Kernel_Name'] or 'conv_reduce' in r['Kernel_Name']]
    n_fwd = int(sys.argv[2])
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    per = len(rows) // n_fwd
    assert per * n_fwd == len(rows), (len(rows), n_fwd)
    tot = 0.0
    print(f'{"#":>3s} {"kernel":40s} {"wgs":>6s} {"avg us":>8s}')
    for i in range(per):
        sel = rows[per + i::per] if n_fwd > 1 else rows[i::per]          # skip the first (cold) forward
        us = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in sel) / len(sel) / 1e3
        r = sel[0]
        g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))), 1)
        name = r['Kernel_Name'].split('(')[0].replace('void sg::', '').replace('gather_conv_', '')[:40]
        tot += us
        print(f'{i:3d} {name:40s} {g:6d} {us:8.1f}')
    print(f'total {tot / 1e3:.3f} ms per forward over {per} launches')


if __name__ == '__main__':
    main()
