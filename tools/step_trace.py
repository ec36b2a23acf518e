#!/usr/bin/env python3
"""Every launch of ONE training step, in issue order, from a rocprofv3 kernel trace CSV (cut at the sampler launches):
start offset, duration, grid, LDS, kernel name.  Usage: step_trace.py kernel_trace.csv [step_index=3] [out.md]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:64]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cuts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void noise_kernel')]
    seg = rows[cuts[idx]:cuts[idx + 1]]
    t0 = int(seg[0]['Start_Timestamp'])
    lines = ['| # | start ms | us | gap us | grid | lds | kernel |', '|---|---|---|---|---|---|---|']
    prev_end = t0
    for i, r in enumerate(seg):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        grid = r.get('Grid_Size_X', r.get('Grid_Size', '?'))
        wg = r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))
        lines.append('| %d | %.3f | %.1f | %.1f | %s/%s | %s | `%s` |' % (i, (s - t0) / 1e6, (e - s) / 1e3, (s - prev_end) / 1e3, grid, wg, r.get('LDS_Block_Size', '?'), short(r['Kernel_Name'])))
        prev_end = max(prev_end, e)
    out = '\n'.join(lines)
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(out + '\n')


if __name__ == '__main__':
    main()
