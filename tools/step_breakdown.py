#!/usr/bin/env python3
"""Per-kernel time of ONE training step from a rocprofv3 kernel trace CSV: the trace is cut at the sampler launches
(one noise_kernel per step) and the chosen step's launches are summed by kernel name.
Usage: step_breakdown.py kernel_trace.csv [step_index=3] [out.md]"""
import csv
import re
import sys
from collections import OrderedDict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cuts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('void noise_kernel')]
    lo, hi = cuts[idx], cuts[idx + 1]
    seg = rows[lo:hi]
    agg = OrderedDict()
    for r in seg:
        k = short(r['Kernel_Name'])
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        c = agg.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += d
    tot = sum(v[1] for v in agg.values())
    span = int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])
    lines = ['step %d: %d launches, kernel time %.3f ms, span %.3f ms' % (idx, len(seg), tot / 1e6, span / 1e6), '',
             '| kernel | launches | total ms | avg us | % |', '|---|---|---|---|---|']
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('| `%s` | %d | %.3f | %.1f | %.2f |' % (k, c, d / 1e6, d / c / 1e3, 100.0 * d / tot))
    out = '\n'.join(lines)
    print(out)
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write(out + '\n')


if __name__ == '__main__':
    main()
