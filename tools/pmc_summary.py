#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv per kernel: sum of each counter + derived ratios."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    rows[k][r['Counter_Name']] += float(r['Counter_Value'])
    d = (r['Dispatch_Id'])
    if d not in seen:
        seen.add(d); calls[k] += 1; dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
names = sorted({c for v in rows.values() for c in v})
print('| kernel | calls | ms | ' + ' | '.join(names) + ' |')
print('|---|---|---|' + '---|' * len(names))
for k in sorted(rows, key=lambda k: -dur[k])[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print('| `%s` | %d | %.2f | ' % (k, calls[k], dur[k]) + ' | '.join('%.4g' % rows[k][c] for c in names) + ' |')
