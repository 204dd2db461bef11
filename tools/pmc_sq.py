#!/usr/bin/env python3
"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass: pmc_sq.py <counter_collection.csv>"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('st2::', '')
    acc[name][r['Counter_Name']] += float(r['Counter_Value'])
    calls[(name, r['Counter_Name'])] += 1
for name in sorted(acc):
    c = acc[name]
    n = max(calls[(name, k)] for k in c)
    print('%s  (n=%d)' % (name[:60], n))
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    for k in sorted(c):
        print('    %-28s %16.0f per launch %s' % (k, c[k] / n, ('= %5.1f %% of SQ_WAVE_CYCLES' % (100 * c[k] / wc)) if wc and k != 'SQ_WAVE_CYCLES' else ''))
