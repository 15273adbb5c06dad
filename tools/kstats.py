#!/usr/bin/env python
"""print name / calls / average us of the kernels of a rocprofv3 kernel_stats.csv whose name matches any of the given substrings"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if len(sys.argv) < 3 or any(s in r['Name'] for s in sys.argv[2:]):
        print('%-90s calls %5s avg %9.1f us  total %9.3f ms' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
