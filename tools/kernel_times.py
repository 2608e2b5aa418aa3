#!/usr/bin/env python3
"""Per-kernel average durations from a rocprofv3 kernel_stats.csv (or several, side by side)."""
import csv, sys
for path in sys.argv[1:]:
    print("==", path)
    for r in csv.DictReader(open(path)):
        print("  %-52s calls %5s avg %8.1f us" % (r['Name'].split('(')[0].replace('void vof::', '')[:52], r['Calls'], float(r['AverageNs']) / 1e3))
