"""Development aid: prints a rocprofv3 *_kernel_stats.csv compactly (kernel names contain commas)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-60s calls %6s  avg %10.1f us  total %9.2f ms  %5s%%' % (r['Name'].split('(')[0][-60:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
