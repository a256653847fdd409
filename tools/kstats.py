"""Print per-kernel totals from a rocprofv3 --kernel-trace --stats output directory (kernel_stats.csv)."""
import csv, sys, glob
for fn in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(fn)))
    for r in rows:
        if float(r['TotalDurationNs']) < 2e4:
            continue
        print('%-48s calls %4s total %9.3f ms avg %9.3f ms' % (r['Name'][:48], r['Calls'], float(r['TotalDurationNs']) / 1e6,
                                                         float(r['AverageNs']) / 1e6))
