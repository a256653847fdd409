"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch and dispatch count."""
import csv, sys, collections, glob
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row['Kernel_Name'][:60]
        tot[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[k][row['Counter_Name']] += 1
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        print('    %-22s sum %.4g  dispatches %d' % (c, tot[k][c], cnt[k][c]))
