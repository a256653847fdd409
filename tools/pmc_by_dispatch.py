"""Per-dispatch view of a rocprofv3 --pmc collection: one line per k_s_* dispatch in launch order (bounce 0 and bounce 1 of a
batch are different dispatches of the same kernels), counters as columns.  usage: pmc_by_dispatch.py <dir> [last_n]"""
import csv, sys, collections, glob
disp = collections.OrderedDict()
names = []
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(fn)):
        if 'k_s_' not in row['Kernel_Name']:
            continue
        key = int(row['Dispatch_Id'])
        d = disp.setdefault(key, {'k': row['Kernel_Name'][:28]})
        d[row['Counter_Name']] = d.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
        if row['Counter_Name'] not in names:
            names.append(row['Counter_Name'])
keys = sorted(disp)
if len(sys.argv) > 2:
    keys = keys[-int(sys.argv[2]):]
print('%-28s ' % 'kernel' + ' '.join('%16s' % n[:16] for n in names))
for k in keys:
    d = disp[k]
    print('%-28s ' % d['k'] + ' '.join('%16.5g' % d.get(n, float('nan')) for n in names))
