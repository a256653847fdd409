"""HBM traffic per bench step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units).  FETCH_SIZE is doubled per the
gfx950 correction of the micro-architecture guide (it counts 64 B per 128-B read request of a wide coalesced stream); for
scattered 64-byte gathers the count is already exact (tools/ubench/fetch64.hip), so the uncorrected sum is kept beside it as
the lower bound.  usage: pmc_traffic.py <fetch_dir> <write_dir> <steps incl. warm-up> <rays per step> <accel 0|1>"""
import csv, sys, glob, json, collections
def total(d, name):
    tot = collections.defaultdict(float)
    for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(fn)):
            if row['Counter_Name'] == name:
                tot[row['Kernel_Name'].split('(')[0]] += float(row['Counter_Value'])
    return tot
steps = int(sys.argv[3])
rays = int(float(sys.argv[4])) if len(sys.argv) > 4 else 100000000
accel = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
f, w = total(sys.argv[1], 'FETCH_SIZE'), total(sys.argv[2], 'WRITE_SIZE')
out = {'unit': 'bytes per bench step (all kernels of the fast engine)', 'rays_per_launch': rays, 'accel': accel, 'per_kernel': {}}
tf = tw = 0.0
for k in sorted(set(f) | set(w)):
    if not (k.startswith('k_') or k.startswith('void k_')):
        continue
    fb, wb = f.get(k, 0.) * 1024. / steps, w.get(k, 0.) * 1024. / steps
    out['per_kernel'][k] = {'fetch_bytes_as_counted': fb, 'write_bytes': wb}
    tf += fb; tw += wb
out['fetch_bytes_as_counted'] = tf; out['write_bytes'] = tw
out['hbm_bytes_per_launch'] = 2.0 * tf + tw                     # the guide's correction: upper bound
out['hbm_bytes_per_launch_fetch_as_counted'] = tf + tw         # lower bound
print(json.dumps(out, indent=1))
