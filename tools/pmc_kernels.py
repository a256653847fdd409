"""Per-kernel SQ counters of the bench workload, one batch alone on the device (TRC_STREAM_SLOTS=1), from one or more rocprofv3
--pmc passes: sums per kernel per bench step and the ratios the bench line quotes --
    valu_per_wave_cycle = SQ_INSTS_VALU / SQ_WAVE_CYCLES    (vector instructions issued per cycle a wave is resident)
    wait_frac           = SQ_WAIT_ANY / SQ_WAVE_CYCLES      (share of those cycles spent waiting for anything)
    wait_inst_frac      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (... waiting for an instruction to be issued)
usage: pmc_kernels.py <steps incl. warm-up> <rays per step> <pmc_dir> [<pmc_dir> ...] > profiles/sq_counters.json"""
import csv, sys, glob, json, collections
steps = int(sys.argv[1])
rays = int(float(sys.argv[2]))
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in sys.argv[3:]:
    for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(fn)):
            k = row['Kernel_Name'].split('(')[0]
            if not (k.startswith('k_') or k.startswith('void k_')):
                continue
            tot[k][row['Counter_Name']] += float(row['Counter_Value'])
            disp[k].add((d, row['Dispatch_Id']))
out = {'unit': 'counter sums per bench step, one batch in flight (TRC_STREAM_SLOTS=1)', 'rays_per_launch': rays, 'per_kernel': {}}
for k in sorted(tot):
    c = dict((name, v / steps) for name, v in tot[k].items())
    wc = c.get('SQ_WAVE_CYCLES', 0.)
    e = {'counters': c}
    if wc > 0:
        if 'SQ_INSTS_VALU' in c:
            e['valu_per_wave_cycle'] = c['SQ_INSTS_VALU'] / wc
        if 'SQ_WAIT_ANY' in c:
            e['wait_frac'] = c['SQ_WAIT_ANY'] / wc
        if 'SQ_WAIT_INST_ANY' in c:
            e['wait_inst_frac'] = c['SQ_WAIT_INST_ANY'] / wc
    out['per_kernel'][k] = e
print(json.dumps(out, indent=1))
