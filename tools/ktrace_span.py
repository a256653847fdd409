"""Busy time of the fast engine per bench step from a rocprofv3 kernel trace: the union of the intervals of its kernels
(k_s_* / k_trace_*), split into steps at gaps longer than 2 ms.  With two batches in flight the kernels overlap, so the
sum of their durations exceeds the elapsed time; this union is what bench.py's HIP events bracket."""
import csv, sys, glob
iv = []
for fn in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        name = r['Kernel_Name']
        if 'k_s_' in name or 'k_trace_' in name:
            iv.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
iv.sort()
steps, cur_s, cur_e, busy = [], None, None, 0
for s, e in iv:
    if cur_s is None:
        cur_s, cur_e, step_s, busy = s, e, s, 0
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        if s - cur_e > 2_000_000:
            steps.append((cur_e - step_s, busy)); step_s, busy = s, 0
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
steps.append((cur_e - step_s, busy))
for i, (span, b) in enumerate(steps):
    print('run %d: span %.3f ms, kernels busy %.3f ms' % (i, span / 1e6, b / 1e6))
if len(sys.argv) > 2:      # bench steps (warm-up included) in the trace: back-to-back steps show up as one run
    n = int(sys.argv[2])
    print('per step (%d steps): busy %.3f ms' % (n, sum(b for _, b in steps) / n / 1e6))
