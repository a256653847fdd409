"""Timeline of the last N fast-engine dispatches of a rocprofv3 kernel trace (start relative to the first shown, duration, grid).
usage: ktrace_tail.py <dir> [N]"""
import csv, glob, sys
rows = []
for fn in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows = [r for r in rows if 'k_s_' in r['Kernel_Name'] or 'k_trace' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-36s q%-2s start %8.3f dur %6.3f ms  grid %7s wg %4s lds %6s' % (r['Kernel_Name'][:36], r['Queue_Id'], (s - t0) / 1e6, (e - s) / 1e6,
          r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', r.get('Workgroup_Size')), r.get('LDS_Block_Size')))
