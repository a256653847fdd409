"""Static VALU instruction count per source line of one kernel, from `hipcc -S -gline-tables-only` output.
usage: asm_lines.py <file.s> <mangled kernel name> [top_n]"""
import re, sys, collections
s = open(sys.argv[1]).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
start = s.index(sys.argv[2] + ':')
body = s[start:s.index('s_endpgm', start)]
cur, cnt = None, collections.Counter()
for l in body.split('\n'):
    l = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
    elif l.startswith('v_'):
        cnt[cur] += 1
byfile = collections.Counter()
for (f, ln), c in cnt.items():
    byfile[f] += c
print('total', sum(cnt.values()), byfile.most_common(8))
for (f, ln), c in sorted(cnt.items(), key=lambda x: -x[1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print('%-24s %5d %5d' % (f, ln, c))
