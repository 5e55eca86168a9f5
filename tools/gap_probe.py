#!/usr/bin/env python3
"""Inter-kernel gaps of the timed configuration from a rocprofv3 kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline --no-sustained
   python tools/gap_probe.py /tmp/gp"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    for k in ('pfb_kernel', 'fir_small', 'copyBuffer', 'copy8', 'fillBuffer'):
        if k in n: return k
    return n.split('(')[0][-30:]
seq = [(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
w = seq[-600:-100]
gaps, durs, prev = collections.defaultdict(list), collections.defaultdict(list), None
for name, s, e in w:
    durs[name].append((e - s) / 1e3)
    if prev: gaps[(prev[0], name)].append((s - prev[2]) / 1e3)
    prev = (name, s, e)
for k, v in durs.items(): print('dur  %-28s n=%4d  %8.2f us' % (k, len(v), sum(v) / len(v)))
for k, v in gaps.items(): print('gap  %-28s n=%4d  %8.2f us' % ('%s -> %s' % k, len(v), sum(v) / len(v)))
n = len(durs.get('pfb_kernel', [1]))
print('period per step %.2f us' % ((w[-1][2] - w[0][1]) / 1e3 / n))
