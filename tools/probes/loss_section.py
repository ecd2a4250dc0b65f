#!/usr/bin/env python
"""probe: the kernels (and gaps) between the last forward-plan launch and the first backward-plan launch of a steady-state step"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ing = [i for i, r in enumerate(rows) if 'ingest_kernel' in r['Kernel_Name']]
step = rows[ing[-3]:ing[-2]]
q0 = step[0]['Queue_Id']
main = [r for r in step if r['Queue_Id'] == q0]
# the loss section: from the global average pool of the forward to the avgpool backward (rowscale<..,2>)
names = [re.sub(r'\(anonymous namespace\)::|^void ', '', r['Kernel_Name']).split('(')[0][:70] for r in main]
a = max(i for i, n in enumerate(names) if 'spatial_mean' in n)
b = min(i for i, n in enumerate(names) if i > a and 'rowscale_kernel<float, 2>' in n)
t0 = main[a]['e']
print('loss section: %d kernels, %.1f us wall, %.1f us of kernels' % (b - a - 1, (main[b]['s'] - t0) / 1e3, sum(r['e'] - r['s'] for r in main[a + 1:b]) / 1e3))
prev = main[a]
for r, n in zip(main[a + 1:b + 1], names[a + 1:b + 1]):
    print('  gap %6.1f us   %6.1f us  %s' % ((r['s'] - prev['e']) / 1e3, (r['e'] - r['s']) / 1e3, n))
    prev = r
