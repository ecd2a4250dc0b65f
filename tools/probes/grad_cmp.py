import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
print('loss', a['__loss'], b['__loss'])
rows = []
for k in a.files:
    if k.startswith('__'): continue
    x, y = a[k], b[k]
    e = float(np.abs(x - y).max()); s = float(np.abs(x).max())
    rows.append((e / (s + 1e-30), k, e, s))
rows.sort(reverse=True)
for r in rows[:25]:
    print('%.3e  %-60s maxdiff %.3e of %.3e' % r)
