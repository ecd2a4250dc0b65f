#!/usr/bin/env python
"""One steady-state step of bench.py on the timeline (development aid): from a `rocprofv3 --kernel-trace --output-format csv`
run, cut the step between two `ingest` kernels near the end of the trace and print, per HIP queue, its busy time, when its
last kernel ends, and the kernels of the OTHER queue that run after the main queue has gone quiet (the weight-gradient tail).

    python tools/step_timeline.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n.split('(')[0][:60]


def main():
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    ing = [i for i, r in enumerate(rows) if 'ingest_kernel' in r['Kernel_Name']]
    a, b = ing[-3], ing[-2]                      # a step that is neither the calibration step nor the last
    step = rows[a:b]
    t0, t1 = step[0]['s'], rows[b]['s']
    print('step: %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
    byq = collections.defaultdict(list)
    for r in step:
        byq[r['Queue_Id']].append(r)
    mainq = step[0]['Queue_Id']
    ends = {}
    for q, lst in byq.items():
        busy = sum(r['e'] - r['s'] for r in lst) / 1e6
        ends[q] = max(r['e'] for r in lst)
        print('queue %s%s: %d kernels, busy %.3f ms, first start +%.3f ms, last end +%.3f ms' % (
            q, ' (main)' if q == mainq else '', len(lst), busy, (min(r['s'] for r in lst) - t0) / 1e6, (ends[q] - t0) / 1e6))
    # the main queue's last kernels and what runs on the other queue(s) meanwhile / afterwards
    main = byq[mainq]
    sgd = [r for r in main if 'sgd_kernel' in r['Kernel_Name']]
    last_chain = [r for r in main if r['s'] < (sgd[0]['s'] if sgd else t1)][-1]
    print('main chain quiet at +%.3f ms (%s); optimizer at +%.3f ms' % ((last_chain['e'] - t0) / 1e6, short(last_chain['Kernel_Name']),
                                                                       ((sgd[0]['s'] - t0) / 1e6) if sgd else -1))
    for q, lst in byq.items():
        if q == mainq:
            continue
        tail = [r for r in lst if r['e'] > last_chain['e']]
        print('queue %s after the main chain went quiet: %d kernels, %.3f ms' % (q, len(tail), sum(r['e'] - max(r['s'], last_chain['e']) for r in tail) / 1e6))
        for r in tail:
            print('   +%.3f .. +%.3f ms  %7.1f us  %s' % ((r['s'] - t0) / 1e6, (r['e'] - t0) / 1e6, (r['e'] - r['s']) / 1e3, short(r['Kernel_Name'])))
    # coarse phases on the main queue: forward ends at the first loss kernel, backward = until the chain goes quiet
    loss = [r for r in main if 'ntxent' in r['Kernel_Name']]
    if loss:
        print('forward %.3f ms, loss..end of chain %.3f ms' % ((loss[0]['s'] - t0) / 1e6, (last_chain['e'] - loss[0]['s']) / 1e6))
    # idle gaps of the main queue
    gaps = sorted(((y['s'] - x['e'], short(x['Kernel_Name']), short(y['Kernel_Name'])) for x, y in zip(main, main[1:]) if y['s'] > x['e']), reverse=True)
    print('main queue idle inside the step: %.3f ms; largest gaps:' % (sum(g[0] for g in gaps) / 1e6))
    for g in gaps[:8]:
        print('   %7.1f us between %s and %s' % (g[0] / 1e3, g[1], g[2]))
    # how much of that idle time is the ordinary kernel-to-kernel latency of one queue, and how much is the GPU waiting for the host
    for lo, hi in ((0, 3e3), (3e3, 6e3), (6e3, 12e3), (12e3, 30e3), (30e3, 1e12)):
        sel = [g[0] for g in gaps if lo <= g[0] < hi]
        print('   gaps of %4.0f .. %-6s us: %4d, %.3f ms' % (lo / 1e3, ('%.0f' % (hi / 1e3)) if hi < 1e11 else 'inf', len(sel), sum(sel) / 1e6))


if __name__ == '__main__':
    main()
