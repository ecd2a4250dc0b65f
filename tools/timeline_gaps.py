#!/usr/bin/env python
"""Where does a step's wall time go?  From a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: per HIP queue
the busy time and the idle gaps inside the steady-state window (last third of the trace), and the kernels that precede the
longest gaps on the busiest queue (development aid).   python tools/timeline_gaps.py <dir with *_kernel_trace.csv> [steps]"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    t0, t1 = rows[0]['s'], rows[-1]['e']
    w0 = t0 + (t1 - t0) * 2 // 3
    win = [r for r in rows if r['s'] >= w0]
    span = (win[-1]['e'] - win[0]['s']) / 1e6
    print('window %.2f ms, %d kernels' % (span, len(win)))
    byq = collections.defaultdict(list)
    for r in win:
        byq[r['Queue_Id']].append(r)
    union = []
    for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r['e'] - r['s'] for r in lst) / 1e6
        gaps = []
        for a, b in zip(lst, lst[1:]):
            g = b['s'] - a['e']
            if g > 0:
                gaps.append((g, a['Kernel_Name'][:70], b['Kernel_Name'][:70]))
        print('queue %s: %d kernels, busy %.2f ms (%.0f%% of window), gaps total %.2f ms, gaps > 5 us: %d (%.2f ms)' % (
            q, len(lst), busy, 100 * busy / span, sum(g[0] for g in gaps) / 1e6, sum(1 for g in gaps if g[0] > 5000),
            sum(g[0] for g in gaps if g[0] > 5000) / 1e6))
        union += [(r['s'], r['e']) for r in lst]
        if len(lst) > 100:
            hist = collections.Counter()
            for g, a, b in gaps:
                hist[(a.split('<')[0].split('(')[0][-40:], )] += g
            for (a,), g in hist.most_common(8):
                print('     gap time after %-42s %.3f ms' % (a, g / 1e6))
    # the wait for the side stream at the end of each backward pass: from the end of the last main-queue kernel before the SGD
    # launch to the end of the last side-queue kernel
    qs = sorted(byq, key=lambda q: -len(byq[q]))
    if len(qs) >= 2:
        mainq, sideq = byq[qs[0]], byq[qs[1]]
        tails = []
        for i, r in enumerate(mainq):
            if 'sgd_kernel' in r['Kernel_Name'] and i > 0:
                pm = mainq[i - 1]['e']
                ps = max([x['e'] for x in sideq if x['e'] <= r['s']] or [pm])
                side_busy = sum(min(x['e'], ps) - max(x['s'], pm) for x in sideq if x['e'] > pm and x['s'] < ps)
                tails.append(((ps - pm) / 1e3, side_busy / 1e3, (r['s'] - pm) / 1e3))
        if tails:
            print('end-of-backward wait per step (us): side tail after the last main kernel / side busy in it / main idle before SGD')
            for t in tails[-6:]:
                print('    %8.1f %8.1f %8.1f' % t)
    union.sort()
    cur_s, cur_e, tot = union[0][0], union[0][1], 0
    for s, e in union[1:]:
        if s > cur_e:
            tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    tot += cur_e - cur_s
    print('GPU busy (union over queues): %.2f ms = %.0f%% of the window' % (tot / 1e6, 100 * tot / 1e6 / span))


if __name__ == '__main__':
    main()
