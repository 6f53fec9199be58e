"""Two-stream view of a rocprofv3 rocpd database of `bench.py`: per-stream busy time, their union and the main stream's gaps over the
shortest full step (steps are delimited by the gradient-norm kernel that precedes the parameter update). Usage: python tools/rocpd_overlap.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(db.execute(f"select d.start, d.end, d.stream_id, d.queue_id, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
ad = [r for r in rows if 'sqnorm_kernel' in r[4]]          # one per step, right behind the backward pass (the update itself may be chunked / on the second stream)
spans = [(ad[i + 1][0] - ad[i][0], i) for i in range(len(ad) - 1)]
print('steps (ms between gradient-norm kernels):', ' '.join('%.2f' % (d / 1e6) for d, _ in spans), '-> the shortest one below (the host, slowed by the profiler, starves some)')
_, i = min(spans)
t0, t1 = ad[i][0], ad[i + 1][0]
R = [r for r in rows if r[0] >= t0 and r[1] <= t1]


def busy(rs):
    ev = sorted((r[0], r[1]) for r in rs)
    if not ev:
        return 0.0
    tot, (cs, ce) = 0, ev[0]
    for s, e in ev[1:]:
        if s > ce:
            tot += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    return (tot + ce - cs) / 1e6


streams = sorted({(r[2], r[3]) for r in R})
print('step: %.2f ms between gradient-norm kernels; %d kernels' % ((t1 - t0) / 1e6, len(R)))
for st, q in streams:
    rs = [r for r in R if r[2] == st]
    print('  stream %d (hw queue %d): %5d kernels, busy %.2f ms, sum of durations %.2f ms' % (st, q, len(rs), busy(rs), sum(r[1] - r[0] for r in rs) / 1e6))
print('  union of all streams busy %.2f ms; sum of all durations %.2f ms' % (busy(R), sum(r[1] - r[0] for r in R) / 1e6))
main = max(streams, key=lambda s: len([r for r in R if r[2] == s[0]]))[0]
ev = sorted((r[0], r[1]) for r in R if r[2] == main)
gaps = [s1 - e0 for (s0, e0), (s1, e1) in zip(ev, ev[1:]) if s1 - e0 > 5000]
print('  main-stream gaps > 5 us: %d, total %.2f ms' % (len(gaps), sum(gaps) / 1e6))
if '--gaps' in sys.argv:
    # where the main stream stalls: its largest gaps with the kernels on either side, and a histogram by (previous, next) kernel
    import re
    short = lambda n: re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', n)[:40]
    mr = sorted((r for r in R if r[2] == main), key=lambda r: r[0])
    gl = sorted(((b[0] - a[1], short(a[4]), short(b[4])) for a, b in zip(mr, mr[1:]) if b[0] - a[1] > 3000), reverse=True)
    hist = {}
    for g, pa, nb in gl:
        h = hist.setdefault((pa, nb), [0, 0]); h[0] += 1; h[1] += g
    print('  gaps > 3 us on the main stream: %d, total %.2f ms; by (previous kernel -> next kernel):' % (len(gl), sum(g for g, _, _ in gl) / 1e6))
    for (pa, nb), (n, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
        print('    %4d x  avg %6.1f us  total %7.1f us   %s -> %s' % (n, tot / n / 1e3, tot / 1e3, pa, nb))
if '--tail' in sys.argv:
    # the end of the backward pass: every kernel of the last 2.5 ms before the gradient-norm kernel, both streams
    import re
    short = lambda n: re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', n)[:46]
    print('  last 2.5 ms of the step (us relative to the gradient-norm kernel; stream, start, end, kernel):')
    for r in R:
        if r[1] > t1 - 2500000:
            print('    s%d %9.1f %9.1f  %s' % (r[2], (r[0] - t1) / 1e3, (r[1] - t1) / 1e3, short(r[4])))
    nxt = [r for r in rows if r[0] >= t1][:40]
    print('  first 40 kernels of the next step:')
    for r in nxt:
        print('    s%d %9.1f %9.1f  %s' % (r[2], (r[0] - t1) / 1e3, (r[1] - t1) / 1e3, short(r[4])))
