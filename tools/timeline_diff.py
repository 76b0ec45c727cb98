#!/usr/bin/env python3
"""Developer tool: two timelines of tools/timeline.sh side by side, kernel by kernel (per queue, in launch order): duration and idle gap of
each launch under build A and build B, and the totals per queue. usage: timeline_diff.py A.txt B.txt"""
import sys, collections
def load(p):
    q = collections.OrderedDict()
    for line in open(p).read().splitlines()[1:]:
        f = line.split()
        q.setdefault(f[4], []).append((f[6], float(f[2]), float(f[3]), float(f[0])))
    return q, open(p).readline().strip()
a, ha = load(sys.argv[1]); b, hb = load(sys.argv[2])
print("A:", ha); print("B:", hb)
qa, qb = sorted(a, key=lambda k: -len(a[k])), sorted(b, key=lambda k: -len(b[k]))
for ka, kb in zip(qa, qb):
    print("queue %s / %s: %d / %d launches" % (ka, kb, len(a[ka]), len(b[kb])))
    da = db = ga = gb = 0.0
    for i, (x, y) in enumerate(zip(a[ka], b[kb])):
        flag = " <<" if abs(y[1] - x[1]) > 4 or abs(y[2] - x[2]) > 4 else ""
        print("  %-18s start %8.1f %8.1f | dur %7.1f %7.1f (%+6.1f) | gap %6.1f %6.1f (%+5.1f)%s" % (x[0] if x[0] == y[0] else x[0] + "/" + y[0], x[3], y[3], x[1], y[1], y[1] - x[1], x[2], y[2], y[2] - x[2], flag))
        da += x[1]; db += y[1]; ga += x[2]; gb += y[2]
    print("  totals: busy %.1f -> %.1f us, gaps %.1f -> %.1f us" % (da, db, ga, gb))
