"""How much do kernels of different streams overlap?  (rocprofv3 --kernel-trace CSV; steady-state tail of the trace)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?")) for r in rows))
n0, n1 = int(len(ev) * skip), int(len(ev) * 0.8)
ev = ev[n0:n1]
wall = ev[-1][1] - ev[0][0]
tot = sum(e - s for s, e, _, _ in ev)
# union of busy intervals
cur_s, cur_e, union = ev[0][0], ev[0][1], 0
for s, e, _, _ in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("kernels %d  wall %.3f ms  sum of durations %.3f ms  busy (union) %.3f ms  idle %.3f ms  concurrency %.2f" % (
    len(ev), wall / 1e6, tot / 1e6, union / 1e6, (wall - union) / 1e6, tot / union))
by = collections.defaultdict(lambda: [0, 0])
for s, e, k, q in ev:
    by[(k[-36:], q)][0] += e - s; by[(k[-36:], q)][1] += 1
for (k, q), (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:16]:
    print("%9.1f us avg  %5d x   queue %s  %s" % (t / n / 1e3, n, q, k))
