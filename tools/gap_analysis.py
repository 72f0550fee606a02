"""GPU idle-gap analysis of a rocprofv3 --kernel-trace CSV (dev tool): where does the device wait for the host?

usage: python tools/gap_analysis.py <..._kernel_trace.csv> [skip_fraction]
Prints busy / idle time over the steady-state part of the trace and the kernels that most often precede an idle gap.
"""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows))
ev = ev[int(len(ev) * skip):]
busy = sum(e - s for s, e, _ in ev)
wall = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
for (s0, e0, k0), (s1, e1, k1) in zip(ev, ev[1:]):
    g = s1 - e0
    if g > 0:
        key = "%s -> %s" % (k0[-40:], k1[-40:])
        gaps[key][0] += g
        gaps[key][1] += 1
print("kernels %d  wall %.3f ms  busy %.3f ms (%.1f%%)  idle %.3f ms" % (len(ev), wall / 1e6, busy / 1e6, 100.0 * busy / wall, (wall - busy) / 1e6))
for k, (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print("%8.1f us total  %5d x  avg %6.1f us   %s" % (t / 1e3, n, t / n / 1e3, k))
