"""Where does a two-stream step go?  From a rocprofv3 --kernel-trace CSV of bench.py: over the steady-state part of the trace, the
share of wall time during which (a) a VALU-bound kernel (blend_bwd / blend_fwd / ssim) is running, (b) only other kernels run,
(c) nothing runs; and per kernel its average duration in the trace.  python tools/step_timeline.py <kernel_trace.csv> [t0_frac] [t1_frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fdgs::", "")) for r in rows)
ev = [e for e in ev if not e[2].startswith(("naive_conv", "miopen", "Cijk", "ck::", "Im2d", "Col2Im", "_ZN2ck", "clock_sample"))]
# the window: whole optimizer steps from the longest run of evenly spaced fused SH updates (one per step) -- the timed region
marks = [s for s, e, k in ev if k.startswith("sh_flush_kernel")]
if len(marks) >= 8:
    d = sorted(b - a for a, b in zip(marks, marks[1:]))
    med = d[len(d) // 2]
    best, cur = (0, 0), 0
    for i, (a, b) in enumerate(zip(marks, marks[1:])):
        if abs((b - a) - med) <= 0.2 * med:
            cur += 1
            if cur > best[1] - best[0]:
                best = (i + 1 - cur, i + 1)
        else:
            cur = 0
    i0, i1 = best[0] + 2, min(best[1] - 1, best[0] + 14)
    t_lo, t_hi = marks[i0], marks[i1]
    nsteps = i1 - i0
    ev = [x for x in ev if x[0] >= t_lo and x[1] <= t_hi]
    print("window = %d optimizer steps of %.3f ms" % (nsteps, (t_hi - t_lo) / nsteps / 1e6))
else:
    ev = ev[int(len(ev) * f0):int(len(ev) * f1)]
    t_lo, t_hi = ev[0][0], ev[-1][1]
VALU = ("blend_bwd", "blend_fwd", "ssim_fwd", "ssim_bwd")
pts = []
for s, e, k in ev:
    heavy = any(k.startswith(v) for v in VALU)
    pts.append((s, 1, heavy)); pts.append((e, -1, heavy))
pts.sort()
nh = nl = 0
last = t_lo
acc = collections.Counter()
for t, d, heavy in pts:
    dt = t - last
    if dt > 0:
        acc["valu-bound kernel running" if nh > 0 else ("only latency/HBM-bound kernels" if nl > 0 else "idle")] += dt
        if nh > 0 and nl > 0:
            acc["  (of which: together with other kernels)"] += dt
        if nh > 1:
            acc["  (of which: two valu-bound kernels at once)"] += dt
    last = t
    if heavy: nh += d
    else: nl += d
wall = t_hi - t_lo
print("window %.2f ms, %d kernels" % (wall / 1e6, len(ev)))
for k, v in acc.items():
    print("%-48s %6.1f %%" % (k, 100.0 * v / wall))
by = collections.defaultdict(lambda: [0, 0])
for s, e, k in ev:
    by[k][0] += e - s; by[k][1] += 1
print("sum of kernel durations / wall = %.2f" % (sum(v[0] for v in by.values()) / wall))
for k, (t, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:18]:
    print("%9.1f us avg %6d x  %5.1f %% of wall  %s" % (t / n / 1e3, n, 100.0 * t / wall, k[:50]))
