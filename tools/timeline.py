"""Prints a slice of a rocprofv3 --kernel-trace CSV as a per-queue timeline (dev tool): python tools/timeline.py <csv> [first fraction] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("fdgs::", ""), r["Queue_Id"]) for r in rows)
s = int(len(ev) * frac)
t0 = ev[s][0]
for a, b, k, q in ev[s:s + n]:
    print("%9.1f %9.1f  q%-3s %7.1f us  %s" % ((a - t0) / 1e3, (b - t0) / 1e3, q, (b - a) / 1e3, k[:44]))
