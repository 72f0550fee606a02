"""A/B of two builds of libfdgs.so on the stage table of bench.py (same box, alternating runs).
usage: python tools/ab_stages.py <libA.so> <libB.so> [rounds] -- prints the median stage times of each."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, FDGS_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "10", "--cpu-samples", "0",
                              "--host-cost-steps", "0", "--dropin-steps", "0", "--spatial-order-steps", "0", "--reflists-steps", "0", "--clustered-steps", "0",
                              "--axis-steps", "0", "--c5-steps", "0"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[l].append(json.loads(out))
med = lambda xs: sorted(xs)[len(xs) // 2]
keys = list(res[libs[0]][0]["stages"].keys())
print("%-16s" % "stage", *["%24s" % os.path.basename(l) for l in libs])
for k in keys:
    print("%-16s" % k, *["%24.4f" % med([d["stages"].get(k, {"ms": 0.0})["ms"] for d in res[l]]) for l in libs])
for k in ("forward_ms", "ms_per_image", "value"):
    print("%-16s" % k, *["%24.4f" % med([d[k] for d in res[l]]) for l in libs])
