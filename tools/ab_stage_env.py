"""A/B of environment switches on bench.py's single-stream stage table and step rate (same box, interleaved runs):
python tools/ab_stage_env.py [--workload C5] "" "FDGS_PRE_STREAM=0" ...   -- prints each configuration's median stage times (us) and images/s."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
while args and args[0].startswith("--"):
    extra += args[:2]
    args = args[2:]
cfgs = args or [""]
rounds = int(os.environ.get("AB_ROUNDS", "2"))
res = {c: [] for c in cfgs}
for r in range(rounds):
    for c in cfgs:
        env = dict(os.environ)
        for kv in c.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "10", "--cpu-samples", "0", "--host-cost-steps", "0",
                              "--dropin-steps", "0", "--spatial-order-steps", "0", "--reflists-steps", "0", "--clustered-steps", "0", "--axis-steps", "0",
                              "--c5-steps", "0"] + extra, env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
        if not lines:
            print("run failed:", out.stderr[-500:])
            continue
        res[c].append(json.loads(lines[-1]))
med = lambda xs: sorted(xs)[len(xs) // 2] if xs else float("nan")  # noqa: E731
keys = list(res[cfgs[0]][0]["stages"].keys())
print("%-18s" % "stage [us]", *["%28s" % (c or "(default)")[:28] for c in cfgs])
for k in keys:
    print("%-18s" % k, *["%28.1f" % (1e3 * med([d["stages"].get(k, {"ms": 0.0})["ms"] for d in res[c]])) for c in cfgs])
for k in ("forward_ms", "ms_per_image", "value"):
    print("%-18s" % k, *["%28.4f" % med([d[k] for d in res[c]]) for c in cfgs])
