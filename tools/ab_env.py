"""A/B of environment switches on the two-stream C3 step (the child process of tools/sensitivity_probe.py, unmodified step):
python tools/ab_env.py "" "FDGS_BLEND_BWD_SPLIT=2" "FDGS_BLEND_BWD_SPLIT=4 FDGS_BLEND_FWD_SPLIT=2" ...   (each configuration twice, interleaved)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sensitivity_probe import CHILD  # noqa: E402


def run(envs):
    env = dict(os.environ)
    for kv in envs.split():
        k, v = kv.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, "-c", CHILD, "none"], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    return (float(line[0].split()[3]), float(line[0].split()[6])) if line else (None, out.stderr[-300:])


if __name__ == "__main__":
    cfgs = sys.argv[1:] or [""]
    res = {c: [] for c in cfgs}
    for rep in range(2):
        for c in cfgs:
            res[c].append(run(c))
    for c in cfgs:
        print("%-60s median ms/step %s" % (c or "(default)", res[c]))
