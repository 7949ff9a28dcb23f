#!/usr/bin/env python3
"""Per-kernel hipEvent times of a bench run, side by side for several builds / env settings:
   tools/kernel_ms.py "<env assignments>" ["<env assignments>" ...] [-- bench args]"""
import json, os, subprocess, sys
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); extra = args[i + 1:]; args = args[:i]
rows = []
for v in args:
    env = dict(os.environ)
    for kv in v.split():
        k, _, val = kv.partition("="); env[k] = val
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--no-cpu"] + extra,
                         capture_output=True, text=True, env=env).stdout.strip().splitlines()[-1]
    j = json.loads(out)
    rows.append((v, j["ms_per_step"], j["roofline"]["kernel_ms_per_launch"]))
names = [k for k, t in rows[0][2].items() if any(r[2].get(k, 0) > 0 for r in rows)]
print("%-22s" % "kernel" + "".join("%12s" % ("v%d" % i) for i in range(len(rows))))
for k in names:
    print("%-22s" % k + "".join("%12.2f" % (1e3 * r[2].get(k, 0)) for r in rows))
print("%-22s" % "ms_per_step" + "".join("%12.4f" % r[1] for r in rows))
for i, r in enumerate(rows):
    print("v%d = %s" % (i, r[0] or "(default)"))
