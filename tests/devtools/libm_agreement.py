"""How often do the device's log / sin / cos agree bit for bit with glibc's (what the reference calls)?  GPU box only.
python tests/devtools/libm_agreement.py  -> builds tools/micro/libm_probe.hip, runs it on 200 000 arguments of the kind
Optimal_plane feeds them (dist / margin in (0, 1), angles in [-pi/2, pi/2]) and prints the mismatch rates in ulps."""
import math, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
exe = os.path.join(out, "libm_probe")
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-o", exe, os.path.join(ROOT, "tools", "micro", "libm_probe.hip")], stderr=subprocess.DEVNULL)
rng = np.random.default_rng(3)
n = 200000
x = np.concatenate([rng.uniform(1e-6, 1.0, n // 2), rng.uniform(-math.pi / 2, math.pi / 2, n // 2)])
x[n // 2:][:1000] *= 1e-3
x.tofile(os.path.join(out, "libm_in.bin"))
subprocess.check_call([exe, os.path.join(out, "libm_in.bin"), os.path.join(out, "libm_out.bin")])
d = np.fromfile(os.path.join(out, "libm_out.bin")).reshape(-1, 3)
def ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))
lg = np.array([math.log(v) for v in x[:n // 2]]); sn = np.array([math.sin(v) for v in x[n // 2:]]); cs = np.array([math.cos(v) for v in x[n // 2:]])
for name, dev, ref in (("log", d[:n // 2, 0], lg), ("sin", d[n // 2:, 1], sn), ("cos", d[n // 2:, 2], cs)):
    u = ulps(np.ascontiguousarray(dev), ref)
    print(f"{name}: differs from glibc in {100.0 * (u > 0).mean():.3f} % of {len(u)} arguments, max {u.max()} ulp")
os.remove(exe)
