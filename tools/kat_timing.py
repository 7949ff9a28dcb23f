"""micro-timing of the device primitives through the KAT hooks (one wave of 64 cases per launch)"""
import sys, os, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("traj-opt-admm_amd")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "gjk_kat.npz"))
p = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "prims_kat.npz"))
s = pkg.Solver(pkg.scenes.tiny(1), stop=0.0, kat=True)
def t(f, reps=200):
    f(); t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
for shape in ("6v1", "6v6", "12v1", "12v12"):
    a, b = g[f"gjk_{shape}_a"][:64], g[f"gjk_{shape}_b"][:64]
    a1, b1 = a[:1].repeat(64, 0), b[:1].repeat(64, 0)
    print(shape, "64 different cases: %.1f us/launch (incl ~copies)" % t(lambda: s.kat_gjk(a, b)), " 64x same case: %.1f" % t(lambda: s.kat_gjk(a1, b1)))
ok = p["plane_self"][:, 0] == 1
P, Q = p["P"][ok][:64], p["Q"][ok][:64]
print("plane_pair (gjk+newton) 64 cases: %.1f" % t(lambda: s.kat_planes(1, P, Q, 0.3)), " same case x64: %.1f" % t(lambda: s.kat_planes(1, P[:1].repeat(64, 0), Q[:1].repeat(64, 0), 0.3)))
print("plane_obs 64 cases: %.1f" % t(lambda: s.kat_planes(0, p["P"][:64], p["q"][:64], 0.2)))
print("kdop hull/hull 64 cases: %.1f" % t(lambda: s.kat_planes(3, P, Q, 0.3)))
print("empty-ish (kdop point): %.1f" % t(lambda: s.kat_planes(2, p["P"][:64], p["q"][:64], 0.2)))
