"""GPU dev tool: isolate (segment, pair) slots of the `hard` scene where the device's self_optimal_cd differs from the oracle."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("traj-opt-admm_amd")
from oracle.pyoracle import Engine, Prims  # noqa: E402

sc = pkg.scenes.hard(4, 4000)
o = Engine("port", sc); o.set_optimal_plane(True)
s = pkg.Solver(sc, stop=0.0, optimal_plane=1, kat=True)
st = o.get_state()
s.stage_planes(); o.stage_planes()
on_d, c_d = s.get_pair_cache(); on_o, c_o = o.get_pair_cache()
conv, M, basis, kd = pkg.host_tables(sc["P"], 8)
pr = Prims("port")
o2 = Engine("port", sc); o2.set_optimal_plane(True)   # Prims() re-initialised the oracle's globals: set the scene up again
P_, Q_, cin = [], [], []
bad = []
for tr, a, b in np.argwhere(on_o != 0):
    d = np.abs(c_d[tr, a, b] - c_o[tr, a, b]).max()
    if not (d < 1e-9):
        bad.append((tr, a, b, d))
print("slots on:", int(on_o.sum()), "bad:", len(bad), bad[:6])
res = 8
for tr, a, b, d in bad[:6]:
    piece = tr // res
    hull = lambda u: basis[tr] @ st["spline"][u][:, 3 * piece:3 * piece + 6].T
    P, Q = hull(a), hull(b)
    ok, cd0 = pr.plane_self(P, Q, 0.1 + 2 * 0.1, refine=False)
    ref_out = pr.self_optimal_cd(P, Q, cd0)
    fin6, out6 = s.kat_refine_planes(6, P[None], Q[None], cd0[None])
    fin7, out7 = s.kat_refine_planes(7, P[None], Q[None], cd0[None])
    print("slot", tr, a, b, "gjk ok", ok, "in", cd0, "\n   oracle", ref_out, "\n   dev6  ", out6[0], fin6, "\n   dev7  ", out7[0], fin7, "\n   cache dev", c_d[tr, a, b], "cache orc", c_o[tr, a, b])
    np.savez("/root/repo/gpurun_out/badpair_%d_%d_%d.npz" % (tr, a, b), P=P, Q=Q, cd0=cd0, ref_out=ref_out, out6=out6[0])
