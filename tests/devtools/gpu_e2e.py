"""Free-running end-to-end comparison HIP path vs CPU oracle (diagnostic). Usage: gpu_e2e.py <scene> <max_iters> [port|ref]"""
import sys, os, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
pkg = importlib.import_module("traj-opt-admm_amd")
sc = pkg.scenes
from oracle.pyoracle import Engine

def rel(a, b): return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))
which = sys.argv[1]; n = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "port"
scene = {"H": sc.hard, "H8": lambda: sc.hard(8, 20000), "A": sc.scn_a, "B": sc.scn_b, "C": sc.scn_c}[which]()
O = Engine(kind, scene); G = pkg.Solver(scene)
for it in range(n):
    go = O.iterate(); gg, itg, conv = G.iterate(1)
    a, b = G.get_state(), O.get_state()
    if it % 4 == 0 or go < 1e-2:
        print(it, f"g_cpu={go:.9g} g_gpu={gg:.9g} rel_spline={rel(a['spline'], b['spline']):.2e} rel_pt={rel(a['piece_time'], b['piece_time']):.2e} rel_all={max(rel(a[k], b[k]) for k in a):.2e}", flush=True)
    if it > 1 and go < 1e-2: break
gg, itg, conv = G.iterate(1)
print("gpu converged flag after one more call:", conv, "iter", itg, "stats", G.stats())
