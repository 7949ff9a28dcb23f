#!/usr/bin/env python3
"""GPU devtool: distance of the HIP path from the unmodified reference on the single-UAV envelope fixtures
(tests/golden/envelope_{scn_a,scn_a_seed7,hard_single}.npz)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("traj-opt-admm_amd")
from conftest import scene_by_name
sc = pkg.scenes
for name, scene in (("scn_a", sc.scn_a()), ("scn_a_seed7", sc.scn_a(n_points=20000, seed=7)), ("hard_single", scene_by_name(sc, "hard_single"))):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"envelope_{name}.npz"))
    s = pkg.Solver(scene)
    gn, it, conv = s.iterate(300)
    a = s.get_state()["spline"]
    rel = lambda x, y: np.max(np.abs(x - y)) / np.max(np.abs(y))
    print(name, "iters", it, int(g["iters"]), "HIP vs reference", rel(a, g["final_spline"]), "reference 1-ulp envelope", rel(g["final_spline_pert"], g["final_spline"]))
    s.close()
