"""GPU dev tool: plan way points for the wall scene of tests/test_planner.py and watch the solver's gnorm from them."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("traj-opt-admm_amd")
from test_planner import _wall_scene  # noqa: E402

sc, starts, goals = _wall_scene(pkg.scenes)
U = len(starts)
s = pkg.Solver(dict(sc, U=U, waypoints=sc["waypoints"][:U]), stop=0.0)
wp = s.plan_init(starts, goals, min_waypoints=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
s.close()
print("way points per robot", wp.shape[1])
for u in range(U):
    print(u, np.round(wp[u], 3).tolist())
slv = pkg.Solver(dict(sc, U=U, P=wp.shape[1] - 1, waypoints=wp))
for k in range(40):
    g, it, conv = slv.iterate(50)
    print(it, g, conv, slv.get_state()["piece_time"])
    if conv:
        break
print(slv.stats())
