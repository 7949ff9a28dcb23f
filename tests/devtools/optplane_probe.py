"""GPU dev tool: `optimal_plane:1` on a scene, device vs oracle, teacher-forced per iteration; reports the first stage
that differs.  python tests/devtools/optplane_probe.py [hard|scn_b|tiny] [iters]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("traj-opt-admm_amd")
from conftest import canon  # noqa: E402
from oracle.pyoracle import Engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hard"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sc = {"hard": lambda: pkg.scenes.hard(4, 4000), "scn_b": pkg.scenes.scn_b, "tiny": lambda: pkg.scenes.tiny(1)}[name]()
o = Engine("port", sc); o.set_optimal_plane(True)
s = pkg.Solver(sc, stop=0.0, optimal_plane=1)
free = pkg.Solver(sc, stop=0.0, optimal_plane=1)
STATE = ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")
for it in range(iters):
    s.set_state(o.get_state()); s.set_pair_cache(*o.get_pair_cache())
    cd, pd = s.stage_planes()
    co, po = o.stage_planes()
    same_counts = np.array_equal(cd, co)
    diff = np.max(np.abs(canon(cd, pd) - canon(co, po))) if same_counts else None
    on_d, c_d = s.get_pair_cache(); on_o, c_o = o.get_pair_cache()
    print(f"it {it}: plane counts equal {same_counts} ({cd.sum()} vs {co.sum()}), max plane diff {diff}, cache flags equal {np.array_equal(on_d, on_o)}, "
          f"cache diff {np.nanmax(np.abs(c_d - c_o))}, device finite {np.isfinite(pd).all()}, stats err {s.stats()['error_bits']}")
    if not same_counts:
        bad = np.argwhere(cd != co)
        print("   differing (robot, segment):", bad[:8].tolist(), cd[cd != co][:8], co[cd != co][:8])
    o.stage_direction(); o.stage_steps(); o.stage_linesearch(); o.stage_slack()
    g, _, _ = free.iterate(1)
    print(f"      free-running device gnorm {g}  finite {np.isfinite(free.get_state()['spline']).all()}  err {free.stats()['error_bits']}")
