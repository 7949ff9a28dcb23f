"""GPU dev tool: `optimal_plane:1` free-running to convergence on many scenes, device vs oracle: iteration counts, finiteness,
final control points.  (Converged results are only comparable at the mode's own sensitivity)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("traj-opt-admm_amd")
from oracle.pyoracle import Engine  # noqa: E402

sc = pkg.scenes
cases = [("tiny_single", sc.tiny(0, n_points=3000)), ("scn_a", sc.scn_a()), ("scn_a seed 7", sc.scn_a(seed=7)), ("tiny_multi", sc.tiny(1)), ("tiny coupled", dict(sc.tiny(1), mode=2)),
         ("scn_b", sc.scn_b()), ("scn_b coupled", dict(sc.scn_b(), mode=2)), ("hard", sc.hard(4, 4000)), ("hard coupled", dict(sc.hard(4, 4000), mode=2)),
         ("hard 8", sc.hard(8, 20000)), ("scn_c", sc.scn_c()), ("scn_e", sc.scn_e())]
only = sys.argv[1:] 
for name, scene in cases:
    if only and name not in only:
        continue
    o = Engine("port", scene); o.set_optimal_plane(True)
    gn = []
    for it in range(150 if scene["U"] < 100 else 40):
        gn.append(o.iterate())
        if it > 1 and gn[-1] < 1e-2:
            break
    so = o.get_state()
    s = pkg.Solver(scene, optimal_plane=1)
    g, itd, conv = s.iterate(len(gn) + 60)
    sd = s.get_state(); err = s.stats()["error_bits"]
    rel = np.max(np.abs(sd["spline"] - so["spline"])) / np.max(np.abs(so["spline"]))
    print(f"{name:14s} oracle iters {len(gn):3d} gnorm {gn[-1]:.3e} finite {np.isfinite(so['spline']).all()} | device iters {itd:3d} conv {conv} gnorm {g:.3e} finite {np.isfinite(sd['spline']).all()} err {err} | rel {rel:.2e}", flush=True)
    s.close()
