#!/usr/bin/env python3
"""GPU devtool: distribution of |refined plane - reference| after ONE teacher-forced plane stage with "optimal_plane":1
(pre-stage tables from the unmodified reference, tests/golden/optplane_stages_*.npz)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("traj-opt-admm_amd")
import test_gpu_optplane as T
STATE = ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")
for name in ("tiny_multi", "tiny_multi_coupled"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"optplane_stages_{name}.npz"))
    scene = T._scene(pkg.scenes, name)
    s = pkg.Solver(scene, stop=0.0, optimal_plane=1)
    for it in g["kept"]:
        k = f"it{it}_"
        s.set_state({n: g[k + "pre_" + n] for n in STATE})
        s.set_pair_cache(g[k + "pre_cache_on"], g[k + "pre_cache_cd"])
        s.stage_planes()
        on, cd = s.get_pair_cache()
        want = g[k + "post_cache_cd"]
        sel = on.astype(bool)
        err = np.max(np.abs(cd[sel] - want[sel]), axis=1)
        srt = np.sort(err)
        print(name, "it", int(it), "stored planes", int(sel.sum()), "err median %.1e p90 %.1e max %.1e" % (np.median(err), srt[int(0.9 * len(srt))], err.max()), "count > 1e-12:", int((err > 1e-12).sum()), " > 1e-9:", int((err > 1e-9).sum()))
    s.close()
