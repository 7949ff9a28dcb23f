import importlib, sys
sys.path.insert(0, "/root/repo")
import numpy as np
pkg = importlib.import_module("traj-opt-admm_amd")
from oracle.pyoracle import Engine
scene = pkg.scenes.tiny(mode=1, U=3, n_points=600)
g = pkg.Solver(scene, stop=0.0); o = Engine("port", scene)
g.iterate(2); 
for _ in range(2): o.iterate()
st = o.get_state()
worst = 0
for trial, tsc in enumerate((1.0, 0.05, 0.01, 5.0)):
    s2 = {k: v.copy() for k, v in st.items()}
    s2["t_slack"] = s2["t_slack"] * tsc
    rng = np.random.default_rng(trial)
    s2["p_slack"] = s2["p_slack"] + 0.3 * rng.standard_normal(s2["p_slack"].shape)
    g.set_state(s2); o.set_state(s2)
    g.stage_slack(); o.stage_slack()
    a, b = g.get_state(), o.get_state()
    for k in a:
        err = np.max(np.abs(a[k] - b[k])) / max(1.0, np.max(np.abs(b[k])))
        worst = max(worst, err)
        print(trial, tsc, k, "%.2e" % err)
print("worst", worst, "error bits", g.stats()["error_bits"])
