import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("traj-opt-admm_amd"); sc = pkg.scenes
for name, mk in [("hard7", lambda: sc.hard(U=7, n_points=500, seed=0, dz=0.13)), ("hard8", lambda: sc.hard(8, 20000)), ("hard12", lambda: sc.hard(U=12, n_points=2000, seed=5, dz=0.10)),
                 ("hard16", lambda: sc.hard(U=16, n_points=2000, seed=7, dz=0.08, radius=3.0)), ("cross24", lambda: sc.crossing(24, 6000, seed=17)), ("cross16dz", lambda: sc.crossing(16, 4000, seed=3, dz=0.12))]:
    res = []
    for fold in ("1", "0"):
        os.environ["TJ_SEQ_FOLD"] = fold
        try:
            s = pkg.Solver(mk(), stop=0.0); s.iterate(30); st = s.stats(); res.append((s.get_state(), st)); s.close()
        except Exception as e:
            res.append((None, str(e)[:100]))
    if res[0][0] is None or res[1][0] is None:
        print(name, "error", res[0][1], res[1][1]); continue
    same = all(np.array_equal(res[0][0][k], res[1][0][k]) for k in res[0][0])
    print(name, "bitwise equal:", same, "ambiguous", res[0][1]["order_ambiguous"], res[1][1]["order_ambiguous"], "errors", res[0][1]["error_bits"], res[1][1]["error_bits"])
