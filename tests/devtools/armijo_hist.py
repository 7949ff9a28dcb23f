import importlib, sys, ctypes as C
import numpy as np
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
prev = 0
for it in range(40):
    s.iterate(1)
    st = s.stats()
    ev = st["energy_evals"] - prev; prev = st["energy_evals"]
    a = np.zeros(64); b = np.zeros(64); c = np.zeros(64)
    s._check(s.lib.tj_get_steps(s._ctx, a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), c.ctypes.data_as(C.POINTER(C.c_double))))
    k = np.round(np.log(np.maximum(c, 1e-300) / np.minimum(a, b)) / np.log(0.8)).astype(int)
    print(it, "evals/robot %.1f" % (ev / 64), "k_acc max", k.max(), "hist", np.bincount(np.clip(k, 0, 24))[:24].tolist())
