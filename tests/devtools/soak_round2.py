#!/usr/bin/env python3
"""Development tool (GPU only): long runs of the chains that contain in-kernel waits (k_grad's and k_xsolve's wave-group
barriers, the passed-on pair solves of large fleets), twice each -- error bits must stay 0 and the two runs must agree bitwise."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("traj-opt-admm_amd")
sc = pkg.scenes
for name, scene, iters in (("SCN-C", sc.scn_c(), 6000), ("crossing-U256", sc.crossing(256, 20000, seed=31), 1500), ("SCN-A", sc.scn_a(), 6000), ("SCN-B", sc.scn_b(), 6000)):
    states = []
    for rep in range(2):
        s = pkg.Solver(scene, stop=0.0)
        t0 = time.time()
        for _ in range(iters // 500):
            s.iterate(500)
        st = s.stats()
        states.append(s.get_state())
        print(f"{name} run {rep}: {iters} iterations in {time.time() - t0:.2f} s, error_bits {st['error_bits']}, pair_solves {st['pair_solves']}", flush=True)
        assert st["error_bits"] == 0
        s.close()
    for k in states[0]:
        assert np.array_equal(states[0][k], states[1][k]), (name, k)
print("soak ok")
