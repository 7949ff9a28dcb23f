#!/usr/bin/env python3
"""State hashes of the standard scenes after a fixed number of iterations (GPU only): run it with two builds of the library (TRAJADMM_LIB=<other .so>) to see whether a
change moved a bit anywhere.   python tools/state_hashes.py [iterations]      prints: scene, sha256 of the state, error bits, pair planes, energy evaluations"""
import sys, os, importlib, hashlib
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("traj-opt-admm_amd"); sc = pkg.scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for name, mk, kw in (("A", sc.scn_a, {}), ("B", sc.scn_b, {}), ("C", sc.scn_c, {}), ("H8", lambda: sc.hard(8, 8000), {}), ("Bc", lambda: dict(sc.scn_b(), mode=2), {}), ("Cc", lambda: dict(sc.scn_c(), mode=2), {}),
                     ("B-optplane", sc.scn_b, {"optimal_plane": 1}), ("fleet100", lambda: sc.crossing(100, 20000, seed=121), {}), ("P7", lambda: sc.crossing(8, 8000, seed=5, piece_num=7) if "piece_num" in sc.crossing.__code__.co_varnames else sc.scn_b(), {})):
    s = pkg.Solver(mk(), stop=0.0, **kw)
    for b in (1, n // 2, n - 1 - n // 2):
        s.iterate_async(b); s.sync()
    st, t = s.get_state(), s.stats()
    h = hashlib.sha256()
    for k in sorted(st): h.update(np.ascontiguousarray(st[k]).tobytes())
    print(f"{name:11s} {h.hexdigest()[:20]} err {t['error_bits']} planes_self {t['planes_self']} evals {t['energy_evals']}")
    s.close()
