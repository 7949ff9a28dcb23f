#!/usr/bin/env python3
"""Soak of the cross-queue protocols as of round 6 (development tool, GPU only; not part of the pytest suite): long runs of the asynchronous front (k_front next to k_linesearch, k_mid waiting in its solve waves), the asynchronous Newton solve and the asynchronous
plane refinement against the one-queue chain -- states bitwise equal, no error bit, no helper time-out -- on scenes that exercise the steady state (thousands of iterations
at the fixed point with the stop test off), the back-off regime, CCD contacts and acting pairs.   python tests/devtools/soak_round6.py [iterations]"""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys, os, importlib, hashlib, numpy as np
sys.path.insert(0, sys.argv[4])
pkg = importlib.import_module("traj-opt-admm_amd"); sc = pkg.scenes
scene = {"A": sc.scn_a, "B": sc.scn_b, "C": sc.scn_c, "E": sc.scn_e, "D": sc.scn_d, "hard8": lambda: sc.hard(8, 8000), "hard64": lambda: sc.hard(64, 20000), "Cc": lambda: dict(sc.scn_c(), mode=2), "Bc": lambda: dict(sc.scn_b(), mode=2), "Dtri": sc.scn_d_tri}[sys.argv[1]]()
s = pkg.Solver(scene, stop=0.0, optimal_plane=int(sys.argv[3]))
n = int(sys.argv[2]); h = hashlib.sha256()
for chunk in range(4):     # four batches: the gates / tickets restart cleanly between batches, a state read in between changes nothing
    s.iterate_async(n // 4); s.sync()
    st = s.get_state()
    for k in sorted(st): h.update(np.ascontiguousarray(st[k]).tobytes())
t = s.stats()
print(h.hexdigest()[:20], t["error_bits"], t["ls_giveups"], t["ls_helper_timeouts"], t["energy_evals"], t["async_fallbacks"])
'''
def run(scene, n, optplane, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CODE, scene, str(n), str(optplane), ROOT], env=e, capture_output=True, text=True, timeout=900)
    return r.stdout.strip() or ("FAILED " + r.stderr[-300:])
def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    bad = 0
    for scene, scale, op in (("C", 1.0, 0), ("A", 1.0, 0), ("B", 1.0, 0), ("hard8", 0.5, 0), ("hard64", 0.25, 0), ("E", 0.25, 0), ("D", 0.1, 0), ("Dtri", 0.05, 0), ("Cc", 0.5, 0), ("Bc", 1.0, 0), ("C", 0.25, 1), ("B", 0.5, 1)):
        m = max(8, int(n * scale) // 4 * 4)
        a = run(scene, m, op, {}); b = run(scene, m, op, {"TJ_XS_ASYNC": "0", "TJ_KEEP_ASYNC": "0", "TJ_FRONT_ASYNC": "0"})
        ok = a.split()[0] == b.split()[0] and not a.startswith("FAILED") and a.split()[1:4] == ["0", "0", "0"] and a.split()[5] == "0"   # same state hashes; no error bit, give-up, helper time-out or fallback on a GPU of the solver's own
        bad += 0 if ok else 1
        print(f"{scene:7s} optimal_plane={op} {m:5d} iterations: {'EQUAL ' if ok else 'DIFFER'} | two queues: {a} | one queue: {b}", flush=True)
    print("soak:", "ok" if bad == 0 else f"{bad} scene(s) differ")
    sys.exit(1 if bad else 0)
if __name__ == "__main__":
    main()
