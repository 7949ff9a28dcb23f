#!/usr/bin/env python3
"""Development tool (GPU only): long runs of the round-4 launch shapes against the plain ones.  For every scene: N iterations with the defaults
(helper blocks in k_linesearch, k_grad launch order, one-launch coupled search) and N with all of them off (TJ_LS_HELP=1 TJ_GRAD_BALANCE=0
TJ_LSC_WIDE=0), in child processes; the final states must agree bit for bit and no error bit may be set.   python tests/devtools/soak_round4.py [N]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import importlib, sys, hashlib, numpy as np
sys.path.insert(0, sys.argv[3])
pkg = importlib.import_module("traj-opt-admm_amd")
sc = pkg.scenes
name, n = sys.argv[1], int(sys.argv[2])
scene = {"C": sc.scn_c, "B": sc.scn_b, "A": sc.scn_a, "E": sc.scn_e, "hard": sc.hard, "tiny": sc.tiny, "Cc": lambda: dict(sc.scn_c(), mode=2), "Bc": lambda: dict(sc.scn_b(), mode=2),
         "stack030": lambda: sc.crossing(8, 4000, seed=3, dz=0.30) if hasattr(sc, "crossing") else sc.tiny()}[name]()
s = pkg.Solver(scene, stop=0.0)
done = 0
while done < n:
    s.iterate(min(50, n - done)); done += min(50, n - done)
st = s.get_state()
print(hashlib.sha1(b"".join(np.ascontiguousarray(st[k]).tobytes() for k in sorted(st))).hexdigest(), s.stats()["error_bits"], s.stats()["energy_evals"])
'''
n = sys.argv[1] if len(sys.argv) > 1 else "300"
bad = 0
for name in ("tiny", "hard", "A", "B", "C", "Bc", "Cc", "E"):
    outs = []
    # default | everything off | helpers that start 10 us late (TJ_LS_HELP_LATE) | helpers that never post (TJ_LS_HELP_MUTE): round 5 added the two hooks
    for extra in ({}, dict(TJ_LS_HELP="1", TJ_GRAD_BALANCE="0", TJ_LSC_WIDE="0"), dict(TJ_LS_HELP_LATE="10"), dict(TJ_LS_HELP_MUTE="1")):
        env = dict(os.environ)
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code, name, n, ROOT], env=env, capture_output=True, text=True)
        outs.append(r.stdout.strip() or ("ERR " + r.stderr[-300:]))
    ok = all(o == outs[0] for o in outs) and outs[0].split()[1:2] == ["0"]
    bad += 0 if ok else 1
    print(f"{name:6s} {n} iterations: {'same bits, no error bit' if ok else 'MISMATCH'}   default: {outs[0]}   plain: {outs[1]}   late: {outs[2]}   mute: {outs[3]}", flush=True)
sys.exit(1 if bad else 0)
