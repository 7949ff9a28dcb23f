#!/usr/bin/env python3
"""Development tool (timing build, GPU only): where the wave-cooperative GJK of the slow robot pairs of SCN-C spends its time."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 15)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
p = out[names.index("k_obs_solve")]; st = out[names.index("k_sep_self_solve")]
live = p[:, 4:7].sum(1) > 0
its = p[:, 4:7].sum(1)
order = np.argsort(-its)[:10]
print("slowest pairs: iterations (seg, tri, tet) | us: support, seg, tri, tet | us per step: support, seg, tri, tet")
for b in order:
    c = p[b, 4:7]; t = p[b, 0:4] * 0.01
    per = [t[0] / max(c.sum(), 1)] + [t[1 + i] / max(c[i], 1) for i in range(3)]
    print(f"  block {b}: {c.tolist()} | " + " ".join(f"{x:6.2f}" for x in t) + " | " + " ".join(f"{x:5.2f}" for x in per))
tot = p[live]
print("all pairs: steps seg/tri/tet", tot[:, 4:7].sum(0).tolist(), " us per step: support %.2f seg %.2f tri %.2f tet %.2f" % (
    tot[:, 0].sum() * 0.01 / max(tot[:, 4:7].sum(), 1), tot[:, 1].sum() * 0.01 / max(tot[:, 4].sum(), 1), tot[:, 2].sum() * 0.01 / max(tot[:, 5].sum(), 1), tot[:, 3].sum() * 0.01 / max(tot[:, 6].sum(), 1)))
