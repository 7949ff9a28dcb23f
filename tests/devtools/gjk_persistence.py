#!/usr/bin/env python3
"""Development tool (light timing build, GPU only): how stable is the GJK iteration count of a robot pair from one ADMM
iteration to the next?  (The head start of kernels_pairs.h bets on the slow pairs of iteration i being those of i + 1.)"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing_light.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
prev = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    s.iterate(1)
    lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
    flat = out[names.index("k_sep_self_rows")].reshape(-1)
    n = int(min(flat[0], 4000))
    cur = {int(flat[8 + 2 * i]): int(flat[9 + 2 * i]) for i in range(n)}
    slow = {k: g for k, g in cur.items() if g >= 8}
    was = [prev.get(k, 0) for k in slow]   # 0: below 4 in the previous iteration (not recorded)
    print(f"iter {it:2d}: queries >= 4 its: {n:4d}; >= 8: {len(slow):3d}; of those, previous count: <4: {sum(1 for w in was if w == 0):3d}  4-6: {sum(1 for w in was if 4 <= w <= 6):3d}  >=7: {sum(1 for w in was if w >= 7):3d}"
          f"   | the 6 longest now (its, previous): {sorted(((g, prev.get(k, 0)) for k, g in slow.items()), reverse=True)[:6]}")
    prev = cur
