#!/usr/bin/env python3
"""Development tool (timing build, GPU only): GJK iterations per robot pair in the one-pair-per-lane solve of large fleets."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_d(), stop=0.0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s.iterate(n)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
h = out[names.index("k_sep_self_rows")][:64, 0] / n
print("GJK iterations per pair, average per iteration over", n, "iterations (bin 63 = 63+):")
for i, v in enumerate(h):
    if v > 0: print(f"  {i:3d}: {v:9.1f}")
tot = h.sum(); it = (h * np.arange(64)).sum()
print(f"pairs {tot:.0f}, mean iterations {it / max(tot, 1):.2f}, share of pairs with >= 10: {h[10:].sum() / max(tot, 1):.3f}, >= 20: {h[20:].sum() / max(tot, 1):.4f}")
