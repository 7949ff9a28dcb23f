#!/usr/bin/env python3
"""GPU devtool (timing build): when do the blocks of k_front start?  Shows how many one-wave blocks are really resident."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_d(), stop=0.0)
s.iterate(8)
lib = C.CDLL(os.environ["TRAJADMM_LIB"]); lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
t = out[names.index("k_front")]; live = t[:, 0] != 0
st = (t[live, 0] - t[live, 0].min()) * 0.01; en = (t[live, 1] - t[live, 0].min()) * 0.01
print("blocks", live.sum(), "started within 2 us:", int((st < 2).sum()), "5 us:", int((st < 5).sum()), "20 us:", int((st < 20).sum()), "50us:", int((st < 50).sum()))
order = np.argsort(st)
print("start time percentiles:", np.percentile(st, [10, 25, 50, 75, 90, 99]).round(1))
print("first block index starting after 20 us:", int(np.flatnonzero(live)[st > 20].min()) if (st > 20).any() else None)
