#!/usr/bin/env python3
"""Development tool (light timing build: `make -C traj-opt-admm_amd/csrc timing_light`, GPU only): one iteration of SCN-C on the
common 100 MHz wall clock -- per chain kernel the first block entry, the last block end, and what lies between kernels."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing_light.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 25)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
# (kernel, slot of the first stamp of a block, slots that may hold its last stamp)
spec = [("k_front", 0, [1]), ("k_mid", 0, [1]), ("k_grad", 7, [6]), ("k_xsolve", 7, [4, 6]), ("k_ccd", 0, [1, 2]), ("k_linesearch", 7, [5, 6])]
rows = []
for n, s0, ends in spec:
    t = out[names.index(n)]
    live = t[:, s0] != 0
    if not live.any():
        continue
    st = t[live, s0]; en = np.max(t[live][:, ends], axis=1)
    rows.append((n, st.min(), st.max(), np.median(en), en.max(), int(live.sum())))
t0 = rows[0][1]
prev = None
print("kernel         blocks | first entry  last entry | median end   last end | gap to the previous kernel's last end | span")
for n, a, b, c, d, cnt in rows:
    gap = "" if prev is None else "%6.2f" % ((a - prev) * 0.01)
    print(f"{n:14s} {cnt:6d} | {(a - t0) * 0.01:10.2f} {(b - t0) * 0.01:11.2f} | {(c - t0) * 0.01:10.2f} {(d - t0) * 0.01:10.2f} | {gap:>10s} | {(d - a) * 0.01:6.2f}")
    prev = d
