#!/usr/bin/env python3
"""Development tool (timing build, GPU only): first and last phase stamp of every kernel of ONE iteration on the common wall
clock -- what lies between the last stamp of a kernel and the first stamp of the next is dispatch + ramp + the unstamped
prologue (kernel arguments, the `done` test)."""
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
rows = []
for k, n in enumerate(names):
    t = out[k]
    if n in ("k_sep_self_compact", "k_begin", "k_sep_self_solve", "k_obs_query") or not (t[:, 0] != 0).any():
        continue
    live = t[t[:, 0] != 0]
    vals = live[:, :7][live[:, :7] > 0]
    rows.append((live[:, 0].min(), vals.max(), n, len(live)))
rows.sort()
t0 = rows[0][0]
prev_end = None
for a, b, n, cnt in rows:
    gap = "" if prev_end is None else f"   gap since previous kernel's last stamp {0.01 * (a - prev_end):5.2f} us"
    print(f"{n:18s} blocks {cnt:5d}  first stamp {0.01 * (a - t0):7.2f}  last stamp {0.01 * (b - t0):7.2f}{gap}")
    prev_end = b
