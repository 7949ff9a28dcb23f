#!/usr/bin/env python3
"""Development tool (GPU only, `make -C traj-opt-admm_amd/csrc timing`): k_linesearch with helper blocks -- phase stamps of the primary blocks and of the helpers of
one launch (stage / planes / setup / evaluation / decision / publish), and the primary's decision stamps (own decided, posts polled, word stored, barrier, net formed,
published: slots of the unused k_ls_coupled id).   python tests/devtools/ls_super_rounds.py [iteration]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 25)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
t = out[names.index("k_linesearch")]
x = out[names.index("k_ls_coupled")]
live = t[:, 0] != 0
n = int(live.sum())
t0 = t[live, 0].min()
st = s.last_armijo_steps()
print("iter", sys.argv[1], "blocks with stamps:", n, " armijo exponents:", np.unique(np.round(np.log(st / st.max()) / np.log(0.8)).astype(int), return_counts=True), "max step", st.max())
np.set_printoptions(linewidth=200)
for lab, lo, hi in (("primary", 0, 64), ("helper1", 64, 128), ("helper3", 192, 256)):
    if hi > n: break
    tt = (t[lo:hi, :7] - t0) * 0.01
    tt[t[lo:hi, :7] == 0] = np.nan
    print(lab, "mean abs us of slots 0..6:", np.round(np.nanmean(tt, axis=0), 2), " max:", np.round(np.nanmax(tt, axis=0), 2))
xx = (x[0:64, :6] - t0) * 0.01
xx[x[0:64, :6] == 0] = np.nan
print("primary decision stamps (own decided, polled, word stored, barrier, net formed, published): mean", np.round(np.nanmean(xx, axis=0), 2), "max", np.round(np.nanmax(xx, axis=0), 2))
sl = np.argsort(-(t[:64, 5]))[:3]
for b in sl: print("slow primary", b, np.round((t[b, :7] - t0) * 0.01, 2), np.round((x[b, :6] - t0) * 0.01, 2))
