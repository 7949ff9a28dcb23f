import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "traj-opt-admm_amd", "libtrajadmm_timing_light.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(25)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
for n, a, b, c in (("k_grad", 7, 0, 6), ("k_xsolve", 7, 0, 6), ("k_linesearch", 7, 0, 5)):
    t = out[names.index(n)]; live = t[:, a] != 0
    d1 = (t[live, b] - t[live, a]) * 0.01; d2 = (t[live, c] - t[live, b]) * 0.01
    print(n, "entry -> first stamp: mean %.2f max %.2f | first -> last stamp: mean %.2f max %.2f" % (d1.mean(), d1.max(), d2.mean(), d2.max()))
