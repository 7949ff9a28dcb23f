#!/usr/bin/env python3
"""Development tool: absolute phase stamps of k_grad's two wave groups per block (timing build, GPU only)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
scene = pkg.scenes.scn_c()
s = pkg.Solver(scene, stop=0.0)
s.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 15)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
a = out[names.index("k_grad")][:320]; b = out[names.index("k_sep_self_compact")][:320]
t0 = a[:, 0].min()
print("blk | A: start stage planes join consensus lltend psd store | B: start records barrier accum | A, first batch: planes in LDS, derivatives, M sums   (us from the kernel's first stamp)")
order = np.argsort(-(a[:, 6] - a[:, 0]))
for i in list(order[:12]) + list(order[150:156]):
    A = (a[i, [0, 1, 2, 3, 4, 7, 5, 6]] - t0) * 0.01; B = (b[i, :7] - t0) * 0.01
    print(f"{i:3d} | " + " ".join(f"{x:6.2f}" for x in A) + " | " + " ".join(f"{x:6.2f}" for x in B))
