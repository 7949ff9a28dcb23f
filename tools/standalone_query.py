import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ["TRAJADMM_LIB"] = "/root/repo/traj-opt-admm_amd/libtrajadmm_timing.so"
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(10)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
nk = lib.tj_kernel_count()
out = np.zeros((nk, 4096, 8), dtype=np.int64)
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)   # clear
s.run_stage("begin"); s.run_stage("planes_obs")
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
names = [lib.tj_kernel_name(i) for i in range(nk)]
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(nk)]
k = names.index("k_obs_query")
t = out[k]; live = t[:, 0] != 0; t = t[live]
d = np.diff(t[:, :3], axis=1) * 0.01
print("standalone k_obs_query:", live.sum(), "blocks; hull+kdop mean %.2f max %.2f; bvh mean %.2f max %.2f; span %.1f us" % (d[:,0].mean(), d[:,0].max(), d[:,1].mean(), d[:,1].max(), (t[:, :3].max()-t[:,0].min())*0.01))
