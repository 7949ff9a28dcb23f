"""optimal_plane:1 on SCN-C: where the slowest planes of one k_keep launch spend their time (needs `make -C traj-opt-admm_amd/csrc timing`: the
wave-form refinement opt_plane_pair_wave accumulates wall-clock time per phase -- barrier terms, their sums, Newton direction (LLT + Eigen's 3x3
eigenvalue iteration + solve), Armijo passes -- and the waves with >= 8 rounds leave the totals).  GPU only."""
import importlib, sys, os, numpy as np, ctypes as C
sys.path.insert(0, '.')
os.environ["TRAJADMM_LIB"] = os.path.join(os.getcwd(), "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0, optimal_plane=1)
s.iterate(15)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
t = out[names.index("k_keep")]
live = t[:, 0] > 0
print("waves with >= 8 rounds:", live.sum())
for row in sorted(t[live].tolist(), key=lambda r: -r[0])[:10]:
    rounds, npass = row[0], row[1]
    print("rounds", rounds, "passes", npass, "us: term %.1f sums %.1f direction %.1f armijo %.1f total %.1f | per round %.2f, per pass %.2f" % (row[2] * .01, row[3] * .01, row[4] * .01, row[5] * .01, sum(row[2:6]) * .01, sum(row[2:6]) * .01 / rounds, row[5] * .01 / max(npass, 1)))
