#!/usr/bin/env python3
"""Development tool (timing build, GPU only): producers / consumers of the pair solve of large fleets (k_mid, SCN-D)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
scene = pkg.scenes.scn_d()
s = pkg.Solver(scene, stop=0.0)
s.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
km = out[names.index("k_mid")]; ps = out[names.index("k_sep_self_solve")]
live = km[:, 0] != 0
t0 = km[live, 0].min()
n_sl = scene["U"] * scene["P"]
print("k_mid blocks: slack end max %.1f, all end max %.1f us" % ((km[:n_sl, 1].max() - t0) * 0.01, (km[live, 1].max() - t0) * 0.01))
pair = ps[n_sl:n_sl + 1024]
prod = pair[:, 3] > t0          # producers leave slot 3
p = pair[prod]
print("producers %d: start %.1f..%.1f  gjk done mean %.1f max %.1f  finish mean %.1f max %.1f us" % (prod.sum(), (p[:, 0].min() - t0) * .01, (p[:, 0].max() - t0) * .01,
      (p[:, 1].mean() - t0) * .01, (p[:, 1].max() - t0) * .01, (p[:, 3].mean() - t0) * .01, (p[:, 3].max() - t0) * .01))
cons = (~prod) & (pair[:, 0] > t0)
c = pair[cons]
got = c[:, 1] > c[:, 0]
print("consumers %d (with work %d): start %.1f..%.1f  first entry at mean %.1f max %.1f  exit mean %.1f max %.1f us" % (cons.sum(), got.sum(), (c[:, 0].min() - t0) * .01, (c[:, 0].max() - t0) * .01,
      (c[got, 1].mean() - t0) * .01 if got.any() else 0, (c[got, 1].max() - t0) * .01 if got.any() else 0, (c[:, 2].mean() - t0) * .01, (c[:, 2].max() - t0) * .01))
obs = km[n_sl + 1024:]; lo = obs[:, 0] != 0
print("obstacle solve blocks %d: start %.1f..%.1f end max %.1f us" % (lo.sum(), (obs[lo, 0].min() - t0) * .01, (obs[lo, 0].max() - t0) * .01, (obs[lo, 1].max() - t0) * .01))
