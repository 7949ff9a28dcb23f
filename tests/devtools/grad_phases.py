#!/usr/bin/env python3
"""Development tool (GPU only, `make -C traj-opt-admm_amd/csrc timing`): k_grad's phase stamps (thread 0 of every block) as a table -- mean and max time after the
block's own start at which each phase ends, and which blocks end last -- for one iteration of the timed window or the steady phase.
  python tests/devtools/grad_phases.py [iterations ...]        slots: 0 entry, 1 staged, 2 plane terms, 3 hand-over, 4 consensus / LLT check, 5 store phase, 6 end"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
np.set_printoptions(linewidth=220, suppress=True)
for n_it in [int(a) for a in sys.argv[1:]] or [9, 12, 25]:
    s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
    s.iterate_async(n_it); s.sync()
    out = np.zeros((len(names), 65536, 8), dtype=np.int64)
    lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
    t = out[names.index("k_grad")][:320].astype(float)
    t0 = t[:, 0].min()
    rel = (t - t[:, :1]) * 0.01
    rel[t == 0] = np.nan
    print(f"== after {n_it} iterations: block starts {((t[:,0]-t0)*0.01).min():.1f} .. {((t[:,0]-t0)*0.01).max():.1f}; kernel end {((t[:,6]-t0)*0.01).max():.1f}")
    print("   slot mean (us after the block's start):", np.round(np.nanmean(rel[:, 1:7], axis=0), 2))
    print("   slot max                               :", np.round(np.nanmax(rel[:, 1:7], axis=0), 2))
    last = np.argsort(-(t[:, 6]))[:5]
    for b in last: print(f"   block {b:3d} (robot {b // 5}, piece {b % 5}) slots 1..6:", np.round(rel[b, 1:7], 2))
    s.close()
