#!/usr/bin/env python3
"""Development tool (GPU only, `make -C traj-opt-admm_amd/csrc timing`): k_grad block by block on one clock -- thread 0's phase stamps, group B's (thread 192),
the plane batch's sub-steps, and the arrival of every WAVE at the hand-over barrier (timing builds store it under the unused k_ccd_prep id): the instrument
that found the mixed-role wave of round 4 (one wave 4 us behind the other five).   python tests/devtools/grad_waves.py [iteration]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
os.environ["TJ_GRAD_BALANCE"] = "0"
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 25)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
t = out[names.index("k_grad")][:320]; x = out[names.index("k_sep_self_compact")][:320]; w = out[names.index("k_ccd_prep")][:320]
np.set_printoptions(linewidth=220)
for b in (70, 71, 72, 73, 74, 100, 101, 102):
    t0 = t[b, 0]
    print("block", b, "piece", b % 5, "A (thread 0) slots 0..7:", np.round((t[b, :8] - t0) * 0.01, 2), " B (thread 192) records start/end, barrier, accumulate end:", np.round((x[b, :4] - t0) * 0.01, 2), " A batch stamps (planes staged, derivatives, M sums):", np.round((x[b, 4:7] - t0) * 0.01, 2), " waves 0..6 at the hand-over:", np.round((w[b, :7] - t0) * 0.01, 2))
