#!/usr/bin/env python3
"""Per-block phase stamps of the asynchronous Newton solve (development tool, GPU only): when k_grad's blocks end, when k_xsolve's blocks see their tickets, load,
assemble, factor, solve and raise their flags, and what k_ccd's units do after the flag (flag seen, record built, signalled, walk) -- all on one time axis that starts
with k_grad's first block.  Needs the timing build:  make -C traj-opt-admm_amd/csrc timing ;  python tools/xs_async_phases.py   (TJ_XS_ASYNC=0 python ...: the one-queue
chain for comparison).  Output of round 5: profiles/round5_xs_async_phase_stamps.txt."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate_async(25); s.sync()
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
np.set_printoptions(linewidth=220, suppress=True)
g = out[names.index("k_grad")][:320]; x = out[names.index("k_xsolve")][:64]; c = out[names.index("k_ccd")]
t0 = g[:, 0].min()
us = lambda a: (a - t0) * 0.01
print("k_grad: first stamp min/max", us(g[:, 0]).min(), us(g[:, 0]).max(), " store-phase start (5) mean/max", us(g[:, 5]).mean().round(1), us(g[:, 5]).max().round(1), " end (6) mean/max", us(g[:, 6]).mean().round(1), us(g[:, 6]).max().round(1))
ge = us(g[:, 6]).reshape(64, 5).max(axis=1) if False else None
print("k_xsolve stamps (us since k_grad's first block): slot 0 start, 7 after wait, 1 load end, 2 assemble end, 3 chol end, 4 backsolve end, 5 tail start, 6 end")
# a slot no block stamps in this schedule (7 on the one-queue chain: no wait; 5 with the asynchronous solve: no swept-hull tail in k_xsolve) holds zeros: it is left out of
# the table and of the differences instead of printing the clock's origin (round 5's file had "slot 5: min -98857243642.1")
stamped = [sl for sl in (0, 7, 1, 2, 3, 4, 5, 6) if (x[:, sl] != 0).all()]
for sl in (0, 7, 1, 2, 3, 4, 5, 6):
    if sl not in stamped:
        print(f"  slot {sl}: not stamped in this schedule"); continue
    v = us(x[:, sl]); print(f"  slot {sl}: min {v.min():7.1f} mean {v.mean():7.1f} max {v.max():7.1f}")
names_ph = {1: "load", 2: "assemble", 3: "chol+fwd", 4: "backsolve", 5: "finish", 6: "tail / end"}
seq = [sl for sl in (7, 1, 2, 3, 4, 5, 6) if sl in stamped]
if 7 not in stamped: seq = [0] + seq
d = np.diff(us(x[:, seq]), axis=1)
print("  phase durations (" + ", ".join(f"{names_ph[b]} [{a_}->{b}]" for a_, b in zip(seq[:-1], seq[1:])) + "): mean", d.mean(0).round(2), "max", d.max(0).round(2))
live = c[:, 0] != 0
print("k_ccd blocks:", live.sum(), " start min/max", us(c[live, 0]).min().round(1), us(c[live, 0]).max().round(1), " end(1) mean/max", us(c[live, 1]).mean().round(1), us(c[live, 1]).max().round(1), " finisher end(2)", us(c[0, 2]).round(1))

n_obs = 64 * 40
ob = c[1:1 + n_obs]; pt = c[1 + n_obs:]; pt = pt[pt[:, 0] != 0]
fl = us(x[:, 6])   # flag time per robot
u_of = np.arange(n_obs) // 40
print("obstacle units: flag seen - robot's flag:", (us(ob[:, 3]) - fl[u_of]).mean().round(2), (us(ob[:, 3]) - fl[u_of]).max().round(2), " record built (4) - flag seen:", (us(ob[:, 4]) - us(ob[:, 3])).mean().round(2), " signalled (5) - built:", (us(ob[:, 5]) - us(ob[:, 4])).mean().round(2), " walk end (1) - signalled:", (us(ob[:, 1]) - us(ob[:, 5])).mean().round(2), (us(ob[:, 1]) - us(ob[:, 5])).max().round(2))
print("pair tiles:", len(pt), " wait over (6): mean/max", us(pt[:, 6]).mean().round(1), us(pt[:, 6]).max().round(1), " end(1) - wait over: mean/max", (us(pt[:, 1]) - us(pt[:, 6])).mean().round(2), (us(pt[:, 1]) - us(pt[:, 6])).max().round(2), " end max", us(pt[:, 1]).max().round(1))
print("last flag", fl.max().round(1), " last obstacle unit end", us(ob[:, 1]).max().round(1))
