#!/usr/bin/env python3
"""Development tool (light timing build: `make -C traj-opt-admm_amd/csrc timing_light`, GPU only): per iteration of SCN-C, when the
last block of each class of the union kernels k_front / k_mid ends (us from the kernel's first block start)."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing_light.so")
pkg = importlib.import_module("traj-opt-admm_amd")
scene = pkg.scenes.scn_c()
s = pkg.Solver(scene, stop=0.0)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sf = False; hs = os.environ.get("TJ_PAIR_HEAD_START", "1") != "0"
U, P, S = scene["U"], scene["P"], scene["P"] * 8
n_rows = S * ((U + 7) // 8)
front = ([("head", 128)] if hs else []) + [("query", U * S), ("rows", 65536)]
mid = ([] if sf else [("slack", U * P)]) + [("pair", 1024), ("obs", 1024)]
print("iter | k_front: " + " ".join(f"{n:>6s}" for n, _ in front) + " | k_mid: " + " ".join(f"{n:>6s}" for n, _ in mid) + " | gjk max | head starts")
prev_hs = 0
for it in range(n_it):
    s.iterate(1)
    lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
    row = []
    for kname, classes in (("k_front", front), ("k_mid", mid)):
        t = out[names.index(kname)]
        live = t[:, 0] != 0
        if not live.any():
            row.append("   (no stamps)"); continue
        t0 = t[live, 0].min(); lo = 0; cells = []
        for n, cnt in classes:
            e = t[lo:lo + cnt, 1]; e = e[e != 0]
            cells.append("%6.1f" % ((e.max() - t0) * 0.01) if e.size else "     -")
            lo += cnt
        row.append(" ".join(cells))
    pr = out[names.index("k_sep_self_solve")]; gk = pr[:, 6]
    st = s.stats()
    km = out[names.index("k_mid")]; t0 = km[km[:, 0] != 0, 0].min()
    lo = 0 if sf else U * P
    e = (km[lo:lo + 1024, 1] - t0) * 0.01
    top = np.argsort(-e)[:4]
    slow = " ".join(f"[{e[i]:.1f}us gjk {int(gk[lo + i] % 1000)}{'h' if gk[lo + i] >= 1000 else ' '} nt {int(pr[lo + i, 7])}]" for i in top)
    print(f"{it:4d} | {row[0]} | {row[1]} | {int((gk % 1000).max()):3d} | {st['head_starts'] - prev_hs:3d} | slowest pair blocks: {slow}")
    prev_hs = st["head_starts"]
