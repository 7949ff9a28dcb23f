#!/usr/bin/env python3
"""Development tool (timing build, GPU only): the phase stamps of the slowest robot-pair solves of one k_mid launch."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing_light.so" if os.environ.get("TJ_LIGHT") else "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
names = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(names), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
t = out[names.index("k_sep_self_solve")]; km = out[names.index("k_mid")]
live = t[:, 0] != 0
t0 = km[km[:, 0] != 0, 0].min()
idx = np.flatnonzero(live)
end = np.maximum(t[idx, 2], t[idx, 5])
order = idx[np.argsort(-(end - t0))[:12]]
print("k_mid: first block starts at 0; last stamp of any block %.2f us" % ((km[km[:, 1] != 0, 1].max() - t0) * 0.01))
print("block | gjk its, head start, newton | start, loaded, gjk done, offsets done, newton done, end (us from the kernel's first stamp) | k_mid block end")
for b in order:
    r = t[b]
    f = lambda x: "%6.2f" % ((x - t0) * 0.01) if x else "   -  "
    print(f"  {b:5d} | {int(r[6] % 1000):3d} {'hs' if r[6] >= 1000 else '  '} {int(r[7]):3d} | {f(r[0])} {f(r[1])} {f(r[3])} {f(r[4])} {f(r[5])} {f(r[2])} | {f(km[b, 1])}")
pf = out[names.index("k_obs_solve")]
print("GJK step profile of the same blocks (first work item): steps seg/tri/tet | us support, seg, tri, tet | where: xcc se cu simd | others on that SIMD (block: start..end)")
n_sl = 0 if os.environ.get("TJ_SLACK_FRONT", "1") != "0" else 320
def where(b):
    h = int(km[b, 2]); x = int(km[b, 3]) & 15
    return (x, (h >> 13) & 7, (h >> 8) & 15, (h >> 4) & 3)
loc = {}
for b in np.flatnonzero(km[:, 0] != 0):
    loc.setdefault(where(b), []).append(b)
for b in order:
    c = pf[b, 4:7]; tt = pf[b, 0:4] * 0.01
    kb = b   # the pair stamps are indexed by k_mid's block index
    oth = [o for o in loc.get(where(kb), []) if o != kb]
    f = lambda x: "%.1f" % ((x - t0) * 0.01)
    print(f"  {b:5d} | {c.tolist()} | " + " ".join(f"{x:6.2f}" for x in tt) + f" | {where(kb)} | " + ", ".join(f"{o}({'slack' if o < n_sl else 'pair' if o < n_sl + 1024 else 'obs'}): {f(km[o, 0])}..{f(km[o, 1])}" for o in oth[:6]))
from collections import Counter
print("blocks per SIMD:", sorted(Counter(len(v) for v in loc.values()).items()))
for lab, lo, hi in (("slack", 0, n_sl), ("pair", n_sl, n_sl + 1024), ("obs", n_sl + 1024, n_sl + 2048)):
    if hi <= lo: continue
    e = km[lo:hi, 1]; st = km[lo:hi, 0]; ok = e != 0
    print(f"{lab}: ends {np.sort((e[ok] - t0) * 0.01)[-5:]}  starts max {((st[ok] - t0) * 0.01).max():.2f}")
