"""Per-block phase stamps of the asynchronous front (development tool, GPU only): when k_front's obstacle units see their robot's commit flag, have formed and published
the hull record, walked and listed their work, when the pair tiles and head starts end, and when k_mid's waves see the go word -- all on one clock that starts with the first
k_front block.  Needs the timing build:  make -C traj-opt-admm_amd/csrc timing ;  python tools/front_async_phases.py [A|B|C|D|Dtri|E|H] [iterations]
Output of round 6: profiles/round6_fa_phase_stamps_scn{B,C}.txt."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
sc = pkg.scenes
name = sys.argv[1] if len(sys.argv) > 1 else "C"
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 25
scene = {"A": sc.scn_a, "B": sc.scn_b, "C": sc.scn_c, "D": sc.scn_d, "Dtri": sc.scn_d_tri, "E": sc.scn_e, "H": lambda: sc.hard(8, 8000)}[name]()
s = pkg.Solver(scene, stop=0.0)
s.iterate_async(n_it); s.sync()
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
NAMES = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
out = np.zeros((len(NAMES), 65536, 8), dtype=np.int64)
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
assert lib.tj_debug_phase_times(s._ctx, out.ctypes.data) == len(NAMES)
U, P = scene["U"], scene["P"]; S = P * 8
kf, km, ko, kl = (NAMES.index(x) for x in ("k_front", "k_mid", "k_obs_query", "k_linesearch"))
f = out[kf]; live = f[:, 0] != 0
t0 = f[live, 0].min()
us = lambda x: (x - t0) * 0.01
n_ord = (U * P + 63) // 64 if 256 < U * P < 512 else 0
n_hs = 128 if scene["mode"] >= 1 else 0
n_obs = U * S
print(f"scene {name}: k_front blocks {live.sum()}  (t = 0: first k_front block's start)")
for lab, lo, hi in (("grad order", 0, n_ord), ("head start", n_ord, n_ord + n_hs), ("obs units", n_ord + n_hs, n_ord + n_hs + n_obs), ("pair tiles", n_ord + n_hs + n_obs, 65536)):
    sel = live.copy(); sel[:lo] = False; sel[hi:] = False
    if sel.any():
        st, en = us(f[sel, 0]), us(f[sel, 1])
        print(f"  {lab:11s} {sel.sum():5d} blocks: start min/mean/max {st.min():6.1f} {st.mean():6.1f} {st.max():6.1f}   end min/mean/max {en.min():6.1f} {en.mean():6.1f} {en.max():6.1f}")
o = out[ko]
lo = n_ord + n_hs
sel = (o[:, 6] != 0)
if sel.any():
    fl = us(o[sel, 6]); print(f"  obs units: flag seen (slot 6) min/mean/max {fl.min():6.1f} {fl.mean():6.1f} {fl.max():6.1f}")
    for slot, lab in ((1, "hull+publish done"), (4, "walk+cull done"), (5, "work items done")):
        ok = sel & (o[:, slot] != 0)
        d = (o[ok, slot] - o[ok, 6]) * 0.01
        print(f"     {lab:18s} - flag seen: mean {d.mean():5.2f} max {d.max():5.2f}")
    # per robot: flag seen min (~ commit time)
    idx = np.flatnonzero(sel)
    rob = (idx - lo) // S
    per = [us(o[idx[rob == r], 6]).min() for r in range(U) if (rob == r).any()]
    print(f"  per-robot first 'flag seen': min {min(per):.1f} median {np.median(per):.1f} max {max(per):.1f}")
m = out[km]; livem = m[:, 0] != 0
if livem.any():
    st = us(m[livem, 0]); en = us(m[livem, 1])
    print(f"k_mid: {livem.sum()} blocks start min/max {st.min():6.1f} {st.max():6.1f} end max {en.max():6.1f}")
    g = m[:, 4] != 0
    if g.any():
        go = us(m[g, 4]); print(f"  go seen (slot 4): {g.sum()} waves min/mean/max {go.min():6.1f} {go.mean():6.1f} {go.max():6.1f}")
    n_sl = U * P
    off = 1 if g.any() else 0
    for lab, a, b in (("slack", off, off + n_sl), ("pair", off + n_sl, off + n_sl + 1728), ("obs solve", off + n_sl + 1728, 65536)):
        sel = livem.copy(); sel[:a] = False; sel[b:] = False
        if sel.any(): print(f"  {lab:10s} {sel.sum():5d}: start {us(m[sel,0]).min():6.1f}..{us(m[sel,0]).max():6.1f} end mean/max {us(m[sel,1]).mean():6.1f} {us(m[sel,1]).max():6.1f}")
l = out[kl]; livel = l[:, 0] != 0
if livel.any():
    print(f"k_linesearch (the NEXT pairing's launch, for scale): start {us(l[livel,0]).min():.1f}, stage end(2) mean {us(l[livel & (l[:,2]!=0), 2]).mean():.1f}, commit done (5) mean/max {us(l[livel & (l[:,5]!=0),5]).mean():.1f} {us(l[livel & (l[:,5]!=0),5]).max():.1f}, end (6) max {us(l[livel & (l[:,6]!=0),6]).max():.1f}")
