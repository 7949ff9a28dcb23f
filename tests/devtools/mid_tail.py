"""Development tool (timing build, GPU only): which waves end k_mid in an iteration of the timed window -- the latest blocks of the launch with their class (slack / pair /
obstacle solve), for the pair waves the GJK iteration count of their first item (1000 + n: a wave dedicated to a head-start entry) and whether a plane came out, and the iteration
histograms of the pairs with and without a plane.   make -C traj-opt-admm_amd/csrc timing ;  python tests/devtools/mid_tail.py [iterations ...]
Output of round 6: profiles/round6_k_mid_tail_scnC.txt."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")
pkg = importlib.import_module("traj-opt-admm_amd")
sc = pkg.scenes
lib = C.CDLL(os.environ["TRAJADMM_LIB"])
lib.tj_kernel_name.restype = C.c_char_p
NAMES = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
for n_it in [int(a) for a in sys.argv[1:]] or [6, 9, 12, 16]:
    scene = sc.scn_c()
    s = pkg.Solver(scene, stop=0.0)
    s.iterate_async(n_it); s.sync()
    out = np.zeros((len(NAMES), 65536, 8), dtype=np.int64)
    lib.tj_debug_phase_times(s._ctx, out.ctypes.data)
    U, P = scene["U"], scene["P"]
    kf, km, ks, ko = (NAMES.index(x) for x in ("k_front", "k_mid", "k_sep_self_solve", "k_obs_solve"))
    f = out[kf]; live = f[:, 0] != 0; t0 = f[live, 0].min(); us = lambda x: (x - t0) * 0.01
    m = out[km]; sv = out[ks]; ob = out[ko]
    lm = m[:, 0] != 0
    fend = us(f[live, 1]).max()
    go = m[:, 4] != 0
    print(f"== after {n_it} iterations: k_front end {fend:.1f}; k_mid blocks {lm.sum()} start {us(m[lm,0]).min():.1f} end max {us(m[lm,1]).max():.1f}; go seen mean {us(m[go,4]).mean() if go.any() else -1:.1f}")
    n_sl = U * P; off = 1 if go.any() else 0
    idx = np.flatnonzero(lm); ends = us(m[idx, 1]); order = np.argsort(-ends)[:14]
    for k in order:
        b = idx[k]
        cls = "watch" if b < off else "slack" if b < off + n_sl else "pair" if b < off + n_sl + 1728 else "obs"
        extra = ""
        if cls == "pair":
            r = sv[b]
            extra = f" solve stamps 0:{us(r[0]) if r[0] else -1:.1f} 1:{us(r[1]) if r[1] else -1:.1f} 2:{us(r[2]) if r[2] else -1:.1f} gk {r[6]} nit {r[7]}"
        print(f"   block {b:5d} {cls:5s} start {us(m[b,0]):6.1f} end {ends[k]:6.1f}{extra}")
    for cls, a, b in (("slack", off, off + n_sl), ("pair", off + n_sl, off + n_sl + 1728), ("obs", off + n_sl + 1728, 65536)):
        sel = lm.copy(); sel[:a] = False; sel[b:] = False
        if sel.any(): print(f"   {cls:6s} n {sel.sum():5d} end mean {us(m[sel,1]).mean():6.1f} p90 {np.percentile(us(m[sel,1]),90):6.1f} max {us(m[sel,1]).max():6.1f}")
    pr = sv[:, 6] != 0
    gk = sv[pr, 6]
    ded = gk >= 1000
    print(f"   pair waves stamped {pr.sum()}, dedicated {ded.sum()} (gk of dedicated: {np.sort(gk[ded]-1000)[::-1][:20]}), others' gk top {np.sort(gk[~ded])[::-1][:20]}")
    nit = sv[pr, 7]
    far = (nit < 0) & ~ded
    near = (nit >= 0) & ~ded
    print(f"   first items of generic waves: no plane {far.sum()} (gk histogram {np.bincount(gk[far].astype(int), minlength=16)[:16]}), plane {near.sum()} (gk histogram {np.bincount(gk[near].astype(int), minlength=16)[:16]})")
    s.close()
