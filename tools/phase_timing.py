#!/usr/bin/env python3
"""Phase breakdown inside the kernels of one iteration (development tool, GPU only).

Builds nothing itself: run `make -C traj-opt-admm_amd/csrc timing` first (compiles the library with
-DTJ_PHASE_TIMING into libtrajadmm_timing.so; thread 0 of every block stamps the 100 MHz wall clock
at its phase boundaries).  Usage:  python tools/phase_timing.py [--scene C] [--iter 10]
Prints, per instrumented kernel, mean / max phase times over blocks and the slowest blocks.
"""
import argparse, ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TRAJADMM_LIB"] = os.path.join(ROOT, "traj-opt-admm_amd", "libtrajadmm_timing.so")

PHASES = {
    "k_begin": ["hull", "planes", "velacc", "reduce", "consensus"],
    "k_grad": ["stage", "planes", "wait for B", "consensus", "psd", "store"],
    "k_sep_self_compact": ["(k_grad group B) records", "barrier", "accumulate"],
    "k_xsolve": ["load", "assemble", "chol+fwd", "backsolve", "finish", "swept-hull tail"],
    "k_linesearch": ["stage", "planes->lds", "setup", "E round0", "later rounds"],
    "k_sep_self_solve": ["load", "gjk+newton+store"],
    "k_sep_self_solve_inner": ["gjk", "normal+offsets", "newton"],
    "k_obs_query": ["hull+kdop", "walk", "leaves", "last cull", "work items"],
    "k_ccd_self_seq": ["stage counts", "segment loop", "k_self + gn stage", "gnorm"],
}
NAMES = []   # filled from tj_kernel_name() in main(): the library's own enumeration order

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="C")
    ap.add_argument("--iter", type=int, default=10)
    a = ap.parse_args()
    pkg = importlib.import_module("traj-opt-admm_amd")
    sc = pkg.scenes
    scene = {"A": sc.scn_a, "B": sc.scn_b, "C": sc.scn_c, "D": sc.scn_d, "Dtri": sc.scn_d_tri, "E": sc.scn_e}[a.scene]()
    s = pkg.Solver(scene, stop=0.0)
    s.iterate(a.iter)
    lib = C.CDLL(os.environ["TRAJADMM_LIB"])
    lib.tj_kernel_name.restype = C.c_char_p
    NAMES[:] = [lib.tj_kernel_name(i).decode() for i in range(lib.tj_kernel_count())]
    out = np.zeros((len(NAMES), 65536, 8), dtype=np.int64)
    lib.tj_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.tj_debug_phase_times(s._ctx, out.ctypes.data) == len(NAMES)
    km = NAMES.index("k_mid")
    t = out[km]; live = t[:, 0] != 0
    if live.any():
        t0 = t[live, 0].min()
        n_sl = scene["U"] * scene["P"]; n_pair = 1024 if scene["mode"] >= 1 else 0
        for lab, lo, hi in (("slack", 0, n_sl), ("pair solve", n_sl, n_sl + n_pair), ("obstacle solve", n_sl + n_pair, 65536)):
            sel = live.copy(); sel[:lo] = False; sel[hi:] = False
            if sel.any():
                st = (t[sel, 0] - t0) * 0.01; en = (t[sel, 1] - t0) * 0.01; du = en - st
                print(f"k_mid {lab:15s}: {sel.sum():5d} blocks  start {st.min():6.1f}..{st.max():6.1f}  end max {en.max():6.1f}  dur mean {du.mean():6.1f} max {du.max():6.1f} us")
    kf = NAMES.index("k_front")
    t = out[kf]; live = t[:, 0] != 0
    if live.any():
        t0 = t[live, 0].min(); n_obs = scene["U"] * scene["P"] * 8
        n_hs = 128 if (scene["mode"] >= 1 and os.environ.get("TJ_PAIR_HEAD_START", "1") != "0") else 0   # GJK head-start blocks lead the grid
        n_ord = (scene["U"] * scene["P"] + 63) // 64 if 256 < scene["U"] * scene["P"] < 512 else 0   # k_grad's launch-order blocks lead the grid (Dev::grad_bal; 256 compute units)
        for lab, lo, hi in (("k_grad order", 0, n_ord), ("gjk head start", n_ord, n_ord + n_hs), ("obstacle query", n_ord + n_hs, n_ord + n_hs + n_obs), ("pair rows", n_ord + n_hs + n_obs, 65536)):
            sel = live.copy(); sel[:lo] = False; sel[hi:] = False
            if sel.any():
                st = (t[sel, 0] - t0) * 0.01; en = (t[sel, 1] - t0) * 0.01; du = en - st
                print(f"k_front {lab:15s}: {sel.sum():5d} blocks  start {st.min():6.1f}..{st.max():6.1f}  end max {en.max():6.1f}  dur mean {du.mean():6.1f} max {du.max():6.1f} us")
                pc = np.percentile(du, [50, 90, 99, 99.9])
                print(f"    duration percentiles 50/90/99/99.9: {pc[0]:.1f} {pc[1]:.1f} {pc[2]:.1f} {pc[3]:.1f} us; blocks > 20 us: {(du > 20).sum()}, > 50 us: {(du > 50).sum()}; sum of durations {du.sum():.0f} us")
                idx = np.nonzero(sel)[0]
                top = np.argsort(-du)[:8]
                S = scene["P"] * 8
                print("    longest: " + ", ".join(f"blk {idx[i]} (u {(idx[i] - lo) // S if lab.startswith('obst') else -1}, seg {(idx[i] - lo) % S if lab.startswith('obst') else -1}) {du[i]:.0f}us@{st[i]:.0f}" for i in top))
    for k, name in enumerate(NAMES):
        if name not in PHASES:
            continue
        t = out[k]
        live = t[:, 0] != 0
        if not live.any():
            continue
        t = t[live]
        nph = len(PHASES[name])
        # a block that leaves early (helper blocks of k_linesearch after the DONE word, k_front's empty units, retired waves) has no later stamps: zeros there made
        # the differences garbage (round 4: "block total mean -35325103373.4").  Only blocks that left every stamp enter the phase statistics.
        full = (t[:, :nph + 1] != 0).all(axis=1)
        n_early = int((~full).sum())
        span = (t[:, :nph + 1].max() - t[:, 0].min()) * 0.01
        if not full.any():
            print(f"{name}: {live.sum()} blocks, all of them left before their last stamp; first start -> last stamp {span:.1f} us")
            continue
        live_idx = np.flatnonzero(live)[full]
        t = t[full]
        d = np.diff(t[:, :nph + 1], axis=1) * 0.01   # us
        tot = d.sum(1)
        print(f"{name}: {live.sum()} blocks ({n_early} left early and are not in the phase statistics), first start -> last stamp {span:.1f} us, block total mean {tot.mean():.1f} max {tot.max():.1f} us")
        print("   phase        " + " ".join(f"{p:>12s}" for p in PHASES[name]))
        print("   mean us      " + " ".join(f"{x:12.2f}" for x in d.mean(0)))
        print("   max us       " + " ".join(f"{x:12.2f}" for x in d.max(0)))
        if name == "k_sep_self_solve":
            hs = t[:, 6] >= 1000; gk, nit = t[:, 6] % 1000, t[:, 7]   # + 1000: the query was continued from a head start (k_front)
            print(f"   head starts continued: {int(hs.sum())}; queries of >= 5 iterations: {int((gk >= 5).sum())}, of them with a head start: {int((hs & (gk >= 5)).sum())}")
            print("   GJK iterations: hist", np.bincount(gk.astype(int), minlength=51)[[1,2,3,4,5,6,8,10,15,20,30,40,50]], "(at 1,2,3,4,5,6,8,10,15,20,30,40,50); mean", gk.mean(), "max", gk.max())
            print("   Newton iterations (accepted pairs): mean", nit[nit >= 0].mean() if (nit >= 0).any() else 0, "max", nit.max(), "rejected", int((nit < 0).sum()))
            print("   corr(total us, gjk iters) =", np.corrcoef(tot, gk)[0, 1], " corr(total us, newton) =", np.corrcoef(tot, np.maximum(nit, 0))[0, 1])
            o = np.argsort(-tot)[:8]
            print("   slowest (us, gjk iterations, newton, head start):", [(round(float(tot[i]), 1), int(gk[i]), int(nit[i]), bool(hs[i])) for i in o])
        if name == "k_sep_self_solve":
            acc = (t[:, 3] >= t[:, 1]) & (t[:, 4] >= t[:, 3]) & (t[:, 5] >= t[:, 4]) & (t[:, 2] >= t[:, 5])   # stamps of THIS iteration (rejected pairs leave older ones)
            if acc.any():
                a = t[acc]
                print(f"   accepted pairs ({acc.sum()}): gjk {((a[:,3]-a[:,1])*0.01).mean():.2f} (max {((a[:,3]-a[:,1])*0.01).max():.2f})  normal+offsets {((a[:,4]-a[:,3])*0.01).mean():.2f}  newton {((a[:,5]-a[:,4])*0.01).mean():.2f} (max {((a[:,5]-a[:,4])*0.01).max():.2f})  store+end {((a[:,2]-a[:,5])*0.01).mean():.2f} us")
                one = acc & (gk == 1); two = acc & (gk == 2)
                for lab, sel in (("1 GJK iteration", one), ("2 GJK iterations", two)):
                    if sel.any(): print(f"   {lab}: {sel.sum()} pairs, gjk phase {((t[sel,3]-t[sel,1])*0.01).mean():.2f} us")
        if name == "k_grad":
            llt = (t[:, 7] - t[:, 4]) * 0.01; rest = (t[:, 5] - t[:, 7]) * 0.01
            failed = rest > 2.0
            print(f"   psd split: LLT check mean {llt.mean():.2f} max {llt.max():.2f} us; after the check: {failed.sum()} blocks repair, mean {rest[failed].mean() if failed.any() else 0:.2f} max {rest.max():.2f} us")
        worst = np.argsort(-tot)[:3]
        for w in worst:
            print(f"   slow block {live_idx[w]:5d}: " + " ".join(f"{x:12.2f}" for x in d[w]))


if __name__ == "__main__":
    main()
