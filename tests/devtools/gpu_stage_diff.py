"""Teacher-forced, stage-by-stage comparison of the HIP path against the CPU oracle (diagnostic
script; the asserting versions live in tests/).  Usage: python tools/gpu_stage_diff.py <scene> <iters>"""
import sys, os, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
pkg = importlib.import_module("traj-opt-admm_amd")
sc = pkg.scenes
from oracle.pyoracle import Engine


def md(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def canon(counts, planes):
    """sort planes within each (robot, segment) so list order does not matter"""
    out = []; w = 0
    for n in counts.ravel():
        blk = planes[w:w + n]; w += n
        if n:
            idx = np.lexsort(blk.T[::-1]); blk = blk[idx]
        out.append(blk)
    return np.concatenate(out, axis=0) if out else planes


which = sys.argv[1]; n = int(sys.argv[2])
scene = {"H": sc.hard, "H8": lambda: sc.hard(8, 20000), "A": sc.scn_a, "B": sc.scn_b, "C": sc.scn_c,
         "t0": lambda: sc.tiny(0), "t1": lambda: sc.tiny(1)}[which]()
O = Engine("port", scene)
t0 = time.time(); G = pkg.Solver(scene, stop=0.0); print("gpu setup %.2fs" % (time.time() - t0), flush=True)
s0o, s0g = O.get_state(), G.get_state()
print("init state diff", {k: md(s0o[k], s0g[k]) for k in s0o})
for it in range(n):
    G.set_state(O.get_state())
    co, po = O.stage_planes(); cg, pg = G.stage_planes()
    same = np.array_equal(co, cg)
    pd = md(canon(co, po), canon(cg, pg)) if same else -1
    if not same:
        bad = np.argwhere(co != cg)
        print("   count mismatch at", bad[:5].tolist(), co[co != cg][:5], cg[co != cg][:5])
    G.set_planes(co, po)
    do = O.stage_direction(); dg = G.stage_direction()
    lg_diff = max(max(md(O.local_grad(u, sp)[0], G.local_grad(u, sp)[0]) for sp in range(O.P)) for u in range(min(O.U, 4)))
    so = O.stage_steps(); sg = G.stage_steps()
    lo = O.stage_linesearch(); lg = G.stage_linesearch()
    st_o, st_g = O.get_state(), G.get_state()
    ls_diff = max(md(st_o["spline"], st_g["spline"]), md(st_o["piece_time"], st_g["piece_time"]))
    G.set_state(st_o)
    O.stage_slack(); G.stage_slack()
    st_o, st_g = O.get_state(), G.get_state()
    sl_diff = max(md(st_o[k], st_g[k]) for k in st_o)
    print(f"it{it} planes={len(po)}/{len(pg)} same={same} pd={pd:.2e} lgrad={lg_diff:.2e} dir={md(do['direction'],dg['direction']):.2e} "
          f"tdir={md(do['t_direction'],dg['t_direction']):.2e} wolfe={md(do['wolfe'],dg['wolfe']):.2e} gn={md(do['gn'],dg['gn']):.2e} "
          f"self={md(so[0],sg[0]):.1e} pos={md(so[1],sg[1]):.1e} ls={md(lo,lg):.1e} state_ls={ls_diff:.2e} slack={sl_diff:.2e} "
          f"gnorm={do['gnorm']:.3g} minstep={min(so[0].min(), so[1].min()):.3g}", flush=True)
print("stats", G.stats())
