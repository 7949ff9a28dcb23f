"""CPU: the oracle restatement (oracle/liboracle.so) against golden vectors produced by the
unmodified reference (tests/golden/make_golden.py).  This is what pins the oracle on machines where
/root/reference does not exist."""
import numpy as np
import pytest

from conftest import canon, check_scene_matches_fixture, gold, maxdiff, rel, scene_by_name
from oracle.pyoracle import Engine, Prims


@pytest.mark.parametrize("P", [2, 5])
def test_tables_bit_exact(scenes, P):
    g = gold(f"tables_P{P}.npz")
    sc = dict(scenes.tiny(1)); sc["P"] = P; sc["waypoints"] = sc["waypoints"][:, :P + 1]
    e = Engine("port", sc)
    conv, M, basis = e.tables()
    assert np.array_equal(conv, g["convert"])      # C2 junction blocks (CCDUtils.h:137-170)
    assert np.array_equal(M, g["mdyn"])            # jerk Gram matrix, same rounding (CCDUtils.h:172-227)
    assert np.array_equal(basis, g["basis"])       # blossom subdivision x conversion
    assert np.array_equal(e.kdop_axes(), g["kdop"])


@pytest.mark.parametrize("shape", ["6v1", "6v6", "12v1", "12v12"])
def test_gjk_witness_bit_exact(shape):
    g = gold("gjk_kat.npz")
    pr = Prims("port")
    A, B, V = g[f"gjk_{shape}_a"], g[f"gjk_{shape}_b"], g[f"gjk_{shape}_v"]
    for a, b, v in zip(A, B, V):
        got = pr.gjk(a, b)
        assert np.array_equal(got, v) or (np.isnan(got).all() and np.isnan(v).all()), (shape, got, v)


def test_planes_kdop_ccd_primitives():
    g = gold("prims_kat.npz")
    pr = Prims("port")
    for i in range(len(g["P"])):
        ok, cd = pr.plane_obs(g["P"][i], g["q"][i], 0.2)
        assert ok == bool(g["plane_obs"][i, 0])
        if ok:
            assert np.array_equal(cd, g["plane_obs"][i, 1:])            # bitwise: same GJK path, same rounding
        ok, cd = pr.plane_self(g["P"][i], g["Q"][i], 0.3, refine=True)
        assert ok == bool(g["plane_self"][i, 0])
        if ok:
            assert np.array_equal(cd[:3], g["plane_self"][i, 1:4])
            want = g["plane_self"][i, 4]
            # Newton on the offset (libm log only).  When no barrier term is active the reference
            # divides 0/0 and returns NaN (Optimal_plane.h:63-66); that quirk is reproduced.
            assert (np.isnan(want) and np.isnan(cd[3])) or abs(cd[3] - want) <= 1e-14
        assert pr.kdop_dcd(g["P"][i], g["q"][i], 0.2) == bool(g["kdop_dcd"][i])
        assert pr.kdop_self_dcd(g["P"][i], g["Q"][i], 0.3) == bool(g["kdop_self_dcd"][i])
    for i in range(len(g["ccd_P"])):
        a, da, b, db, pt = g["ccd_P"][i], g["ccd_D"][i], g["ccd_Q"][i], g["ccd_E"][i], g["ccd_q"][i]
        t1, u1 = g["ccd_t"][i]
        assert pr.kdop_ccd(a, da, pt, 0.1, 0.0, t1) == bool(g["kdop_ccd"][i])
        assert pr.gjk_ccd(a, da, pt, 0.1, 0.0, t1) == bool(g["gjk_ccd"][i])
        assert pr.self_kdop_ccd(a, da, b, db, 0.1, t1, u1) == bool(g["self_kdop_ccd"][i])
        assert pr.self_gjk_ccd(a, da, b, db, 0.1, t1, u1) == bool(g["self_gjk_ccd"][i])


def test_dynamic_tree_pair_order():
    """the inter-robot step clamp is order dependent (Step.h:184-256): the oracle must reproduce the
    pair ORDER of the reference's incrementally balanced tree, not just the set"""
    g = gold("prims_kat.npz")
    pr = Prims("port")
    for lo, hi, pairs, n in zip(g["tree_lo"], g["tree_hi"], g["tree_pairs"], g["tree_npairs"]):
        got = pr.self_pairs(lo, hi, 0.1)
        assert np.array_equal(got, pairs[:n])


def test_llt_and_min_eigenvalue():
    g = gold("prims_kat.npz")
    pr = Prims("port")
    for m, f, ev in zip(g["llt_mats"], g["llt_fails"], g["min_eig"]):
        assert pr.llt_fails(m) == bool(f)
        assert abs(pr.min_eig(m) - ev) <= 1e-12 * max(1.0, np.abs(m).max())


@pytest.mark.parametrize("name", ["tiny_multi", "tiny_single", "hard"])
def test_stages_teacher_forced(scenes, name):
    """every stage of one ADMM iteration, started from the reference's own state (tol 1e-12 abs per
    SURVEY 8c; planes / gnorm / CCD steps are bit-exact)"""
    g = gold(f"stages_{name}.npz")
    scene = scene_by_name(scenes, name)
    check_scene_matches_fixture(scene, g)
    e = Engine("port", scene)
    for it in g["kept"]:
        k = f"it{it}_"
        e.set_state({n: g[k + "pre_" + n] for n in ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")})
        counts, planes = e.stage_planes()
        assert np.array_equal(counts, g[k + "counts"])
        assert np.array_equal(planes, g[k + "planes_raw"])       # same order, same bits
        d = e.stage_direction()
        assert maxdiff(d["gn"], g[k + "gn"]) <= 1e-12 * max(1.0, np.abs(g[k + "gn"]).max())  # |g|: whole gradient assembly
        assert maxdiff(d["direction"], g[k + "direction"]) <= 1e-10
        assert maxdiff(d["wolfe"], g[k + "wolfe"]) <= 1e-9 * max(1.0, np.abs(g[k + "wolfe"]).max())
        s_self, s_pos = e.stage_steps()
        assert np.array_equal(s_self, g[k + "step_self"]) and np.array_equal(s_pos, g[k + "step_pos"])
        arm = e.stage_linesearch()
        if scene["mode"] == 1:
            assert maxdiff(arm, g[k + "step_armijo"]) <= 1e-12
        st = e.get_state()
        assert maxdiff(st["spline"], g[k + "mid_spline"]) <= 1e-10
        e.set_state({n: (g[k + "mid_" + n] if n in ("spline", "piece_time") else g[k + "pre_" + n])
                     for n in ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")})
        e.stage_slack()
        st = e.get_state()
        for n in st:
            assert maxdiff(st[n], g[k + "post_" + n]) <= 1e-12 * max(1.0, np.abs(g[k + "post_" + n]).max()), (it, n)


@pytest.mark.parametrize("name", ["scn_b", "scn_a"])
def test_end_to_end_vs_reference(scenes, name):
    """free-running to the mains' stop test: same iteration count, final control points within 1e-8"""
    g = gold(f"e2e_{name}.npz")
    scene = scene_by_name(scenes, name)
    check_scene_matches_fixture(scene, g)
    e = Engine("port", scene)
    gn = []
    for it in range(200):
        gn.append(e.iterate())
        if it > 1 and gn[-1] < 1e-2:
            break
    assert len(gn) == int(g["iters"])
    st = e.get_state()
    assert rel(st["spline"], g["final_spline"]) <= 1e-8           # BASELINE.json tolerance
    assert rel(st["piece_time"], g["final_piece_time"]) <= 1e-8
    # the residual gradient norm is a difference of nearly cancelling terms: compare loosely
    assert abs(gn[-1] - g["gnorm_hist"][-1]) <= 1e-3 * g["gnorm_hist"][-1]


# ---- coupled mode ("decouple":0): Optimization3D_multi::optimization, one shared piece_time ----------------
STATE = ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")


def test_coupled_stages_teacher_forced(scenes):
    """planes -> update_spline (arrowhead Newton system, couple_self_step, Armijo on the summed energy) ->
    slack/dual, each started from the reference's own state"""
    g = gold("stages_hard_coupled.npz")
    scene = scene_by_name(scenes, "hard_coupled")
    check_scene_matches_fixture(scene, g)
    e = Engine("port", scene)
    for it in g["kept"]:
        k = f"it{it}_"
        e.set_state({n: g[k + "pre_" + n] for n in STATE})
        counts, planes = e.stage_planes()
        assert np.array_equal(counts, g[k + "counts"])
        assert np.array_equal(canon(counts, planes), g[k + "planes"])
        gnorm, wolfe = e.stage_update_spline()
        assert abs(gnorm - g[k + "gnorm"]) <= 1e-11 * max(1.0, float(g[k + "gnorm"]))
        assert abs(wolfe - g[k + "wolfe"]) <= 1e-9 * max(1.0, abs(float(g[k + "wolfe"])))
        st = e.get_state()
        assert maxdiff(st["spline"], g[k + "mid_spline"]) <= 1e-10
        assert maxdiff(st["piece_time"], g[k + "mid_piece_time"]) <= 1e-10
        e.set_state({n: (g[k + "mid_" + n] if n in ("spline", "piece_time") else g[k + "pre_" + n]) for n in STATE})
        e.stage_slack()
        st = e.get_state()
        for n in st:
            assert maxdiff(st[n], g[k + "post_" + n]) <= 1e-12 * max(1.0, np.abs(g[k + "post_" + n]).max()), (it, n)


def test_coupled_end_to_end_vs_reference(scenes):
    g = gold("e2e_scn_b_coupled.npz")
    scene = scene_by_name(scenes, "scn_b_coupled")
    check_scene_matches_fixture(scene, g)
    e = Engine("port", scene)
    gn = []
    for it in range(200):
        gn.append(e.iterate())
        if it > 1 and gn[-1] < 1e-2:
            break
    assert len(gn) == int(g["iters"])
    st = e.get_state()
    assert rel(st["spline"], g["final_spline"]) <= 1e-8
    assert rel(st["piece_time"], g["final_piece_time"]) <= 1e-8
    assert np.all(st["piece_time"] == st["piece_time"][0])   # one piece_time for every robot
