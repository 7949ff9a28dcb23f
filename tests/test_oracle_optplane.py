"""CPU: the oracle's `optimal_plane:1` branch against golden vectors the unmodified reference produced
(tests/golden/optplane_*.npz, written by tests/golden/make_golden.py --optplane-only).

The plane stage -- Optimal_plane::optimal_cd / self_optimal_cd, the persistent tables, the emitted lists -- is
pinned BIT-EXACT, including Eigen's SelfAdjointEigenSolver eigenvalue on 2x2 / 3x3 matrices.  Converged runs are
compared at the sensitivity the reference shows against itself in this mode."""
import numpy as np
import pytest

from conftest import canon, check_scene_matches_fixture, gold, rel

STATE = ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")


def _scene(scenes, name):
    if name == "tiny_single":
        return scenes.tiny(0, n_points=3000)
    sc = dict(scenes.tiny(1))
    if name.endswith("coupled"):
        sc["mode"] = 2
    return sc


def _unflat(n, ids, cd):
    out, w = [], 0
    for k in n:
        out.append((ids[w:w + k], cd[w:w + k])); w += k
    return out


def test_oracle_plane_refinement_known_answers_bit_exact():
    from oracle.pyoracle import Prims
    g = gold("optplane_kat.npz"); pr = Prims("port")
    for P, q, cin, cout in zip(g["P_obs"], g["q_obs"], g["in_obs"], g["out_obs"]):
        assert np.array_equal(pr.optimal_cd(P, q, cin), cout)
    for P, Q, cin, cout in zip(g["P_self"], g["Q_self"], g["in_self"], g["out_self"]):
        assert np.array_equal(pr.self_optimal_cd(P, Q, cin), cout)
    assert all(pr.min_eig_small(m) == e for m, e in zip(g["mats2"], g["eig2"]))
    assert all(pr.min_eig_small(m) == e for m, e in zip(g["mats3"], g["eig3"]))


@pytest.mark.parametrize("name", ["tiny_single", "tiny_multi", "tiny_multi_coupled"])
def test_oracle_persistent_plane_stage_bit_exact(scenes, name):
    from oracle.pyoracle import Engine
    g = gold(f"optplane_stages_{name}.npz"); scene = _scene(scenes, name)
    check_scene_matches_fixture(scene, g)
    e = Engine("port", scene); e.set_optimal_plane(True)
    for it in g["kept"]:
        k = f"it{it}_"
        e.set_state({n: g[k + "pre_" + n] for n in STATE})
        if scene["mode"] == 0:
            e.set_obs_cache(_unflat(g[k + "pre_cache_n"], g[k + "pre_cache_ids"], g[k + "pre_cache_cd"]))
        else:
            e.set_pair_cache(g[k + "pre_cache_on"], g[k + "pre_cache_cd"])
        counts, planes = e.stage_planes()
        assert np.array_equal(counts, g[k + "counts"])
        assert np.array_equal(canon(counts, planes), g[k + "planes"])
        if scene["mode"] == 0:
            n, ids, cd = [], [], []
            for i_, c_ in e.get_obs_cache():
                n.append(len(i_)); ids.extend(i_); cd.extend(c_)
            assert np.array_equal(np.array(n), g[k + "post_cache_n"]) and np.array_equal(np.array(ids, dtype=np.int32), g[k + "post_cache_ids"])
            assert np.array_equal(np.array(cd).reshape(-1, 4), g[k + "post_cache_cd"])
        else:
            on, cd = e.get_pair_cache()
            assert np.array_equal(on, g[k + "post_cache_on"]) and np.array_equal(cd, g[k + "post_cache_cd"])


@pytest.mark.parametrize("name,tol", [("tiny_single", 5e-6), ("tiny_multi", 1e-8)])
def test_oracle_converged_run_with_persistent_planes(scenes, name, tol):
    """same iteration count as the reference; final control points within the mode's own sensitivity (the single-UAV
    path, ks = 1e-8, moves by 1e-7-class amounts under 1-ulp perturbations even without this branch)"""
    from oracle.pyoracle import Engine
    g = gold(f"optplane_e2e_{name}.npz"); scene = _scene(scenes, name)
    check_scene_matches_fixture(scene, g)
    e = Engine("port", scene); e.set_optimal_plane(True)
    gn = []
    for it in range(200):
        gn.append(e.iterate())
        if it > 1 and gn[-1] < 1e-2:
            break
    assert len(gn) == int(g["iters"])
    assert rel(e.get_state()["spline"], g["final_spline"]) <= tol
