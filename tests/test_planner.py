"""Initial-trajectory planner (ompl_init / simplify_path / edge_collision without OMPL).

CPU: the oracle's restatement of the motion validator against decisions of the unmodified reference
(tests/golden/planner_kat.npz: BVH::EdgeCollision + CCD::GJKDCD on 4000 seeded edges, with and without prior edges).
GPU: the device predicate against the same vectors, and the planner's output contract -- every edge valid under the
reference's predicate, later robots clear of earlier paths, equal way-point counts, exact end points, deterministic,
and the ADMM solver converges from it.  The reference plans with OMPL's randomised RRTConnect: paths themselves are not
comparable."""
import os
import subprocess

import numpy as np
import pytest

from conftest import check_scene_matches_fixture, gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_motion_validator_vs_reference(scenes):
    from oracle.pyoracle import Engine
    g = gold("planner_kat.npz"); scene = scenes.scn_b()
    check_scene_matches_fixture(scene, g)
    e = Engine("port", scene)
    assert np.array_equal(e.edge_collision(g["edges"]), g["hit_cloud"])
    assert np.array_equal(e.edge_collision(g["edges"], g["prior"]), g["hit_all"])
    assert 0.2 < g["hit_cloud"].mean() < 0.8


def _wall_scene(scenes, U=3):
    """SCN-B's robots crossing, plus a wall of points across the middle with one gap: straight lines are blocked"""
    sc = dict(scenes.scn_b())
    rng = np.random.default_rng(11)
    y = rng.uniform(-12, 12, size=6000); z = rng.uniform(-1.0, 3.0, size=6000)
    keep = np.abs(y - 6.0) > 1.2                       # the gap
    wall = np.stack([rng.uniform(-0.3, 0.3, size=6000), y, z], axis=1)[keep]
    sc["cloud"] = np.concatenate([sc["cloud"], wall], axis=0)
    starts = np.array([[-9.0, -3.0 + 2.5 * u, 0.5 + 0.4 * u] for u in range(U)])
    goals = np.array([[9.0, 3.0 - 2.5 * u, 0.5 + 0.4 * u] for u in range(U)])
    return sc, starts, goals


@pytest.mark.gpu
def test_device_motion_validator_vs_reference(pkg, scenes):
    g = gold("planner_kat.npz"); scene = scenes.scn_b()
    s = pkg.Solver(scene, stop=0.0)
    assert np.array_equal(s.edge_collision(g["edges"]), g["hit_cloud"])
    assert np.array_equal(s.edge_collision(g["edges"], g["prior"]), g["hit_all"])
    s.close()


@pytest.mark.gpu
def test_planner_output_contract_and_solver_converges(pkg, scenes):
    from oracle.pyoracle import Engine
    sc, starts, goals = _wall_scene(scenes)
    U = len(starts)
    s = pkg.Solver(dict(sc, U=U, waypoints=sc["waypoints"][:U]), stop=0.0)
    wp = s.plan_init(starts, goals)
    assert np.array_equal(wp, s.plan_init(starts, goals))                       # deterministic
    assert wp.shape[0] == U and wp.shape[1] >= 6
    assert np.array_equal(wp[:, 0], starts) and np.array_equal(wp[:, -1], goals)
    assert np.abs(wp[:, :, 1] - 6.0).min() < 1.5                                 # went through the gap, not the wall
    o = Engine("port", dict(sc, U=U, waypoints=sc["waypoints"][:U]))           # the reference's predicate (oracle restatement)
    prior = np.zeros((0, 6))
    for u in range(U):
        edges = np.concatenate([wp[u, :-1], wp[u, 1:]], axis=1)
        edges = edges[np.linalg.norm(edges[:, :3] - edges[:, 3:], axis=1) > 0]
        assert not o.edge_collision(edges, prior).any(), u
        prior = np.concatenate([prior, edges], axis=0)
    s.close()
    # the planned way points are a usable initial trajectory: feasible at the start (finite barrier energy) ...
    scene = dict(sc, U=U, P=wp.shape[1] - 1, waypoints=wp)
    o = Engine("port", scene); o.stage_planes()
    assert all(np.isfinite(o.spline_energy(u)) for u in range(U)), [o.spline_energy(u) for u in range(U)]
    # ... and the solver converges from them
    slv = pkg.Solver(scene)
    gn, it, conv = slv.iterate(400)
    assert conv and np.isfinite(slv.get_state()["spline"]).all() and slv.stats()["error_bits"] == 0
    slv.close()


@pytest.mark.gpu
def test_planner_reports_blocked_start(pkg, scenes):
    sc = scenes.scn_b()
    s = pkg.Solver(sc, stop=0.0)
    inside = sc["cloud"][0]                                                      # a start on top of an obstacle point
    with pytest.raises(pkg.TrajAdmmError):
        s.plan_init([inside], [[9.0, 0.0, 0.5]], nodes=62)
    s.close()


@pytest.mark.gpu
def test_cli_init_2_plans_and_writes_the_init_file(pkg, scenes, tmp_path):
    """`"init":2`: way points planned from init/<mesh>_start_goal.txt, init/<mesh>_init_file.txt written in the reference's
    format (one line per way point, 3 numbers per robot), then the solve as usual"""
    sc, starts, goals = _wall_scene(scenes)
    mesh = "x.obj"
    scenes.write_reference_files(dict(sc, mode=1), str(tmp_path), mesh)          # cloud / 5 (the reader multiplies by 5)
    os.remove(tmp_path / "init" / (mesh + "_init_file.txt"))
    with open(tmp_path / "init" / (mesh + "_start_goal.txt"), "w") as f:
        for a, b in zip(starts, goals):
            f.write(" ".join("%.17g" % v for v in list(a) + list(b)) + "\n")
    os.makedirs(tmp_path / "Config_File", exist_ok=True)
    (tmp_path / "Config_File" / "3D.json").write_text(
        '{"auto":0,"init":2,"gui":0,"optimal_plane":0,"decouple":1,"res":8,"vel_limit":2,"acc_limit":2,"lambda":1e1,'
        '"epsilon":1e-1,"margin":1e-1,"offset":1e-1,"stop":1e-2,"exit":0,"init_ob":1,"mu":0.1}')
    exe = os.path.join(ROOT, "traj-opt-admm_amd", "multiPathPlanning3D")
    r = subprocess.run([exe, mesh, "--max-iter", "400"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout[-500:]
    rows = [l.split() for l in open(tmp_path / "init" / (mesh + "_init_file.txt")).read().strip().split("\n")]
    assert len(rows) >= 6 and all(len(x) == 3 * len(starts) for x in rows)
    first = np.array(rows[0], dtype=float).reshape(-1, 3); last = np.array(rows[-1], dtype=float).reshape(-1, 3)
    assert np.allclose(first, starts, rtol=1e-5) and np.allclose(last, goals, rtol=1e-5)
    assert open(tmp_path / "result" / (mesh + "_result_file_multi.txt")).read().startswith("iter: ")


@pytest.mark.gpu
def test_planner_at_scn_c_size_reproduces_the_straight_crossing(pkg, scenes):
    """64 robots through the 100k-point cloud of SCN-C: every straight start-goal line is free (the robots fly at
    different heights), so the planner must return exactly the evenly spaced straight way points of the scene --
    and 63 robots' edges are prior obstacles for the last one"""
    sc = scenes.scn_c()
    s = pkg.Solver(sc, stop=0.0)
    wp = s.plan_init(sc["waypoints"][:, 0], sc["waypoints"][:, -1])
    s.close()
    assert wp.shape == sc["waypoints"].shape
    assert np.max(np.abs(wp - sc["waypoints"])) <= 1e-12


@pytest.mark.gpu
def test_planner_without_cloud_and_second_robot_avoids_the_first(pkg, scenes):
    """`init_ob:0` (no obstacle cloud): the first robot gets its straight line, evenly split; a second robot whose straight
    line crosses it must leave the line (the first path is an obstacle with clearance offset + margin/2)"""
    sc = dict(scenes.scn_b(), cloud=np.zeros((0, 3)))
    s = pkg.Solver(sc, stop=0.0)
    starts = np.array([[-5.0, 0.0, 0.0], [0.0, -5.0, 0.0]]); goals = np.array([[5.0, 0.0, 0.0], [0.0, 5.0, 0.0]])
    wp = s.plan_init(starts, goals)
    assert wp.shape[0] == 2 and wp.shape[1] >= 6
    n = wp.shape[1]
    assert np.allclose(wp[0], starts[0] + np.linspace(0, 1, n)[:, None] * (goals[0] - starts[0]), atol=1e-12)
    d = s.params["offset"] + 0.5 * s.params["margin"]
    e0 = np.concatenate([wp[0, :-1], wp[0, 1:]], axis=1); e1 = np.concatenate([wp[1, :-1], wp[1, 1:]], axis=1)
    assert not s.edge_collision(e1, e0, d).any()                 # clear of the first robot's path
    assert s.edge_collision(np.concatenate([starts[1], goals[1]])[None], e0, d).all()   # the straight line is not
    s.close()
