"""GPU (-m gpu): edge cases of the boundary -- empty / tiny inputs, capacity and infeasibility errors,
the drop-in CLIs on the reference's file formats.  All through the C ABI / the binaries."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, maxdiff, rel

pytestmark = pytest.mark.gpu
STATE = ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")


def _oracle_run(scene, n):
    from oracle.pyoracle import Engine
    e = Engine("port", scene)
    for _ in range(n):
        e.iterate()
    return e.get_state()


def test_empty_cloud_init_ob_0(pkg, scenes):
    """`init_ob: 0` in 3D.json skips the BVH build: the solver must run with zero obstacle points"""
    scene = dict(scenes.tiny(1)); scene["cloud"] = np.zeros((0, 3))
    s = pkg.Solver(scene, stop=0.0)
    s.iterate(4)
    want = _oracle_run(scene, 4)
    got = s.get_state()
    for k in STATE:
        assert maxdiff(got[k], want[k]) <= 1e-9 * max(1.0, np.abs(want[k]).max()), k
    st = s.stats()
    assert st["planes_obs"] == 0 and st["nodes_dcd"] == 0 and st["error_bits"] == 0


def test_tiny_cloud_fewer_points_than_a_bvh_leaf(pkg, scenes):
    scene = dict(scenes.tiny(1)); scene["cloud"] = np.ascontiguousarray(scene["cloud"][:5])
    s = pkg.Solver(scene, stop=0.0)
    s.iterate(3)
    want = _oracle_run(scene, 3)
    got = s.get_state()
    for k in STATE:
        assert maxdiff(got[k], want[k]) <= 1e-9 * max(1.0, np.abs(want[k]).max()), k


def test_single_robot_in_multi_mode_and_two_pieces(pkg, scenes):
    """U = 1 through the multi-UAV entry point (no pairs at all) and the smallest piece count"""
    base = scenes.tiny(1)
    scene = dict(base); scene["U"] = 1; scene["waypoints"] = np.ascontiguousarray(base["waypoints"][:1])
    s = pkg.Solver(scene, stop=0.0)
    s.iterate(5)
    want = _oracle_run(scene, 5)
    for k in STATE:
        assert maxdiff(s.get_state()[k], want[k]) <= 1e-9 * max(1.0, np.abs(want[k]).max()), k
    scene2 = dict(base); scene2["P"] = 2; scene2["waypoints"] = np.ascontiguousarray(base["waypoints"][:, :3])
    s2 = pkg.Solver(scene2, stop=0.0)
    s2.iterate(5)
    want2 = _oracle_run(scene2, 5)
    for k in STATE:
        assert maxdiff(s2.get_state()[k], want2[k]) <= 1e-9 * max(1.0, np.abs(want2[k]).max()), k


def test_plane_capacity_overflow_is_reported(pkg, scenes):
    s = pkg.Solver(scenes.hard(), stop=0.0, cap_obs=1)
    with pytest.raises(pkg.TrajAdmmError) as ei:
        s.iterate(1)
    assert "-3" in str(ei.value) and "cap_" in str(ei.value)          # TJ_ERR_CAPACITY, tells which knob


def test_refused_obstacle_set_leaves_the_previous_one_usable(pkg, scenes):
    """tj_set_mesh that is refused (triangles + optimal_plane:1 in single-UAV mode) must not touch the cloud that was set
    before: the next iterations run on it and give what an undisturbed solver gives; an invalid face index is refused likewise"""
    import ctypes as C
    scene = scenes.tiny(mode=0, U=1, n_points=500)
    a = pkg.Solver(scene, stop=0.0, optimal_plane=1)
    b = pkg.Solver(scene, stop=0.0, optimal_plane=1)
    verts = np.ascontiguousarray(np.random.default_rng(1).uniform(-1, 1, (30, 3)))
    faces = np.arange(30, dtype=np.int32).reshape(-1, 3)
    rc = a.lib.tj_set_mesh(a._ctx, verts.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(30), faces.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(10))
    assert rc == -5, rc                                               # TJ_ERR_UNSUPPORTED
    bad = faces.copy(); bad[3, 1] = 99
    rc = a.lib.tj_set_mesh(a._ctx, verts.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(30), bad.ctypes.data_as(C.POINTER(C.c_int)), C.c_int(10))
    assert rc == -1, rc                                               # TJ_ERR_INVALID
    a.iterate(3); b.iterate(3)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    a.close(); b.close()


def test_fleet_of_1500_robots_vs_oracle(pkg, scenes):
    """beyond the 1 024 robots of rounds 1-2 (robot ids in the packed pair keys are 11 bits now): 1 500 UAVs crossing, two whole
    iterations each started from the CPU oracle's state; plane counts equal (the pair set is exact), state at the full-size bars"""
    from oracle.pyoracle import Engine
    from conftest import observe_iteration, TOL_STATE_FULL, TOL_GNORM_FULL
    scene = scenes.crossing(1500, 6000, seed=9)
    o = Engine("port", scene); o2 = Engine("port", scene)
    s = pkg.Solver(scene, stop=0.0)
    for it in range(2):
        st0 = o.get_state()
        s.set_state(st0); o2.set_state(st0)
        co, _ = o2.stage_planes()
        go = o.iterate()
        gg, _, _ = s.iterate(1)
        cg, _ = s.get_planes()
        assert np.array_equal(co, cg), f"it{it}: plane counts differ"
        observe_iteration(s.get_state(), o.get_state(), gg, go, TOL_STATE_FULL, TOL_GNORM_FULL, it)
    st = s.stats()
    assert st["error_bits"] == 0 and st["order_unresolved"] == 0 and st["planes_self"] > 0
    s.close()
    with pytest.raises(pkg.TrajAdmmError) as ei:
        pkg.Solver(scenes.crossing(2049, 100, seed=1), stop=0.0)
    assert "-5" in str(ei.value) and "2048" in str(ei.value)


def test_infeasible_start_does_not_hang(pkg, scenes):
    """robots closer than `offset` at the start: the reference spins forever in Step::self_step
    (Step.h:232-250); the device loops are capped and the call returns TJ_ERR_NO_PROGRESS"""
    scene = scenes.hard(4, 2000, dz=0.03)
    s = pkg.Solver(scene, stop=0.0)
    with pytest.raises(pkg.TrajAdmmError) as ei:
        s.iterate(2)
    assert "-4" in str(ei.value)


def test_device_stop_test_matches_the_mains(pkg, scenes):
    """`iter>1 && gnorm<stop` evaluated on the device: a big batch stops at the same iteration as
    single-step calls driven from the host"""
    scene = scenes.scn_b()
    a = pkg.Solver(scene)
    g_a, it_a, conv_a = a.iterate(100)
    b = pkg.Solver(scene)
    it_b, conv_b = 0, False
    while not conv_b and it_b < 100:
        g_b, it_b, conv_b = b.iterate(1)
    assert conv_a and conv_b and it_a == it_b
    sa, sb = a.get_state(), b.get_state()
    for k in STATE:
        assert np.array_equal(sa[k], sb[k]), k                             # graph replay == step-by-step, bitwise
    # further calls after convergence change nothing
    a.iterate(5)
    for k in STATE:
        assert np.array_equal(a.get_state()[k], sa[k]), k


@pytest.mark.parametrize("multi", [False, True])
def test_cli_drop_in(pkg, scenes, tmp_path, multi):
    """admmPathPlanning3D / multiPathPlanning3D on the reference's working-directory layout"""
    scene = scenes.scn_b() if multi else scenes.tiny(0, n_points=3000)
    mesh = "x.obj"
    scenes.write_reference_files(scene, str(tmp_path), mesh)
    os.makedirs(tmp_path / "Config_File", exist_ok=True)
    (tmp_path / "Config_File" / "3D.json").write_text(
        '{"auto":0,"init":1,"gui":0,"optimal_plane":0,"decouple":1,"res":8,"vel_limit":2,"acc_limit":2,"lambda":1e1,'
        '"epsilon":1e-1,"margin":1e-1,"offset":1e-1,"stop":1e-2,"exit":0,"init_ob":1,"mu":0.1}')
    exe = os.path.join(ROOT, "traj-opt-admm_amd", "multiPathPlanning3D" if multi else "admmPathPlanning3D")
    r = subprocess.run([exe, mesh, "--dump-state", "state.txt", "--max-iter", "300"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    res = open(tmp_path / "result" / (mesh + ("_result_file_multi.txt" if multi else "_result_file_admm.txt"))).read().split("\n")
    assert res[0].startswith("iter: ") and res[1].startswith("running time: ") and res[2].startswith("point cloud size: ")
    iters = int(res[0].split()[1])
    assert int(res[2].split()[-1]) == scene["cloud"].shape[0]
    # same problem through the library: the CLI read the (x5-rescaled) files, so compare loosely
    lines = open(tmp_path / "state.txt").read().strip().split("\n")
    hdr = lines[0].split()
    assert int(hdr[1]) == scene["U"] and int(hdr[3]) == scene["P"] and int(hdr[9]) == 1
    s = pkg.Solver(scene)
    g, it, conv = s.iterate(300)
    assert conv and abs(it - iters) <= 1
    T = 3 * scene["P"] + 3
    cli_spline = np.array([[float(x) for x in l.split()] for l in lines if len(l.split()) == 3 and l[0] not in "u"]).reshape(scene["U"], T, 3)
    lib_spline = s.get_state()["spline"].transpose(0, 2, 1)
    assert rel(cli_spline, lib_spline) <= (1e-6 if multi else 1e-9)       # x0.2 / x5 file round trip is not bit exact


def test_cli_coupled_mode_and_log_data(pkg, scenes, tmp_path):
    """multiPathPlanning3D with "decouple":0 (Main/multiPathPlanning3D.cpp:674-677) and the log_data lines
    ("ccd time:", "ccd len:", Main/multiPathPlanning3D.cpp:33-77), checked against an independent numpy
    restatement of the sampling loop on the dumped control points"""
    from math import comb
    scene = scenes.scn_b()
    mesh = "y.obj"
    scenes.write_reference_files(scene, str(tmp_path), mesh)
    os.makedirs(tmp_path / "Config_File", exist_ok=True)
    (tmp_path / "Config_File" / "3D.json").write_text(
        '{"auto":0,"init":1,"gui":0,"optimal_plane":0,"decouple":0,"res":8,"vel_limit":2,"acc_limit":2,"lambda":1e1,'
        '"epsilon":1e-1,"margin":1e-1,"offset":1e-1,"stop":1e-2,"exit":0,"init_ob":1,"mu":0.1}')
    exe = os.path.join(ROOT, "traj-opt-admm_amd", "multiPathPlanning3D")
    r = subprocess.run([exe, mesh, "--dump-state", "state.txt", "--sample-traj", "traj.txt", "--max-iter", "300"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    iters = int(open(tmp_path / "result" / (mesh + "_result_file_multi.txt")).read().split("\n")[0].split()[1])
    g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_scn_b_coupled.npz"))
    assert abs(iters - int(g["iters"])) <= 1                                # the reference's coupled run of the same scene
    lines = open(tmp_path / "state.txt").read().strip().split("\n")
    U, P = scene["U"], scene["P"]
    T = 3 * P + 3
    pts = [float(l.split()[3]) for l in lines if l.startswith("uav ")]
    assert len(set(pts)) == 1                                              # one piece_time for every robot
    assert abs(pts[0] - float(g["final_piece_time"][0])) <= 1e-5 * pts[0]   # x0.2 / x5 file round trip is not bit exact
    spl = np.array([[float(x) for x in l.split()] for l in lines if len(l.split()) == 3 and l[0] not in "u"]).reshape(U, T, 3)
    conv, _, _, _ = pkg.host_tables(P, 8)
    times = [float(l.split(":")[1]) for l in r.stdout.split("\n") if l.startswith("ccd time:")]
    lens = [float(l.split(":")[1]) for l in r.stdout.split("\n") if l.startswith("ccd len:")]
    assert len(times) == U and len(lens) == U
    smp = np.array([[float(x) for x in l.split()] for l in open(tmp_path / "traj.txt").read().strip().split("\n")])
    for u in range(U):
        assert abs(times[u] - P * pts[0]) <= 1e-5 * P * pts[0]                             # stdout prints 6 significant digits
        coeff = np.stack([conv[i] @ spl[u, 3 * i:3 * i + 6] for i in range(P)])      # [P,6,3] Bezier points of each piece
        t, out = 0.0, []
        while t < P:
            i = int(np.floor(t)); ct = t - i
            w = np.array([comb(5, j) * ct ** j * (1 - ct) ** (5 - j) for j in range(6)])
            out.append(w @ coeff[i]); t += 0.1 / pts[0]
        out = np.array(out)
        want = np.linalg.norm(np.diff(out, axis=0), axis=1).sum()
        assert abs(lens[u] - want) <= 1e-5 * want                                      # stdout prints 6 significant digits
        mine = smp[smp[:, 0] == u][:, 2:]
        assert mine.shape == out.shape and np.max(np.abs(mine - out)) <= 1e-12


@pytest.mark.parametrize("P,res,mode", [(3, 8, 1), (4, 4, 1), (7, 8, 1), (8, 8, 1), (9, 8, 2), (3, 8, 2), (6, 8, 0), (10, 8, 1), (12, 8, 1), (20, 8, 1), (14, 8, 0), (31, 8, 1), (12, 8, 2), (16, 8, 2), (3, 16, 1), (3, 15, 1), (2, 16, 0)])
def test_piece_counts_and_resolutions_vs_oracle(pkg, scenes, P, res, mode):
    """every size class of the per-robot Newton system: n = 9P-2 in {25,...,61} takes the register-resident
    factorisation, P = 8..10 the dense LDS one, P > 10 (the reference sizes everything from the init file,
    Main/admmPathPlanning3D.cpp:249-353) the band-storage kernel with 4 or 2 Armijo candidates per round instead of 8;
    res != 8 changes the segment tables; all three modes.  Each iteration starts from the oracle's state (teacher-forced),
    tolerances as in test_gpu_parity.py"""
    from oracle.pyoracle import Engine
    if mode == 0:
        scene = scenes.scn_a(n_points=4000, pieces=P)
    else:
        scene = dict(scenes.hard(4, 3000, pieces=P)); scene["mode"] = mode
    params = {"res": res}
    o = Engine("port", scene, params)
    s = pkg.Solver(scene, params, stop=0.0)
    for it in range(6):
        s.set_state(o.get_state())
        go = o.iterate()
        gg, _, _ = s.iterate(1)
        a, b = s.get_state(), o.get_state()
        assert abs(gg - go) <= 1e-9 * max(1.0, go), (it, gg, go)
        for n in STATE:
            assert maxdiff(a[n], b[n]) <= 1e-9 * max(1.0, np.abs(b[n]).max()), (it, n, maxdiff(a[n], b[n]))
    assert s.stats()["error_bits"] == 0
    s.close()


@pytest.mark.parametrize("res", [15, 16])
def test_folded_gradient_equals_one_group_gradient_at_max_res(pkg, scenes, res, monkeypatch):
    """res = 15 / 16 segments per piece: 540 / 576 basis entries for the 512 threads of the folded k_grad (entries beyond 512 were
    never staged).  The folded launch and the one-group launch (TJ_GRAD_FOLD=0) use the same association: states bitwise equal."""
    scene = scenes.hard(4, 3000, pieces=3)
    a = pkg.Solver(scene, {"res": res}, stop=0.0)
    monkeypatch.setenv("TJ_GRAD_FOLD", "0")
    b = pkg.Solver(scene, {"res": res}, stop=0.0)
    for it in range(10):
        a.iterate(1); b.iterate(1)
        sa, sb = a.get_state(), b.get_state()
        for n in STATE:
            assert np.array_equal(sa[n], sb[n]), (it, n)
    assert a.stats()["error_bits"] == 0 and b.stats()["error_bits"] == 0
    a.close(); b.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_band_storage_solve_equals_dense_solve(pkg, scenes, mode, monkeypatch):
    """k_xsolve_band (long trajectories) keeps the operation order of the dense kernels: forced on at P = 9 (TJ_XS_BAND=1) it must
    reproduce the dense LDS factorisation's results bit for bit, PSD fallback included (the `hard` scene takes it)"""
    scene = scenes.scn_a(n_points=4000, pieces=9) if mode == 0 else scenes.hard(4, 3000, pieces=9)
    a = pkg.Solver(scene, stop=0.0)
    monkeypatch.setenv("TJ_XS_BAND", "1")
    b = pkg.Solver(scene, stop=0.0)
    for it in range(12):
        a.iterate(1); b.iterate(1)
        sa, sb = a.get_state(), b.get_state()
        for n in STATE:
            assert np.array_equal(sa[n], sb[n]), (it, n)
    assert a.stats()["llt_fail_robot"] == b.stats()["llt_fail_robot"]
    a.close(); b.close()


def test_long_trajectory_converges(pkg, scenes):
    """piece_num = 16 (128 segments per robot, decoupled): free-running to the stop test through the band-storage solve, same
    iteration count as the oracle, control points within the scene's own sensitivity (the unmodified reference moves its result by
    5.1e-6 under a 1-ulp input change on this scene, same 53 iterations; long single-UAV scenes are chaotic outright -- the
    reference needs 134 vs 159 iterations on scn_a with 12 pieces -- and are covered teacher-forced above)"""
    from oracle.pyoracle import Engine
    scene = scenes.crossing(4, 10000, seed=5, pieces=16, name="cross4-P16")
    o = Engine("port", scene)
    gn = []
    for it in range(200):
        gn.append(o.iterate())
        if it > 1 and gn[-1] < 1e-2:
            break
    assert gn[-1] < 1e-2
    s = pkg.Solver(scene)
    g, iters, conv = s.iterate(200)
    assert conv and iters == len(gn)
    assert rel(s.get_state()["spline"], o.get_state()["spline"]) <= 5e-6
    assert s.stats()["error_bits"] == 0
    s.close()


def test_many_obstacle_candidates_per_segment(pkg, scenes):
    """a robot skimming a dense strip of obstacle points: several hundred candidates per segment (many query batches,
    thousands of per-candidate solves per iteration) must give the oracle's planes and iterates; a candidate capacity
    that is too small is reported, not silently truncated"""
    from oracle.pyoracle import Engine
    from conftest import canon
    rng = np.random.default_rng(3)
    scene = dict(scenes.crossing(2, 1000, seed=5, name="dense"))
    x = rng.uniform(-10, 10, 40000); y = rng.uniform(-0.3, 0.3, 40000); z = rng.uniform(-0.19, -0.125, 40000)
    scene["cloud"] = np.ascontiguousarray(np.stack([x, y, z], 1))
    o = Engine("port", scene)
    s = pkg.Solver(scene, stop=0.0, cap_obs=1024)
    co, po = o.stage_planes(); cg, pg = s.stage_planes()
    assert co.max() > 500
    assert np.array_equal(co, cg)
    assert maxdiff(canon(co, po), canon(cg, pg)) <= 1e-13
    s.reset(); o = Engine("port", scene)
    for it in range(3):
        s.set_state(o.get_state())
        o.iterate(); s.iterate(1)
        a, b = s.get_state(), o.get_state()
        for n in STATE:
            assert maxdiff(a[n], b[n]) <= 1e-9 * max(1.0, np.abs(b[n]).max()), (it, n)
    s.close()
    s2 = pkg.Solver(scene, stop=0.0, cap_obs=64)
    with pytest.raises(pkg.TrajAdmmError) as ei:
        s2.iterate(1)
    assert "-3" in str(ei.value)


@pytest.mark.parametrize("coupled", [False, True])
def test_bench_two_processes_sharded_equals_single_process(tmp_path, coupled):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank, robots sharded,
    two all-gathers per iteration) on a 1-GPU box: both ranks share device 0 and the gathers go through host memory
    over gloo (RCCL refuses two ranks on one device).  The owned robots' final state must be bitwise what the
    single-process run produces."""
    import json
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "6", "--warmup", "2", "--scene", "B", "--no-cpu", "--state-checksum"] + (["--coupled"] if coupled else [])   # coupled: 6 phases, 5 gathers
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534" if coupled else "29533",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu"] + common, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    j = json.loads(two.stdout[two.stdout.index('{"metric"'):].split("\n")[0])
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["value"] > 0
    # the line validates itself: before timing, both ranks compared their robots with a one-rank run, bit for bit
    assert j["group"]["bitwise_equal_to_one_rank"] is True and j["group"]["validation"]["ranks_equal"] == 2 and j["group"]["validation"]["iterations"] == 8
    # decoupled mode: the DIRECT exchange between the two processes (receive blocks mapped through hipIpc, in-kernel pushes; ranks sharing a device wait in
    # a one-wave launch) is what ran and what validated; coupled mode keeps the collectives
    assert j["group"]["validation"]["timed_path"] == ("rccl" if coupled else "direct"), j["group"]["validation"]
    # single-process checksum covers all 8 robots; recompute the halves from a library run to compare per rank
    import hashlib
    import importlib
    pkg = importlib.import_module("traj-opt-admm_amd")
    s = pkg.Solver(dict(pkg.scenes.scn_b(), mode=2) if coupled else pkg.scenes.scn_b(), stop=0.0)
    s.iterate(2); s.reset(); s.iterate(6)            # bench: warmup, reset, K timed iterations
    st = s.get_state()
    want = {}
    for r in (0, 1):
        u0, u1 = r * 4, (r + 1) * 4
        want[r] = hashlib.sha256(np.ascontiguousarray(st["spline"][u0:u1]).tobytes() + np.ascontiguousarray(st["piece_time"][u0:u1]).tobytes()).hexdigest()
    import re
    got = {int(r): h for r, h in re.findall(r"CHECK (\d+) ([0-9a-f]{64})", two.stdout)}   # the ranks' lines may interleave
    assert got == want, (got, want)
    s.close()


def test_bench_group_line_validates_itself(tmp_path):
    """bench.py --group-devices 0,0 (tj_group, two ranks on the one GPU): every transport is tried in the order flag -> event -> rccl, each is
    compared bitwise with a one-rank run BEFORE anything is timed, and the line says what happened to each -- rccl is refused on repeated
    devices and the line carries the library's reason instead of the run dying on it"""
    import json
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TJ_GROUP_TRANSPORT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--group-devices", "0,0", "--steps", "4", "--warmup", "1", "--scene", "B", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads(r.stdout[r.stdout.index('{"metric"'):].split("\n")[0])
    g = j["group"]
    # (every transport that validated is timed, the line carries the fastest)
    assert j["value"] > 0 and g["bitwise_equal_to_one_rank"] is True and g["transport"] in ("flag", "event") and g["validation"]["timed_transport"] == g["transport"]
    assert set(g["validation"]["ms_per_step_by_transport"]) == {"flag", "event"}
    tr = g["validation"]["transports"]
    assert tr["flag"]["bitwise_equal_to_one_rank"] and tr["event"]["bitwise_equal_to_one_rank"]
    assert tr["rccl"]["ran"] is False and "own device" in tr["rccl"]["error"]
