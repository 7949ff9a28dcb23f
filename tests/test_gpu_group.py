"""tj_group: the library sharding one problem over several ranks by itself (csrc/tj_group.h) -- one context and one host
thread per rank, peer stores + events for the exchanges.  A one-GPU box runs the ranks on the same device (the device list
may repeat), which exercises everything but the xGMI hop: the schedule, the receive-buffer parity, the event choreography
between the host threads."""
import subprocess
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
STATE = ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")


@pytest.mark.parametrize("mode,ranks", [(1, 2), (1, 3), (2, 2), (2, 4)])
def test_group_equals_one_context_bitwise(pkg, scenes, mode, ranks):
    """decoupled and coupled ("decouple":0) mode, 2-4 ranks, batches of different length (the exchange parity carries over
    between tj_group_iterate calls): every robot's state bitwise equal to the unsharded run after every batch"""
    scene = dict(scenes.hard(4, 4000)); scene["mode"] = mode
    ref = pkg.Solver(scene, stop=0.0)
    grp = pkg.Group(scene, [0] * ranks, stop=0.0)
    done = 0
    for batch in (1, 3, 2, 5):
        g0, _, _ = ref.iterate(batch)
        g, it, cv = grp.iterate(batch)
        done += batch
        assert it == done and not cv
        a, b = ref.get_state(), grp.get_state()
        for n in STATE:
            assert np.array_equal(a[n], b[n]), (n, done)
        assert g == g0
    ref.close(); grp.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_flag_transport_equals_one_context_bitwise(pkg, scenes, mode):
    """the device-to-device transport (peer stores + sequence flags polled by the consumer's next kernel, no host on the path),
    forced on two ranks that share device 0; switching transports between batches restarts the sequence numbers"""
    scene = dict(scenes.hard(4, 4000)); scene["mode"] = mode
    ref = pkg.Solver(scene, stop=0.0)
    grp = pkg.Group(scene, [0, 0], stop=0.0)
    assert grp.transport == "event"            # the default when devices repeat
    grp.set_transport("flag")
    assert grp.transport == "flag"
    done = 0
    for batch, transport in ((2, "flag"), (3, "flag"), (2, "event"), (4, "flag")):
        if grp.transport != transport:
            grp.set_transport(transport)
        g0, _, _ = ref.iterate(batch)
        g, it, cv = grp.iterate(batch)
        done += batch
        assert it == done and g == g0
        a, b = ref.get_state(), grp.get_state()
        for n in STATE:
            assert np.array_equal(a[n], b[n]), (n, done, transport)
    us = grp.profile_exchange(20)
    assert (us[:2] > 0).all() and (us[:2] < 1e5).all()
    a, b = ref.get_state(), grp.get_state()     # re-sending the slices changed nothing
    for n in STATE:
        assert np.array_equal(a[n], b[n]), n
    ref.close(); grp.close()


@pytest.mark.parametrize("poll,ranks", [("1", 2), ("0", 2), ("1", 3)])
def test_direct_exchange_is_the_fused_chain(pkg, scenes, poll, ranks, monkeypatch):
    """flag transport, decoupled mode = the DIRECT exchange: k_linesearch / k_begin push control points, k_xsolve pushes direction records,
    the foreign units at the head of the peers' k_front / k_ccd wait and rebuild the caches -- a sharded iteration is the six-kernel
    chain of one context.  TJ_XCH_POLL=1: the units poll the arrival counters themselves (what ranks on distinct devices do; a small
    fleet can do it on a shared device); 0: two one-wave wait launches per iteration (what ranks sharing a device do).  Bitwise the
    one-context states through batches of different length, the stop-free run, and a transport switch in between."""
    monkeypatch.setenv("TJ_XCH_POLL", poll)
    scene = scenes.hard(5 if ranks == 3 else 4, 4000)
    ref = pkg.Solver(scene, stop=0.0)
    grp = pkg.Group(scene, [0] * ranks, stop=0.0)
    grp.set_transport("flag")
    done = 0
    for batch, transport in ((1, "flag"), (4, "flag"), (3, "event"), (6, "flag"), (2, "flag")):
        if grp.transport != transport:
            grp.set_transport(transport)
        l0 = grp.launch_counts()
        g0, _, _ = ref.iterate(batch)
        g, it, cv = grp.iterate(batch)
        l1 = grp.launch_counts()
        done += batch
        assert it == done and g == g0 and not cv
        a, b = ref.get_state(), grp.get_state()
        for n in STATE:
            assert np.array_equal(a[n], b[n]), (n, done, transport)
        per_iter = max((y - x) for x, y in zip(l0, l1)) / batch
        if transport == "flag":   # six kernels (+ two wait launches on a shared device) per iteration; the batch adds k_begin, the flush and the slack update it pays
            assert per_iter <= (6 if poll == "1" else 8) + 4.0 / batch + 1e-9, per_iter   # (+ k_hullinfo after a host write, k_begin, k_flush + k_slack at the end of the batch)
        else:                     # event: the same six kernels (the slices travel by the in-kernel pushes), events order the streams
            assert per_iter <= 6 + 4.0 / batch + 1e-9, per_iter
    ref.close(); grp.close()


def test_rccl_transport_is_refused_on_repeated_devices(pkg, scenes):
    grp = pkg.Group(scenes.hard(4, 4000), [0, 0], stop=0.0)
    with pytest.raises(pkg.TrajAdmmError) as ei:
        grp.set_transport("rccl")
    assert "-5" in str(ei.value) and "own device" in str(ei.value)      # TJ_ERR_UNSUPPORTED, says why
    assert grp.transport == "event"
    grp.iterate(2)                                                       # the group is still usable
    grp.close()


@pytest.mark.parametrize("transport", ["flag", "event", "rccl"])
def test_group_on_two_devices(pkg, scenes, transport):
    """the configuration the feature exists for: two ranks on two DIFFERENT GPUs (peer stores over xGMI into uncached receive
    buffers, or RCCL all-gathers from host C++).  Skipped on a one-GPU box -- nothing else in this suite crosses a device."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    for mode in (1, 2):
        scene = dict(scenes.hard(4, 4000)); scene["mode"] = mode
        ref = pkg.Solver(scene, stop=0.0)
        grp = pkg.Group(scene, [0, 1], stop=0.0)
        assert grp.transport == "event"     # the default everywhere; flag is opt-in until it has passed HERE
        if transport != "event":
            grp.set_transport(transport)
        if transport == "rccl":
            assert grp.rccl_ranks == 2
        for batch in (1, 4, 7):
            g0, _, _ = ref.iterate(batch)
            g, _, _ = grp.iterate(batch)
            assert g == g0
            a, b = ref.get_state(), grp.get_state()
            for n in STATE:
                assert np.array_equal(a[n], b[n]), (n, mode, transport)
        ref.close(); grp.close()


def test_group_stop_test_and_uneven_partition(pkg, scenes):
    """5 robots over 2 and 3 ranks (uneven blocks); the device stop test ends the run on every rank in the same iteration"""
    scene = scenes.crossing(5, 3000, seed=4)
    ref = pkg.Solver(scene)
    g0, it0, cv0 = ref.iterate(200)
    assert cv0
    for ranks, transport in ((2, "event"), (3, "event"), (2, "flag"), (3, "flag")):
        grp = pkg.Group(scene, [0] * ranks)
        grp.set_transport(transport)
        g, it, cv = grp.iterate(200)
        assert (it, cv) == (it0, cv0) and g == g0
        a, b = ref.get_state(), grp.get_state()
        for n in STATE:
            assert np.array_equal(a[n], b[n]), n
        grp.close()
    ref.close()


def test_group_rejects_what_cannot_shard(pkg, scenes):
    with pytest.raises(pkg.TrajAdmmError):
        pkg.Group(scenes.tiny(mode=0, U=1), [0, 0])
    with pytest.raises(pkg.TrajAdmmError):
        pkg.Group(scenes.hard(4, 4000), [0] * 17)


def test_cli_devices_flag_gives_the_one_device_trajectory(scenes, tmp_path):
    """multiPathPlanning3D --devices 0,0,0 (three ranks): same iteration count and bitwise the same --dump-state file as the
    plain run"""
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scene = scenes.scn_b()
    mesh = "x.obj"
    scenes.write_reference_files(scene, str(tmp_path), mesh)
    os.makedirs(tmp_path / "Config_File", exist_ok=True)
    (tmp_path / "Config_File" / "3D.json").write_text(
        '{"auto":0,"init":1,"gui":0,"optimal_plane":0,"decouple":1,"res":8,"vel_limit":2,"acc_limit":2,"lambda":1e1,'
        '"epsilon":1e-1,"margin":1e-1,"offset":1e-1,"stop":1e-2,"exit":0,"init_ob":1,"mu":0.1}')
    exe = os.path.join(ROOT, "traj-opt-admm_amd", "multiPathPlanning3D")
    out = []
    for extra, name in (([], "a.txt"), (["--devices", "0,0,0"], "b.txt")):
        r = subprocess.run([exe, mesh, "--dump-state", name, "--max-iter", "300"] + extra, cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        out.append(open(tmp_path / name).read())
        if extra:
            assert "devices: 3" in r.stdout
    assert out[0] == out[1]
    r = subprocess.run([os.path.join(ROOT, "traj-opt-admm_amd", "admmPathPlanning3D"), mesh, "--gpus", "2"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 1 and "does not shard" in r.stderr


@pytest.mark.gpu
def test_passing_long_pair_solves_on_changes_no_bit(pkg, scenes, monkeypatch):
    """Large fleets: a lane stops its pair's GJK after 3 iterations and idle waves finish the pair with the wave-cooperative form
    (sep_self_solve_body).  Which wave solves a pair must not matter: the state after several iterations is bitwise the one
    of the run that keeps every pair on its lane (TJ_PAIR_PASS_ON=0), and the pairs counted as solved are the same."""
    scene = scenes.crossing(256, 20000, seed=31, name="crossing-U256-pass-on")
    monkeypatch.setenv("TJ_PAIR_PASS_ON", "0")
    a = pkg.Solver(scene, stop=0.0)
    monkeypatch.setenv("TJ_PAIR_PASS_ON", "1")
    b = pkg.Solver(scene, stop=0.0)
    a.iterate(5); b.iterate(5)
    sa, sb = a.get_state(), b.get_state()
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), f"{n} differs when long pair solves are passed on"
    ta, tb = a.stats(), b.stats()
    assert ta["pair_solves"] == tb["pair_solves"] and ta["newton_iters"] == tb["newton_iters"]
    assert ta["error_bits"] == 0 and tb["error_bits"] == 0
    assert ta["pair_solves"] / 5 > 4096, "the scene must be in the one-pair-per-lane regime"
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["scn_c", "hard8", "scn_b_coupled"])
def test_gjk_head_start_changes_no_bit(pkg, scenes, monkeypatch, scene_name):
    """k_front runs the first GJK iterations of the robot pairs that were slow in the previous iteration and k_mid continues the
    saved loop (kernels_pairs.h: spec_pair_body).  Where a query's iterations run must not matter: the state after many
    iterations is bitwise the one with TJ_PAIR_HEAD_START=0, the pair-solve statistics agree, and head starts were actually
    taken (the longest query of a launch is unchanged: the iteration count continues across the two kernels)."""
    scene = {"scn_c": scenes.scn_c, "hard8": lambda: scenes.hard(8, 20000), "scn_b_coupled": lambda: dict(scenes.scn_b(), mode=2)}[scene_name]()
    monkeypatch.setenv("TJ_PAIR_HEAD_START", "0")
    a = pkg.Solver(scene, stop=0.0)
    monkeypatch.setenv("TJ_PAIR_HEAD_START", "1")
    b = pkg.Solver(scene, stop=0.0)
    for n_it in (1, 2, 9, 28):   # several batches: the list of slow pairs crosses batch boundaries
        a.iterate(n_it); b.iterate(n_it)
        sa, sb = a.get_state(), b.get_state()
        for n in sa:
            assert np.array_equal(sa[n], sb[n]), f"{n} differs with the GJK head start after a batch of {n_it}"
    ta, tb = a.stats(), b.stats()
    assert ta["pair_solves"] == tb["pair_solves"] and ta["newton_iters"] == tb["newton_iters"] and ta["gjk_max_sum"] == tb["gjk_max_sum"]
    assert ta["error_bits"] == 0 and tb["error_bits"] == 0
    assert ta["head_starts"] == 0 and tb["head_starts"] > 0, "no head start was ever continued: the test compares nothing"
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2], ids=["decoupled", "coupled"])
def test_round4_launch_shapes_over_a_long_run(pkg, scenes, monkeypatch, mode):
    """150 iterations of a fleet whose robots really back off (4 robots stacked 0.13 apart: tens of Armijo steps per iteration at the start, PSD repairs,
    acting robot pairs) with every launch shape of round 4 on -- helper blocks and super-rounds in k_linesearch, the quiet counter switching them off and on
    again, the one-launch coupled search with its folded commit and begin -- against all of them off: same bits, same evaluation counts, no error bit.
    (tests/devtools/soak_round4.py runs the same comparison for 300 iterations on eight scenes.)"""
    scene = dict(scenes.hard(), mode=mode)
    for k in ("TJ_LS_HELP", "TJ_GRAD_BALANCE", "TJ_LSC_WIDE"):
        monkeypatch.delenv(k, raising=False)
    a = pkg.Solver(scene, stop=0.0)
    monkeypatch.setenv("TJ_LS_HELP", "1"); monkeypatch.setenv("TJ_GRAD_BALANCE", "0"); monkeypatch.setenv("TJ_LSC_WIDE", "0")
    b = pkg.Solver(scene, stop=0.0)
    for _ in range(3):
        a.iterate(50); b.iterate(50)
        sa, sb = a.get_state(), b.get_state()
        for n in sa:
            assert np.array_equal(sa[n], sb[n]), f"{n} differs"
    ta, tb = a.stats(), b.stats()
    assert ta["error_bits"] == 0 and tb["error_bits"] == 0
    assert ta["energy_evals"] == tb["energy_evals"]
    assert ta["energy_evals"] > 150 * scene["U"] * 2, "no robot ever backed off: the run compares nothing"
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["scn_c", "hard64"])
def test_helper_blocks_that_start_late_change_no_bit(pkg, scenes, monkeypatch, which):
    """k_linesearch's helper protocol, the case its late-start guard exists for: TJ_LS_HELP_LATE=<us> makes every helper block idle that long
    before it stages its robot -- by then the primary has decided, stored DONE and (in most iterations) committed the new control net.  A late helper must
    either see DONE and leave or have staged the state of before the commit; what it posts is never looked at.  150 iterations, four delays from "inside the
    primary's first super-round" to "long after its commit": states bitwise those of a run without helpers, no error bit, and the helpers really were there
    (give-ups counted: a primary that finds no post after 10 us searches on alone).  TJ_LS_HELP_MUTE=1 (helpers that never post) likewise."""
    scene = scenes.scn_c() if which == "scn_c" else scenes.hard(64, 20000)
    for k in ("TJ_LS_HELP", "TJ_LS_HELP_LATE", "TJ_LS_HELP_MUTE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TJ_LS_HELP", "1")
    ref = pkg.Solver(scene, stop=0.0)
    ref.iterate(150)
    want = ref.get_state()
    ref.close()
    monkeypatch.delenv("TJ_LS_HELP")
    base = pkg.Solver(scene, stop=0.0)
    base.iterate(150)
    sb, tb = base.get_state(), base.stats()
    base.close()
    for n in want:
        assert np.array_equal(want[n], sb[n]), n
    assert tb["error_bits"] == 0 and tb["ls_helper_timeouts"] == 0
    for env in ({"TJ_LS_HELP_LATE": "2"}, {"TJ_LS_HELP_LATE": "6"}, {"TJ_LS_HELP_LATE": "15"}, {"TJ_LS_HELP_LATE": "60"}, {"TJ_LS_HELP_MUTE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = pkg.Solver(scene, stop=0.0)
        s.iterate(150)
        st, ts = s.get_state(), s.stats()
        s.close()
        for k in env:
            monkeypatch.delenv(k)
        for n in want:
            assert np.array_equal(want[n], st[n]), (n, env)
        assert ts["error_bits"] == 0 and ts["ls_helper_timeouts"] == 0, env
        assert ts["energy_evals"] == tb["energy_evals"], env
        if "TJ_LS_HELP_MUTE" in env or int(env.get("TJ_LS_HELP_LATE", 0)) >= 15:
            assert ts["ls_giveups"] > 0, (env, "the late / mute helpers were never waited for: the run proves nothing")


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["scn_c", "scn_a", "hard"])
def test_asynchronous_newton_solve_changes_no_bit(pkg, scenes, monkeypatch, scene_name):
    """Round 5: in the one-context chain k_xsolve runs on a second hardware queue next to k_grad (tickets per robot, a one-wave gate in front of the launch) and k_ccd's
    units wait for the robots' flags and build the swept-hull records themselves.  Against the one-queue chain (TJ_XS_ASYNC=0): same state bit for bit over 60 iterations
    from the initial trajectory (through the back-off regime into the steady one), no error bit, the same number of energy evaluations -- and the launch count shows that
    the second queue was really in use (one gate launch per iteration on top of the six kernels)."""
    scene = {"scn_c": scenes.scn_c, "scn_a": scenes.scn_a, "hard": lambda: scenes.hard(8, 8000)}[scene_name]()   # hard: robots that meet -- CCD candidates, acting pairs, the replay
    for k in ("TJ_XS_ASYNC", "TJ_XS_ONE_QUEUE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("TJ_FRONT_ASYNC", "0")   # (round 6's asynchronous front rides on the second queue too and has a gate of its own: its test is below)
    n_it = 60
    a = pkg.Solver(scene, stop=0.0)
    l0 = a.launch_count(); a.iterate_async(n_it); a.sync(); la = a.launch_count() - l0
    sa, ta = a.get_state(), a.stats()
    a.close()
    monkeypatch.setenv("TJ_XS_ASYNC", "0")
    b = pkg.Solver(scene, stop=0.0)
    l0 = b.launch_count(); b.iterate_async(n_it); b.sync(); lb = b.launch_count() - l0
    sb, tb = b.get_state(), b.stats()
    b.close()
    monkeypatch.delenv("TJ_XS_ASYNC")
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), f"{n} differs between the asynchronous solve and the one-queue chain"
    assert ta["error_bits"] == 0 and tb["error_bits"] == 0
    assert ta["energy_evals"] == tb["energy_evals"]
    assert la == lb + n_it, f"expected one gate launch per iteration on top of the chain ({lb} launches): {la}"


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["scn_c", "scn_b", "hard", "fleet100", "scn_a", "scn_b_coupled", "hard_coupled"])
def test_asynchronous_front_changes_no_bit(pkg, scenes, monkeypatch, scene_name):
    """Round 6: inside a batch the NEXT iteration's k_front runs on the second hardware queue next to k_linesearch (residency gate, per-robot commit flags behind the
    written-through control nets, units that form and publish the hull records, done counters), and -- where k_front's whole grid is resident at once -- k_mid starts while
    k_front still runs and waits for it in its solve waves (watcher block, go words, reads past the caches).  Against TJ_FRONT_ASYNC=0 (k_linesearch publishes the hull cache,
    k_front follows on the chain's queue) and against TJ_FRONT_ASYNC_MID=0 (k_linesearch waits for k_front's end): the same state bit for bit over 60 iterations in batches of
    uneven length, no error bit, the same number of energy evaluations; the launch count shows the gate (one per pairing = per iteration that has a successor in its batch).
    fleet100: 100 robots -- one k_linesearch block per robot, k_front's grid too large to be resident at once (k_mid follows plainly)."""
    scene = {"scn_c": scenes.scn_c, "scn_b": scenes.scn_b, "hard": lambda: scenes.hard(8, 8000), "fleet100": lambda: scenes.crossing(100, 20000, seed=121), "scn_a": scenes.scn_a,
             "scn_b_coupled": lambda: dict(scenes.scn_b(), mode=2), "hard_coupled": lambda: dict(scenes.hard(8, 8000), mode=2)}[scene_name]()   # all three modes (coupled: the one-launch search commits every robot and raises every flag)
    for k in ("TJ_XS_ASYNC", "TJ_XS_ONE_QUEUE", "TJ_FRONT_ASYNC", "TJ_FRONT_ASYNC_MID", "TJ_FRONT_ASYNC_ONE_QUEUE"):
        monkeypatch.delenv(k, raising=False)
    batches = (1, 7, 20, 2, 30)
    def run():
        s = pkg.Solver(scene, stop=0.0)
        l0 = s.launch_count()
        for b in batches:
            s.iterate_async(b); s.sync()
        n = s.launch_count() - l0
        st, ts = s.get_state(), s.stats()
        s.close()
        return st, ts, n
    sa, ta, la = run()
    monkeypatch.setenv("TJ_FRONT_ASYNC_MID", "0")
    sm, tm, lm = run()
    monkeypatch.delenv("TJ_FRONT_ASYNC_MID")
    monkeypatch.setenv("TJ_XS_ONE_QUEUE", "1"); monkeypatch.setenv("TJ_FRONT_ASYNC_ONE_QUEUE", "1")   # the schedule's data flow on one queue (what the counter passes run)
    se, te, le = run()
    monkeypatch.delenv("TJ_XS_ONE_QUEUE"); monkeypatch.delenv("TJ_FRONT_ASYNC_ONE_QUEUE")
    monkeypatch.setenv("TJ_FRONT_ASYNC", "0")
    sb, tb, lb = run()
    for n in sa:
        assert np.array_equal(se[n], sb[n]), f"{n} differs between the one-queue emulation of the asynchronous front and the one-queue front"
    assert te["error_bits"] == 0 and te["energy_evals"] == tb["energy_evals"]
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), f"{n} differs between the asynchronous front and the one-queue front"
        assert np.array_equal(sm[n], sb[n]), f"{n} differs between the asynchronous front (k_linesearch waits for its end) and the one-queue front"
    assert ta["error_bits"] == 0 and tm["error_bits"] == 0 and tb["error_bits"] == 0
    assert ta["energy_evals"] == tb["energy_evals"] == tm["energy_evals"]
    pairings = sum(b - 1 for b in batches)
    assert la == lb + pairings and lm == lb + pairings, f"expected one gate launch per pairing ({pairings}) on top of {lb} launches: {la}, {lm}"


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,opt", [("scn_b", 0), ("scn_c", 0), ("scn_b", 1)])
def test_cross_queue_timeout_heals_itself(pkg, scenes, monkeypatch, scene_name, opt):
    """A wait between the queues of a context that runs out (error bit 2048: in practice a second process on the GPU) must not fail the run: the library abandons the batch
    (what is still enqueued returns at once), restores the state the batch started from, runs the same iterations again on ONE queue, and keeps the one-queue chain from
    then on.  Driven by the test hook TJ_XS_FAULT=n (the n-th gate of the asynchronous solve reports a time-out without waiting for it): the healed run ends in the same state
    bit for bit as an undisturbed one, with the same counters, no error bit, one fallback counted -- in the middle of the second of three batches, with the asynchronous front
    and (opt: "optimal_plane":1) the asynchronous plane refinement and its persistent tables in play."""
    scene = {"scn_b": scenes.scn_b, "scn_c": scenes.scn_c}[scene_name]()
    for k in ("TJ_XS_ASYNC", "TJ_XS_ONE_QUEUE", "TJ_FRONT_ASYNC", "TJ_HEAL", "TJ_XS_FAULT", "TJ_KEEP_ASYNC"):
        monkeypatch.delenv(k, raising=False)
    def run():
        s = pkg.Solver(scene, stop=0.0, optimal_plane=1) if opt else pkg.Solver(scene, stop=0.0)
        for b in (7, 20, 9):
            s.iterate_async(b); s.sync()
        st, ts = s.get_state(), s.stats()
        s.close()
        return st, ts
    sa, ta = run()
    monkeypatch.setenv("TJ_XS_FAULT", "12")   # the gate of the 12th iteration = the fifth of the second batch
    sb, tb = run()
    assert ta["async_fallbacks"] == 0 and tb["async_fallbacks"] == 1, (ta["async_fallbacks"], tb["async_fallbacks"])
    assert ta["error_bits"] == 0 and tb["error_bits"] == 0, (ta["error_bits"], tb["error_bits"])
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), f"{n}: the healed run differs from the undisturbed one"
    assert ta["iters"] == tb["iters"] == 36 and ta["energy_evals"] == tb["energy_evals"] and ta["planes_obs"] == tb["planes_obs"] and ta["planes_self"] == tb["planes_self"]
    # TJ_HEAL=0: the incident is reported as in round 5
    monkeypatch.setenv("TJ_HEAL", "0")
    s = pkg.Solver(scene, stop=0.0)
    s.iterate_async(20)
    with pytest.raises(pkg.TrajAdmmError):
        s.iterate(1)
    s.close()


_SHARE_CODE = r"""
import sys, os, importlib, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("traj-opt-admm_amd"); sc = pkg.scenes
s = pkg.Solver(sc.scn_c(), stop=0.0)
time.sleep(max(0.0, float(sys.argv[1]) - time.time()))   # both processes start their batches at the same wall-clock time
t0 = time.time()
for _ in range(4):
    s.iterate_async(50); s.sync()
st, ts = s.get_state(), s.stats()
np.savez(sys.argv[2], **st)
print("RESULT", ts["error_bits"], ts["async_fallbacks"], round(time.time() - t0, 3), flush=True)
"""


@pytest.mark.gpu
def test_two_processes_on_one_gpu_with_default_settings(pkg, scenes, tmp_path):
    """Round 5's probe as a test: two planner PROCESSES on one GPU, default settings (both sleep across queues; this device runs one process's waves at a time, so they shut
    each other out until the 2 s limits fire: round 5 ended in 13 s and error bits 2052 there).  Now each process heals itself: no error bit, and the same state bit for bit
    as a process that has the GPU to itself.  (Whether an incident happens at all is up to the scheduler -- counted and printed, not asserted; the deterministic test is
    test_cross_queue_timeout_heals_itself.)"""
    import subprocess, sys, time
    env = dict(os.environ)
    for k in ("TJ_XS_ASYNC", "TJ_FRONT_ASYNC", "TJ_HEAL", "TJ_XS_FAULT"):
        env.pop(k, None)
    go = time.time() + 8.0
    ps = [subprocess.Popen([sys.executable, "-c", _SHARE_CODE, str(go), str(tmp_path / f"p{i}.npz")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for i in range(2)]
    outs = []
    for p in ps:
        out, _ = p.communicate(timeout=240)
        outs.append(out)
    ref = pkg.Solver(scenes.scn_c(), stop=0.0)
    for _ in range(4):
        ref.iterate_async(50); ref.sync()
    sr = ref.get_state(); ref.close()
    for i, out in enumerate(outs):
        line = [l for l in out.splitlines() if l.startswith("RESULT")]
        assert line, f"process {i} failed:\n{out[-2000:]}"
        err, fb, secs = line[0].split()[1:4]
        print(f"process {i}: error bits {err}, fallbacks {fb}, {secs} s")
        assert int(err) == 0, out[-2000:]
        got = np.load(tmp_path / f"p{i}.npz")
        for n in sr:
            assert np.array_equal(got[n], sr[n]), f"process {i}: {n} differs from a process that has the GPU to itself"


@pytest.mark.gpu
def test_several_contexts_of_one_process_run_their_two_queue_schedules_side_by_side(pkg, scenes, monkeypatch):
    """Three contexts of ONE process, default settings, batches enqueued on all of them before any is waited for: each has its two hardware queues, its gates, flags and
    counters; their kernels share the device.  Every wait of the cross-queue protocols has its producer resident when it begins, so contexts cannot block each other for
    good -- and if a 2 s limit fires all the same, the context heals itself.  Each must end bit for bit where it ends alone, without an error bit."""
    for k in ("TJ_XS_ASYNC", "TJ_FRONT_ASYNC", "TJ_HEAL", "TJ_XS_FAULT", "TJ_XS_ONE_QUEUE", "TJ_FRONT_ASYNC_ONE_QUEUE"):
        monkeypatch.delenv(k, raising=False)
    makers = (scenes.scn_c, scenes.scn_b, lambda: dict(scenes.scn_b(), mode=2))
    alone = []
    for mk in makers:
        s = pkg.Solver(mk(), stop=0.0)
        for b in (25, 1, 34):
            s.iterate_async(b); s.sync()
        alone.append(s.get_state()); s.close()
    ctxs = [pkg.Solver(mk(), stop=0.0) for mk in makers]
    for b in (25, 1, 34):
        for s in ctxs:
            s.iterate_async(b)
        for s in reversed(ctxs):
            s.sync()
    for i, s in enumerate(ctxs):
        st, ts = s.get_state(), s.stats()
        print(f"context {i}: error bits {ts['error_bits']}, fallbacks {ts['async_fallbacks']}")
        assert ts["error_bits"] == 0 and ts["iters"] == 60, (i, ts["error_bits"], ts["iters"])
        for n in st:
            assert np.array_equal(st[n], alone[i][n]), f"context {i}: {n} differs from the run that had the device to itself"
        s.close()


@pytest.mark.gpu
def test_rank_isolated_replay_reproduces_the_one_context_run(tmp_path):
    """tools/rank_replay.py (round 6): rank r of world N alone on the device, the other ranks' slices copied in from a one-context recording before each consuming phase.
    The tool asserts that every replayed rank's owned robots end bit for bit in the one-context state; here on the 8-robot scene with 1, 2 and 4 ranks."""
    import json, sys
    out = tmp_path / "replay.json"
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rank_replay.py"), "--scene", "B", "--worlds", "1,2,4", "--steps", "6",
                        "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.load(open(out))
    assert set(d["worlds"]) == {"1", "2", "4"} and all(len(w["ranks"]) == int(n) for n, w in d["worlds"].items())


@pytest.mark.gpu
def test_large_fleet_launch_shapes_change_no_bit(pkg, scenes, monkeypatch):
    """256 robots x 1 M obstacle points (the one-pair-per-lane path of k_mid, the deep BVH): the launch-shape switches of round 5 -- pairs per producer wave
    (TJ_PAIR_LPW, the LDS tile of their hulls), producer priority, two BVH levels per step -- against the defaults: three iterations, states bitwise equal"""
    scene = scenes.scn_d()
    for k in ("TJ_PAIR_LPW", "TJ_PAIR_PRIO", "TJ_BVH_SKIP", "TJ_MID_ORDER", "TJ_XS_ASYNC"):
        monkeypatch.delenv(k, raising=False)
    a = pkg.Solver(scene, stop=0.0)
    a.iterate(3)
    sa = a.get_state()
    assert a.stats()["error_bits"] == 0
    a.close()
    for env in ({"TJ_PAIR_LPW": "8", "TJ_PAIR_PRIO": "0"}, {"TJ_BVH_SKIP": "0"}, {"TJ_PAIR_LPW": "16", "TJ_BVH_SKIP": "1"}, {"TJ_MID_ORDER": "0"}, {"TJ_XS_ASYNC": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        b = pkg.Solver(scene, stop=0.0)
        b.iterate(3)
        sb = b.get_state()
        assert b.stats()["error_bits"] == 0
        b.close()
        for k in env:
            monkeypatch.delenv(k)
        for n in sa:
            assert np.array_equal(sa[n], sb[n]), (n, env)


@pytest.mark.gpu
def test_coupled_search_in_one_launch_changes_no_bit(pkg, scenes, monkeypatch):
    """coupled mode: the four evaluation rounds of the Armijo search run in ONE launch where a block per (robot, round) has a compute unit of its own
    (the default on this fleet); TJ_LSC_WIDE=0 launches them one after the other as rounds 1 - 3 did.  Same table, same decision, same state."""
    scene = dict(scenes.crossing(12, 4000, seed=23, name="crossing-U12-coupled"), mode=2)
    monkeypatch.delenv("TJ_LSC_WIDE", raising=False)
    a = pkg.Solver(scene, stop=0.0)
    monkeypatch.setenv("TJ_LSC_WIDE", "0")
    b = pkg.Solver(scene, stop=0.0)
    a.iterate(12); b.iterate(12)
    sa, sb = a.get_state(), b.get_state()
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), f"{n} differs between the one-launch and the four-launch coupled search"
    assert a.stats()["error_bits"] == b.stats()["error_bits"]
    assert a.stats()["energy_evals"] == b.stats()["energy_evals"]
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"TJ_CCD_LEAN": "0"}, {"TJ_CCD_LEAN": "1"}, {"TJ_GRAD_FOLD": "0"}, {"TJ_GRAD_NPL": "8"}, {"TJ_SPLIT_UNIONS": "1"},
                                 {"TJ_USE_GRAPH": "1"}, {"TJ_PAIR_ROWS": "4"}, {"TJ_N_SOLVE": "96"}, {"TJ_N_SOLVE": "96", "TJ_HS_MIN": "1"}, {"TJ_SEQ_FOLD": "0"}, {"TJ_LS_FAST": "0"}, {"TJ_GRAD_BALANCE": "1"}, {"TJ_LS_HELP": "1"}, {"TJ_LS_HELP": "2"}, {"TJ_LS_HELP": "3"}, {"TJ_LS_HELP_MUTE": "1"}, {"TJ_HS_BUDGET": "1", "TJ_HS_MIN": "2"}, {"TJ_HS_BUDGET": "40", "TJ_HS_MIN": "1"}, {"TJ_BVH_SKIP": "1"}, {"TJ_BVH_SKIP": "0"}, {"TJ_MID_ORDER": "1"}, {"TJ_XS_ASYNC": "0"}, {"TJ_XS_ONE_QUEUE": "1"},
                                 {"TJ_XS_ASYNC": "0", "TJ_SPLIT_UNIONS": "1"}, {"TJ_XS_ASYNC": "0", "TJ_CCD_LEAN": "0"}],
                         ids=lambda e: "+".join(f"{k}={v}" for k, v in e.items()))
def test_launch_shape_switches_change_no_bit(pkg, scenes, monkeypatch, env):
    """The launch-shape switches of tj_create select other builds / groupings of the same arithmetic: the state
    after several iterations is bitwise the default's.  (TJ_GRAD_NPL=8 forces k_grad's plane batches through several rounds and
    its HBM staging path, TJ_GRAD_FOLD=0 the one-group k_grad behind a separate compaction; TJ_LS_HELP: blocks per robot in the line
    search, 1 = none -- the default on this fleet is 8; TJ_GRAD_BALANCE=1: k_grad's blocks in the order of their last durations, which this
    fleet of 24 x 5 blocks on 256 units does not switch on by itself; TJ_LS_HELP_MUTE=1: helpers that never post, every primary times out and goes on alone.)"""
    scene = scenes.crossing(24, 6000, seed=17, name="crossing-U24-switches")
    for k in env:
        monkeypatch.delenv(k, raising=False)
    a = pkg.Solver(scene, stop=0.0)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    b = pkg.Solver(scene, stop=0.0)
    n_it = 30 if any(k.startswith("TJ_LS_") for k in env) else 6    # the line search: through the back-off iterations of the start and into the steady phase (the default here is 8 blocks per robot)
    a.iterate(n_it); b.iterate(n_it)
    sa, sb = a.get_state(), b.get_state()
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), f"{n} differs with {env}"
    assert a.stats()["error_bits"] == 0 and b.stats()["error_bits"] == 0
    a.close(); b.close()


@pytest.mark.gpu
def test_narrow_first_round_of_plane_heavy_robots_changes_no_bit(pkg, scenes, monkeypatch):
    """config 5 (256 UAVs x 1 M triangles): the two robots next to the obstacle slabs carry hundreds of planes -- too many terms for the team shape -- and, once the fleet is
    in its steady phase (every robot took the full step in the previous iteration: from iteration 13 on here), evaluate a NARROW first round (E(x) and the full step, six
    waves idle at the barriers) instead of eight candidates side by side.  TJ_LS_FAST=0 switches that (and the team shape) off: 18 iterations, same state bit for bit, same
    number of energy evaluations."""
    scene = scenes.scn_d_tri()
    monkeypatch.delenv("TJ_LS_FAST", raising=False)
    a = pkg.Solver(scene, stop=0.0)
    a.iterate_async(18); a.sync()
    sa, ta = a.get_state(), a.stats()
    a.close()
    monkeypatch.setenv("TJ_LS_FAST", "0")
    b = pkg.Solver(scene, stop=0.0)
    b.iterate_async(18); b.sync()
    sb, tb = b.get_state(), b.stats()
    b.close()
    monkeypatch.delenv("TJ_LS_FAST")
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), n
    assert ta["error_bits"] == 0 and tb["error_bits"] == 0 and ta["energy_evals"] == tb["energy_evals"]


@pytest.mark.gpu
def test_counter_collection_keeps_the_one_queue_chain(pkg, scenes, monkeypatch):
    """rocprofv3 --pmc serialises the dispatches of all queues (in an order of its own): a context created under it (the profiler exports ROCPROF_COUNTER_COLLECTION) keeps
    the Newton solve -- and with it the asynchronous front, which rides on the same second queue -- on the chain's queue: no gate launches, unless TJ_XS_ASYNC=1 says
    otherwise; results are the same either way."""
    scene = scenes.crossing(8, 4000, seed=3, name="crossing-U8-counters")
    for k in ("TJ_XS_ASYNC", "ROCPROF_COUNTER_COLLECTION"):
        monkeypatch.delenv(k, raising=False)
    a = pkg.Solver(scene, stop=0.0)
    l0 = a.launch_count(); a.iterate_async(10); a.sync(); la = a.launch_count() - l0
    sa = a.get_state(); a.close()
    monkeypatch.setenv("ROCPROF_COUNTER_COLLECTION", "1")
    b = pkg.Solver(scene, stop=0.0)
    l0 = b.launch_count(); b.iterate_async(10); b.sync(); lb = b.launch_count() - l0
    sb = b.get_state(); b.close()
    monkeypatch.delenv("ROCPROF_COUNTER_COLLECTION")
    assert la == lb + 10 + 9, (la, lb)   # without counters: one gate per iteration for the solve + one per pairing k_linesearch(i) / k_front(i + 1) of the batch for the asynchronous front (round 6)
    for n in sa:
        assert np.array_equal(sa[n], sb[n]), n
