#!/usr/bin/env python3
"""bench.py -- ADMM iterations/sec of the HIP path on the 64-UAV crossing scene (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W           single GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   robots sharded

A "step" is ONE ADMM iteration (one call of Optimization3D_multi::optimization_decouple in the
reference) over the whole fleet.  The timed region is exactly K iterations starting from the initial
trajectory (init_variable), cloud + BVH + state already resident in HBM, no host read-back inside.
Prints ONE JSON line on rank 0 (see README / DESIGN.md for the roofline and cpu_baseline objects).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch


class _DevView:
    """zero-copy torch view of a libtrajadmm device buffer (for RCCL collectives)"""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def kernel_bytes(st, slv, iters):
    """Algorithmic (compulsory) HBM bytes per iteration of every kernel: inputs read once + outputs
    written once, device-counted where data dependent.  Record sizes of this
    implementation: BVH box 48 B, cloud point 24 B, plane 32 B, hull cache 976 B, swept-hull cache 1168 B."""
    U, S, P, T = slv.U, slv.S, slv.P, slv.T
    it = max(1, iters)
    per = {k: st[k] / it for k in ("nodes_dcd", "cand_dcd", "nodes_ccd", "cand_ccd", "planes_obs", "planes_self", "pair_solves", "energy_evals")}
    planes = per["planes_obs"] + per["planes_self"]
    seg_in = U * S * (18 * 8 + 36 * 8)                 # 6 control points x 3 + 6x6 basis per (robot, segment)
    half_pairs = S * U * (U - 1) / 2
    b = {
        "k_begin": 64,
        "k_obs_query": per["nodes_dcd"] * 48 + per["cand_dcd"] * 24 + per["cand_dcd"] * 12 + seg_in + U * S * 18 * 8,
        "k_obs_solve": per["cand_dcd"] * (12 + 4 + 24 + 144) + per["planes_obs"] * 36,
        "k_hullinfo": seg_in + U * S * 976,
        "k_sep_self_rows": U * S * 976 + half_pairs * 48 + per["pair_solves"] * 12,
        "k_sep_self_solve": per["pair_solves"] * (2 * 144 + 2 * 32 + 2 * 4 + 12),
        "k_sep_self_compact": U * S * U * 4 + per["planes_self"] * 64 + U * S * 4 + per["cand_dcd"] * 4 + per["planes_obs"] * 64,
        "k_grad": planes * 32 + seg_in + U * P * (19 + 361) * 8 + U * P * (36 + 2 * 18) * 8,
        "k_xsolve": U * P * (19 + 361) * 8 + U * (3 * T + 4) * 8,
        "k_ccd_prep": U * S * (2 * 18 * 8 + 36 * 8) + U * S * 1168,
        "k_ccd_obs": per["nodes_ccd"] * 48 + per["cand_ccd"] * 24 + U * S * 1168,
        "k_ccd_self_pairs": U * S * 1168 + half_pairs * 48,
        "k_ccd_self_seq": S * U * 4 + U * 16,
        "k_linesearch": U * (S * 36 * 8 + 2 * 3 * T * 8 + P * (36 + 2 * 18 + 2) * 8) + planes * 32 + U * 3 * T * 8,
        "k_slack": U * P * ((18 + 36) * 8 + 2 * 18 * 8 * 2 + 4 * 8),
    }
    # "optimal_plane":1 -- stored planes refined in place: single UAV id + point + plane r/w + list entry and the hull once per
    # segment; multi UAV two hulls + plane r/w + the two published half-offset planes + stamps + list entry per stored pair
    b["k_keep"] = (per["planes_obs"] * (4 + 24 + 3 * 32) + U * S * 144 + per["cand_dcd"] * 8) if slv.mode == 0 else (per["planes_self"] / 2) * (2 * 144 + 2 * 32 + 64 + 8 + 4)
    # union kernels of the single-GPU graph: sums of their constituents
    b["k_front"] = b["k_obs_query"] + b["k_sep_self_rows"]
    b["k_mid"] = b["k_slack"] + b["k_sep_self_solve"] + b["k_obs_solve"]
    b["k_ccd"] = b["k_ccd_obs"] + b["k_ccd_self_pairs"]
    n = 9 * P - 2
    b["k_xsolve_c2"] = U * (n * n + 2 * n + 4) * 8 + U * (3 * T + 4) * 8                  # coupled mode: factor + rhs in, direction out
    b["k_ls_coupled"] = b["k_linesearch"] + U * 8 * 8                                       # per evaluation round (first round; later rounds exit early)
    b["k_ls_commit"] = U * (2 * 3 * T * 8 + 3 * T * 8) + 4 * U * 8 * 8
    if slv.mode != 2:   # single-GPU graph: the hull cache is written by k_linesearch (Dev::fuse)
        b["k_linesearch"] += U * S * 976 if slv.mode == 1 else 0
    if st is not None and slv.mode == 2:
        b["k_xsolve"] = U * P * (19 + 361) * 8 + U * (n * n + 2 * n + 4) * 8               # writes its factor for k_xsolve_c2
    return b


def survey_bytes(st, slv, iters):
    """SURVEY.md section 8(d)'s algorithmic bytes, w = 8 (fp64), attributed to the kernel that performs each term:
      B_state  = 2 U (3T + 2*18P + 2P + 1) w      (state read + written once per iteration)
      B_bvh    = (V_dcd + V_ccd) * box bytes       (boxes visited; this implementation's box is 24 B: outward-rounded fp32)
      B_cand   = (C_dcd + C_ccd) * 3w              (candidate primitives fetched; 3 vertices each for triangle obstacles)
      B_planes = N_pl * 4w * (1 + 1 + E)           (written once, read by the gradient, read by each of E energy evaluations)
      B_hess   = U P (19^2 + 19) w * 2 + U (9P-2)^2 w * 2
      B_pair   = 2 * S U 18 w                      (all hulls read once per pair pass: DCD, CCD)
    The hull / swept-hull caches this implementation keeps in HBM between kernels are NOT counted (implementation traffic).
    Returns ({kernel: bytes per launch}, whole-iteration bytes)."""
    U, S, P, T, w = slv.U, slv.S, slv.P, slv.T, 8
    it = max(1, iters)
    per = {k: st[k] / it for k in ("nodes_dcd", "cand_dcd", "nodes_ccd", "cand_ccd", "planes_obs", "planes_self", "energy_evals")}
    npl = per["planes_obs"] + per["planes_self"]
    E = per["energy_evals"] / U
    box, prim = slv.box_bytes, 3 * w * slv.prim_vertices
    x_state = U * (3 * T + 1) * w                    # control net + piece_time
    z_state = U * (2 * 18 * P + 2 * P) * w           # slack + dual blocks
    n = 9 * P - 2
    multi = slv.mode >= 1
    b = {
        "k_begin": 0,
        "k_front": per["nodes_dcd"] * box + per["cand_dcd"] * prim + x_state + (S * U * 18 * w if multi else 0),
        "k_mid": npl * 4 * w + 2 * z_state + x_state,                   # planes written; slack/dual read + written
        "k_sep_self_compact": 0,
        "k_keep": npl * 4 * w * 2,
        "k_grad": npl * 4 * w + U * P * (361 + 19) * w + x_state + z_state,
        "k_xsolve": U * P * (361 + 19) * w + 2 * U * n * n * w + U * 3 * T * w,
        "k_xsolve_c2": 2 * U * n * n * w,
        "k_ccd_prep": 0,
        "k_ccd": per["nodes_ccd"] * box + per["cand_ccd"] * prim + 2 * x_state + (S * U * 18 * w if multi else 0),
        "k_ccd_self_seq": 0,
        "k_linesearch": npl * 4 * w * E + 2 * x_state + z_state,
        "k_ls_coupled": npl * 4 * w * E / 4 + x_state + z_state,
        "k_ls_commit": 2 * x_state,
        "k_hullinfo": 0, "k_slack": 2 * z_state + x_state,
    }
    b["k_obs_query"] = b["k_front"]; b["k_sep_self_rows"] = 0; b["k_obs_solve"] = per["planes_obs"] * 4 * w; b["k_sep_self_solve"] = per["planes_self"] * 4 * w
    b["k_ccd_obs"] = b["k_ccd"]; b["k_ccd_self_pairs"] = 0
    total = (2 * (x_state + z_state) + (per["nodes_dcd"] + per["nodes_ccd"]) * box + (per["cand_dcd"] + per["cand_ccd"]) * prim + npl * 4 * w * (2 + E)
             + 2 * U * P * (361 + 19) * w + 2 * U * n * n * w + (2 * S * U * 18 * w if multi else 0))
    return b, total


# What ONE wavefront pays on MI355X, in ns; every entry cites the line of the committed probe output it was read from
# (tools/micro/issue_probe.hip -> profiles/round4_issue_probe.txt, tools/micro/mem_probe.hip -> profiles/round4_mem_probe.txt; 2.39 GHz):
PRIM = {"fp64": 2.6,        # issue_probe:2-3   one fp64 VALU instruction, dependent (2.52) OR one of four independent chains (2.64): a wave issues one every ~6.3 cycles
        "readlane": 16.1,   # issue_probe:5     v_readlane pair -> first VALU use of the SGPR it wrote
        "rsq": 10.6,        # issue_probe:6     v_rsq_f64 + fma
        "sqrt": 46.7,       # issue_probe:7     IEEE sqrt (mem_probe:9 measures 67 with the loop's add and counter)
        "div": 32.8,        # issue_probe:8     IEEE division (round 3's row was folded away by the compiler: 0.94 ns; mem_probe:8: 45 with the loop's add and counter)
        "log": 158.4,       # issue_probe:11    ocml log
        "lds": 50.6,        # issue_probe:10    LDS write -> read round trip
        "mem": 240.0,       # mem_probe:4       dependent global load of a line the PREVIOUS kernel wrote on another XCD (= an Infinity-Cache hit, mem_probe:2: 238;
                            #                   own-L2 hit 101, mem_probe:1; miss to HBM 390, mem_probe:3).  Rounds 1-3 priced this at 700 ns without a measurement.
        "branch": 38.0}     # mem_probe:6-7     scalar branch on a fresh VALU comparison (v_cmp, v_readfirstlane, s_cbranch): 64.6 ns per iteration against 26.8 for the
                            #                   same recurrence with a select (which also pays a second fma): 64.6 - 26.8 = 37.8; round 3 priced it at 10


def critical_path(slv, st, K, per_launch_ms, n_prims, batch_us=None):
    """roofline.critical_path: per chain kernel the DEPENDENCY CHAIN of its longest work item -- the operations that must follow
    one another whatever the parallel width -- priced with the measured single-wave latencies above.  It is the bound this
    latency-dominated path can be judged against (the HBM roofline is kept beside it as SURVEY 8(d) demands): `frac` =
    bound / measured launch time.  Unit counts come from the device (longest robot-pair GJK, Armijo evaluations) and from the
    problem's shape; the formulas are written out in DESIGN.md section 5."""
    p = PRIM
    U, S, P, res = slv.U, slv.S, slv.P, slv.res
    it = max(1, K)
    n = 9 * P - 2
    lv, cnt = 1, (n_prims + 7) // 8
    while cnt > 64:
        cnt = (cnt + 7) // 8; lv += 1
    gjk_max = max(1.0, st["gjk_max_sum"] / it) if slv.mode >= 1 else 1.0
    # Armijo rounds per robot.  One block per robot evaluates 8 candidates per round; with helper blocks on idle compute units (k_linesearch's
    # super-rounds: min(8, CUs // robots) blocks per robot, two candidates each) a round decides 2H candidates and every round after the
    # first adds the primary's permit and the helpers' posts: two dependent memory round trips.
    try:
        import torch as _t
        cus = _t.cuda.get_device_properties(_t.cuda.current_device()).multi_processor_count
    except Exception:
        cus = 256
    helpers = max(1, min(8, cus // max(1, U))) if slv.mode in (0, 1) and os.environ.get("TJ_LS_HELP", "") != "1" else 1
    if os.environ.get("TJ_LS_HELP", "").isdigit() and int(os.environ["TJ_LS_HELP"]) >= 1:
        helpers = min(8, int(os.environ["TJ_LS_HELP"]))
    per_round = 2 * helpers if helpers > 1 else 8
    rounds = 1.0 + max(0.0, (st["energy_evals"] / it / U - 2.0) / per_round)
    walk = (lv + 3) * p["mem"]                                                    # record, one box load per level, leaf primitives, work-item slot
    gjk_iter = (12 * p["fp64"] + p["readlane"]) + (21 * p["fp64"] + p["sqrt"] + p["div"] + p["branch"] + p["readlane"])   # support search + triangle step
    pair = 2 * p["mem"] + gjk_max * gjk_iter + (p["sqrt"] + 3 * p["div"] + 12 * p["fp64"]) + (p["log"] + 2 * p["div"] + 24 * p["fp64"])   # work item -> hulls, GJK, normal + offsets, one Newton round
    m_sum = sum(range(2, 19))                                                     # Householder steps of the 19 x 19 block: trailing sizes 18 .. 2
    eig = 17 * (p["rsq"] + 2 * p["readlane"] + 8 * p["fp64"] + 5 * (3 * p["fp64"]) + p["readlane"]) + 2 * m_sum * p["fp64"] + 9 * (18 * 3 * p["fp64"] + 30.0)
    grad = 3 * p["mem"] + (p["log"] + p["sqrt"] + 6 * p["div"] + 30 * p["fp64"]) + 2 * p["lds"] + eig   # counts -> planes -> (vel/acc record | plane terms) -> consensus -> repair
    xsolve = 2 * p["mem"] + p["lds"] + n * (p["rsq"] + 8 * p["fp64"] + p["readlane"] + 2 * p["fp64"]) + n * (p["readlane"] + 2 * p["fp64"]) + 3 * p["lds"] + p["mem"]
    evalx = p["lds"] + 7 * (p["lds"] + p["sqrt"] + p["div"] + 12 * p["fp64"]) + 2 * p["log"] + 3 * (p["lds"] + 6 * p["fp64"]) + p["log"] + 6 * p["lds"] + 2 * p["lds"] + 36 * p["fp64"]
    ls = 2 * p["mem"] + p["lds"] + rounds * evalx + (max(0.0, rounds - 1.0) * 2 * p["mem"] if helpers > 1 else 0.0) + (p["lds"] + 6 * p["fp64"]) + (p["mem"] + 6 * 5 * p["fp64"]) + 2 * p["mem"]   # stage, rounds (+ signalling), exact hulls, intervals, store + ticket
    folded = slv.mode >= 1 and per_launch_ms.get("k_ccd_self_seq", 0.0) <= 0   # the sequential pair replay + gnorm is the tail of k_ccd: + the counter's landing and one poll
    bound = {"k_front": walk + p["mem"], "k_mid": pair, "k_grad": grad, "k_xsolve": xsolve, "k_ccd": walk + p["mem"] + (2 * p["mem"] if folded else 0.0),
             "k_ccd_self_seq": 3 * p["mem"] + 200 * p["fp64"], "k_linesearch": ls}
    out, tb, tm = {}, 0.0, 0.0
    for k, b in bound.items():
        ms = per_launch_ms.get(k, 0.0)
        if ms <= 0:
            continue
        out["tj::" + k] = {"bound_us": b * 1e-3, "measured_us": ms * 1e3, "frac": b * 1e-3 / (ms * 1e3)}
        tb += b * 1e-3; tm += ms * 1e3
    # frac: against the BATCH CLOCK (the iteration as the timed window measured it) -- the sum of the per-kernel hipEvent times carries ~1 - 2 us of event
    # overhead per kernel and exceeds it; frac_vs_event_sum keeps the old figure
    den = batch_us if batch_us else tm
    return {"unit": "us", "kernels": out, "chain_bound_us": tb, "chain_measured_us": tm, "iteration_batch_clock_us": batch_us, "frac": tb / den if den else None, "frac_vs_event_sum": tb / tm if tm else None,
            "unit_counts": {"bvh_levels": lv, "longest_pair_gjk_iterations": gjk_max, "armijo_rounds": rounds, "armijo_candidates_per_round": per_round, "linesearch_blocks_per_robot": helpers, "newton_system_rows": n},
            "primitives_ns": PRIM,
            "note": "dependency chain of each kernel's longest work item x measured single-wave latencies (profiles/round4_issue_probe.txt, profiles/round4_mem_probe.txt; a dependent global round trip is 240 ns measured, not the 700 ns rounds 1-3 assumed: frac fell from 0.28 to what is printed here); the rest of a launch is "
                    "instruction issue of ONE wave (one fp64 instruction per ~6.3 cycles, dependent or not) plus dispatch / drain: see DESIGN.md section 5"}


def source_id():
    """sha256 over the product sources: keys profile files to the build they were measured on"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "traj-opt-admm_amd", "csrc")
    for f in sorted(os.listdir(src)) + ["../../include/trajadmm.h"]:
        if f.endswith((".h", ".hip", ".cpp", "Makefile")):
            h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(scene, steps, optimal_plane=False):
    """Reference CPU path on this box's host cores: the unmodified reference (oracle/_ref/libref.so,
    prebuilt in the dev container) if present, else this repo's CPU restatement.  Single thread --
    the reference has no threading (no `#pragma omp` anywhere in its first-party code)."""
    from oracle import pyoracle
    # triangle obstacles exist in the port only: the reference's triangle path is dead code
    kind = "reference" if pyoracle.available("ref") and scene.get("tris") is None else "port"
    eng = pyoracle.Engine("ref" if kind == "reference" else "port", scene)
    if optimal_plane:
        eng.set_optimal_plane(True)
    n = min(steps, 20)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.iterate()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "iters/s", "ms_per_iter": 1e3 * dt / n, "cores": 1, "kind": kind,
            "sample": f"the first {n} ADMM iterations of the same scene from the same initial trajectory (BVH build excluded)"}


def roofline_fracs(slv, prof, st, K, dt):
    """(dominant kernel, its SURVEY 8(d) fraction of the HBM peak, whole-iteration fraction) from a profile_kernels() result"""
    alg_bytes, alg_total = survey_bytes(st, slv, K)
    per_launch_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
    dom = max((k for k, v in prof.items() if v[1] >= K), key=lambda k: prof[k][0])
    ach = alg_bytes[dom] / (per_launch_ms[dom] * 1e-3) / 1e9
    return dom, ach / 8000.0, alg_total / (dt / K) / 1e9 / 8000.0


def extra_config(pkg, scene, K, W, device):
    """One more BASELINE config on the already warm GPU: same protocol as the headline (K iterations from the initial
    trajectory, state resident, stop test off), no CPU leg.  Returns the entry of `extra.configs`."""
    slv = pkg.Solver(scene, device=device, stop=0.0)
    slv.iterate(60); slv.sync(); slv.reset()          # the library looks at its counters here and picks kernel builds
    slv.iterate_async(max(W, 1)); slv.sync(); slv.reset(); slv.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slv.iterate_async(K); slv.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = slv.stats()
    if st["error_bits"] or st["order_unresolved"]:
        raise SystemExit(f"{scene['name']}: device error bits {st['error_bits']}, unresolved pair order {st['order_unresolved']}")
    slv.reset()
    prof = slv.profile_kernels(K)
    dom, frac, whole = roofline_fracs(slv, prof, slv.stats(), K, dt)
    n_prim = scene['tris'].shape[0] if scene.get('tris') is not None else scene['cloud'].shape[0]
    mode_name = {0: "single-UAV path (admmPathPlanning3D)", 1: "decoupled", 2: "coupled"}[scene["mode"]]
    out = {"workload": f"{scene['name']}: {scene['U']} UAV{'s' if scene['U'] > 1 else ''}, {n_prim} obstacle {'triangles' if scene.get('tris') is not None else 'points'}, {scene['P']} pieces x res 8, {mode_name}",
           "ms_per_step": 1e3 * dt / K, "iters_per_s": K / dt, "steps": K,
           "roofline": {"kernel": "tj::" + dom, "frac": frac}, "whole_iteration": {"frac": whole}}
    slv.close()
    return out


STATE_KEYS = ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda", "piece_time")


def one_rank_state(pkg, scene, device, n_it, optimal_plane):
    """state after n_it iterations of the DEFAULT single-context path (the fused chain the N = 1 line times) on one device: what every
    sharded run has to reproduce bit for bit"""
    slv = pkg.Solver(scene, device=device, stop=0.0, optimal_plane=int(optimal_plane))
    slv.iterate(n_it)
    st = slv.get_state()
    bits = slv.stats()["error_bits"]
    slv.close()
    if bits:
        raise SystemExit(f"one-rank reference run: device error bits {bits}")
    return st


def states_equal(a, b, u0=0, u1=None):
    return all(np.array_equal(a[k][u0:u1], b[k][u0:u1]) for k in STATE_KEYS)


def bench_group(pkg, scene, devices, args):
    """K iterations through tj_group (one process, one host thread per rank inside the library).  Same top-level keys as the
    single-GPU line; `roofline` / `cpu_baseline` are null here (they are N = 1 objects, see the default run).
    SELF-VALIDATING: before anything is timed, every transport (flag -> event -> rccl, or the one TJ_GROUP_TRANSPORT names) runs V
    iterations and its state is compared BITWISE with V iterations of one rank; the first transport that reproduces it is the one
    timed, and `group.transports` says what happened to each (error string included) -- a wrong exchange cannot yield a number."""
    torch.cuda.set_device(devices[0])
    K, W, V = args.steps, args.warmup, args.validate_iters
    multi = len(devices) > 1
    ref = one_rank_state(pkg, scene, devices[0], V, args.optimal_plane)
    order = [os.environ["TJ_GROUP_TRANSPORT"]] if os.environ.get("TJ_GROUP_TRANSPORT") else (["flag", "event", "rccl"] if multi else ["event"])
    tried, chosen = {}, None
    for t in order:
        grp = None
        try:
            grp = pkg.Group(scene, devices, stop=0.0, optimal_plane=int(args.optimal_plane))
            if multi:
                grp.set_transport(t)
            grp.iterate(1)                 # a transport that cannot deliver fails here, after two exchanges
            grp.iterate(V - 1)
            ok = states_equal(grp.get_state(), ref)
            tried[t] = {"ran": True, "bitwise_equal_to_one_rank": bool(ok), "rccl_ranks": (grp.rccl_ranks if t == "rccl" else None)}
            if ok and chosen is None:
                chosen = t
        except pkg.TrajAdmmError as e:
            tried[t] = {"ran": False, "bitwise_equal_to_one_rank": False, "error": str(e)}
        finally:
            if grp is not None:
                try:
                    grp.close()
                except Exception:
                    pass
    validation = {"iterations": V, "reference": "one rank, default fused chain, same device 0", "transports": tried, "timed_transport": chosen,
                  "bitwise_equal_to_one_rank": chosen is not None}
    if chosen is None:
        print(json.dumps({"metric": "ADMM iterations/sec", "value": None, "unit": "iters/s", "n_gpus": len(set(devices)), "steps": K, "warmup": W,
                          "error": "no transport reproduced the one-rank state bit for bit; nothing was timed", "group": {"validation": validation}}), flush=True)
        raise SystemExit(3)

    def timed(devs, transport):
        grp = pkg.Group(scene, devs, stop=0.0, optimal_plane=int(args.optimal_plane))
        if len(devs) > 1:
            grp.set_transport(transport)
        grp.iterate(300); grp.reset()          # clock ramp, like the default path
        grp.iterate(max(W, 1)); grp.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        grp.iterate(K)                         # returns after every rank's stream has drained
        return grp, time.perf_counter() - t0

    # every transport that validated is timed; the line's value is the fastest of them (all of them reproduce the one-rank state bit for bit)
    timings = {}
    grp, dt = None, None
    for t, v in tried.items():
        if not v.get("bitwise_equal_to_one_rank"):
            continue
        g_t, dt_t = timed(devices, t)
        timings[t] = 1e3 * dt_t / K
        if dt is None or dt_t < dt:
            if grp is not None:
                grp.close()
            grp, dt, chosen = g_t, dt_t, t
        else:
            g_t.close()
    validation["timed_transport"] = chosen
    validation["ms_per_step_by_transport"] = timings
    transport = grp.transport
    rccl_ranks = grp.rccl_ranks
    check = None
    if args.state_checksum:
        import hashlib
        stt = grp.get_state()
        check = hashlib.sha256(np.ascontiguousarray(stt["spline"]).tobytes() + np.ascontiguousarray(stt["piece_time"]).tobytes()).hexdigest()
    ex_us = [float(x) for x in grp.profile_exchange(50)] if multi else [0.0] * 5
    grp.close()
    # the same schedule on ONE rank through the same entry point: what the phase schedule itself costs without any exchange
    # (the default single-GPU line runs the fused chain instead and is faster; SCALE's N = 1 is that default line)
    g1, dt1 = timed([devices[0]], "event")
    g1.close()
    n_ex = 5 if scene["mode"] == 2 else 2
    distinct = len(set(devices)) == len(devices)
    out = {"metric": "ADMM iterations/sec", "value": K / dt, "unit": "iters/s", "n_gpus": len(set(devices)), "steps": K, "warmup": W,
           "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{scene['name']}: {scene['U']} UAVs, {scene['P']} pieces x res 8, {'coupled' if scene['mode'] == 2 else 'decoupled'} mode",
                      "parallelism": f"tj_group: {len(devices)} ranks on devices {devices}, robots block-sharded, {n_ex} exchanges/iter inside the library, transport {transport}",
                      "iters_timed_from": "initial trajectory"},
           "timed_window_ms": 1e3 * dt,
           "group": {"launcher": "tj_group (one process, one host thread per rank)", "transport": transport, "ranks": len(devices), "devices": list(devices), "distinct_devices": distinct,
                     "rccl_ranks": rccl_ranks, "bitwise_equal_to_one_rank": True, "validation": validation,
                     "exchanges_per_iter": n_ex, "exchange_us": dict(zip(["control_points", "directions", "schur_corner", "ccd_exponents", "armijo_energies"], ex_us)),
                     "one_rank_same_schedule_ms_per_step": 1e3 * dt1 / K,
                     "expectation": EXPECT.get(scene["name"].replace("-coupled", ""), "")},
           "roofline": None, "cpu_baseline": None,
           "note": "roofline and cpu_baseline belong to the single-GPU line (python bench.py); strong scaling: the fleet is fixed, each rank owns U / ranks robots"}
    if check:
        print("CHECK group " + check, flush=True)
    print(json.dumps(out), flush=True)


# what sharding can and cannot buy per scene, carried in the multi-GPU lines so that a scaling curve is read correctly
EXPECT = {
    "SCN-C": "64 UAVs: every stage is as long as its slowest single item (one wave), which does not shrink when a rank owns fewer robots; each exchange adds its latency -> N > 1 is expected to be SLOWER than N = 1 here",
    "SCN-B": "8 UAVs: as SCN-C, nothing to gain from sharding",
    "SCN-D": "256 UAVs: k_mid (10 000 robot-pair solves), k_grad (1 280 pieces) and k_front are throughput-bound on one GPU and partition by ownership -> these kernels shrink with N, the latency floor of the chain and the exchanges do not",
    "SCN-D-tri": "256 UAVs x 1M triangles: as SCN-D",
    "SCN-E": "64 UAVs through 1M points: obstacle stages shrink with N, the rest is latency",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="C", choices=["A", "B", "C", "D", "Dtri", "E", "H8"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra.configs legs (BASELINE configs 2, 3 and 5 timed after the headline: SCN-A, SCN-B, SCN-D-tri)")
    ap.add_argument("--coupled", action="store_true", help='time the coupled mode ("decouple":0, one shared piece_time) instead of the shipped decoupled mode; single GPU only')
    ap.add_argument("--optimal-plane", action="store_true", help='time the "optimal_plane":1 variant (persistent planes refined every iteration); not the headline')
    ap.add_argument("--overlap-gather", action="store_true", help="sharded schedule: start the control-point all-gather before phase 0 and join it after (async RCCL op); "
                                                                  "off by default: with one rank it measured 13 us slower per iteration, its effect with real peers is unmeasured")
    ap.add_argument("--no-direct", action="store_true", help="sharded runs under torchrun: do not try the direct (hipIpc, in-kernel) exchange, time the RCCL all-gathers")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded schedule + RCCL collectives even with one rank (self test)")
    ap.add_argument("--same-gpu", action="store_true", help="TEST ONLY: every rank uses device 0 and the all-gathers are staged through host memory over gloo "
                                                            "(RCCL refuses two ranks on one device); exercises the multi-process schedule on a 1-GPU box")
    ap.add_argument("--group-devices", default=None, help="ONE process drives several ranks through the library's own sharding (tj_group: peer stores + events, no torch "
                                                          "collectives), e.g. 0,1,2,3 -- entries may repeat (0,0 = two ranks on one GPU).  `--gpus N` without torchrun selects devices 0..N-1")
    ap.add_argument("--validate-iters", type=int, default=8, help="multi-GPU runs: iterations of the untimed self-validation (sharded state == one-rank state, bitwise) that precedes the timing")
    ap.add_argument("--state-checksum", action="store_true", help="each rank also prints 'CHECK <rank> <sha256 of its owned robots\' final control points and piece times>'")
    args = ap.parse_args()

    pkg = importlib.import_module("traj-opt-admm_amd")
    sc = pkg.scenes
    scene = {"A": sc.scn_a, "B": sc.scn_b, "C": sc.scn_c, "D": sc.scn_d, "Dtri": sc.scn_d_tri, "E": sc.scn_e, "H8": lambda: sc.hard(8, 20000)}[args.scene]()

    if args.coupled:
        scene = dict(scene); scene["mode"] = 2; scene["name"] += "-coupled"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_gpu and world > 1:
        # Processes SHARING a GPU (test arrangement): the device runs one process's waves at a time, so a kernel that sleeps on a word which another queue of its process
        # is to write (the asynchronous Newton solve of the one-rank reference contexts below) shuts the other process out -- and with both processes doing it, each other.
        os.environ["TJ_XS_ASYNC"] = "0"
        os.environ["TJ_KEEP_ASYNC"] = "0"
    group_devices = [int(x) for x in args.group_devices.split(",")] if args.group_devices else None
    if world == 1 and args.gpus > 1 and group_devices is None:
        group_devices = list(range(args.gpus))   # not under torchrun: the library shards by itself
    if group_devices is not None:
        return bench_group(pkg, scene, group_devices, args)
    dist = None
    sharded = world > 1 or args.force_dist
    torch.cuda.set_device(local)   # torch initialises the HIP runtime first; the library then shares it
    if sharded:
        import torch.distributed as dist
        if "TJ_KEEP_NCCL_DEBUG" not in os.environ:
            os.environ["NCCL_DEBUG"] = "NONE"   # RCCL prints its version banner and WARN lines on stdout, next to the JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.same_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if scene["U"] % world != 0:
        raise SystemExit("robot count must divide evenly over the ranks")
    slv = pkg.Solver(scene, device=local, rank=rank, world=world, stop=0.0, optimal_plane=int(args.optimal_plane))  # stop test off: time exactly K iterations
    K, W = args.steps, args.warmup

    if sharded:
        # kernels and collectives are ordered on one dedicated torch stream
        tstream = torch.cuda.Stream(device=local)
        torch.cuda.set_stream(tstream)
        slv.set_stream(tstream.cuda_stream)
        views = []
        for what in ((0, 1, 2, 3, 4) if args.coupled else (0, 1)):
            ptr, per, first, n = slv.exchange_buffer(what)
            full = torch.as_tensor(_DevView(ptr, per * slv.U), device=f"cuda:{local}")
            views.append((full, full[first * per:(first + n) * per]))

        sharding = importlib.import_module("traj-opt-admm_amd.sharding")

        class _Eng:
            chained = True   # sharding.run_sharded says whether another iteration follows: the next begin rides in phase 2's line search

            @staticmethod
            def phase(k, more=0):
                slv.iterate_phase(k, more)

        def _gather(what):  # RCCL all-gather straight on the library's device buffers (in place)
            if args.same_gpu:  # test path: device slice -> host -> gloo all-gather -> device
                tstream.synchronize()
                mine = views[what][1].cpu()
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                views[what][0].copy_(torch.cat(parts).to(views[what][0].device))
                tstream.synchronize()
            else:
                dist.all_gather_into_tensor(views[what][0], views[what][1])

        def _gather_begin(what):  # start the RCCL all-gather on torch's collective stream; the returned call joins it
            if args.same_gpu or not args.overlap_gather:
                return lambda: _gather(what)
            work = dist.all_gather_into_tensor(views[what][0], views[what][1], async_op=True)
            return work.wait

        # DIRECT exchange between the processes (include/trajadmm.h tj_xch_*): every rank's receive block is mapped into every other process through
        # hipIpc, the producing kernels push, the consuming kernels wait -- the fused six-kernel chain per rank, no collective on the path.  Tried first
        # (decoupled mode, more than one rank); it has to reproduce the one-rank state bit for bit like any other path, else the RCCL all-gathers are timed.
        path = {"name": "rccl", "direct_error": None}
        if world > 1 and not args.coupled and not args.no_direct:
            ok_local, err = 1, None
            try:
                mine_h = slv.xch_ipc_export()
            except Exception as e:   # noqa: BLE001 -- any failure sends every rank to the collective path
                mine_h, ok_local, err = None, 0, str(e)
            hs = [None] * world
            dist.all_gather_object(hs, (mine_h, err))
            if all(h[0] is not None for h in hs):
                try:
                    slv.xch_attach_ipc({r: hs[r][0] for r in range(world) if r != rank}, poll_in_kernel=not args.same_gpu)
                except Exception as e:   # noqa: BLE001
                    ok_local, err = 0, str(e)
            else:
                ok_local, err = 0, "; ".join(f"rank {r}: {h[1]}" for r, h in enumerate(hs) if h[1])
            oks = [None] * world
            dist.all_gather_object(oks, (ok_local, err))
            if all(o[0] for o in oks):
                path["name"] = "direct"
            else:
                path["direct_error"] = "; ".join(f"rank {r}: {o[1]}" for r, o in enumerate(oks) if o[1])
                try:
                    slv.xch_enable(False)
                except Exception:   # noqa: BLE001
                    pass

        def run(n_it):
            if path["name"] == "direct":
                slv.iterate_async(n_it)
            elif args.coupled:   # six phases, five small all-gathers per iteration (sharding.COUPLED_SCHEDULE)
                sharding.run_schedule(_Eng, _gather, n_it, sharding.COUPLED_SCHEDULE)
            else:
                sharding.run_sharded(_Eng, _gather, n_it, gather_begin=_gather_begin)
    else:
        def run(n_it):
            slv.iterate_async(n_it)

    def barrier():
        if dist is not None:
            dist.barrier()
        slv.sync()
        torch.cuda.synchronize()

    # SELF-VALIDATION of the sharded path (world > 1, or --force-dist): V iterations through the very schedule that is timed below, then
    # V iterations of ONE rank (the default fused chain, a second context on this rank's own GPU); the robots this rank owns must
    # agree bit for bit.  Every rank checks its slice, the verdict is the minimum over the ranks.
    validation = None
    if sharded:
        V = args.validate_iters
        u0, u1 = rank * scene["U"] // world, (rank + 1) * scene["U"] // world
        ref = one_rank_state(pkg, scene, local, V, args.optimal_plane)
        tried = {}
        while True:
            mine_ok, err = False, None
            try:
                run(V)
                barrier()
                st_sh = slv.get_state()
                bits = slv.stats()["error_bits"]
                mine_ok = bits == 0 and states_equal(st_sh, ref, u0, u1)
                if bits:
                    err = f"device error bits {bits}"
            except pkg.TrajAdmmError as e:   # e.g. a peer's push that never arrived (direct exchange): recorded, the next path is tried
                err = str(e)
            flag = torch.tensor([1.0 if mine_ok else 0.0], dtype=torch.float64, device=f"cuda:{local}")
            nok = flag.clone()
            if not args.same_gpu:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN); dist.all_reduce(nok, op=dist.ReduceOp.SUM)
            else:   # gloo stages through the host
                fh, nh = flag.cpu(), nok.cpu()
                dist.all_reduce(fh, op=dist.ReduceOp.MIN); dist.all_reduce(nh, op=dist.ReduceOp.SUM)
                flag, nok = fh, nh
            tried[path["name"]] = {"bitwise_equal_to_one_rank": bool(flag.item() == 1.0), "ranks_equal": int(nok.item()), "error_rank0": err}
            if flag.item() == 1.0 or path["name"] != "direct":
                break
            # the direct exchange did not reproduce the one-rank state (or failed): every rank falls back to the collectives
            path["name"] = "rccl"
            try:
                slv.xch_enable(False)
            except pkg.TrajAdmmError:
                pass
            slv.reset()
            barrier()
        validation = {"iterations": V, "reference": "one rank, default fused chain, a second context on each rank's own GPU", "checked": "every state array of the robots each rank owns",
                      "bitwise_equal_to_one_rank": bool(flag.item() == 1.0), "ranks_equal": int(nok.item()), "ranks": world, "paths": tried, "timed_path": path["name"],
                      "direct_exchange_setup_error": path["direct_error"]}
        slv.reset()
        barrier()

    # clock ramp: a freshly started process finds the GPU at its idle clock (543 MHz sclk on the bench box); ~60 ms of the same
    # work, untimed and before the W warm-up steps, lets the power management settle so that K short steps are not timed on the ramp
    if sharded:
        run(300)
    else:
        slv.iterate(300)   # the synchronous call: it is also where the library looks at its counters and picks kernel builds for what follows
    barrier()
    slv.reset()
    # warmup (also instantiates the hipGraph), then restart from the initial trajectory
    run(max(W, 1))
    barrier()
    slv.reset()
    barrier()
    t0 = time.perf_counter()
    run(K)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local}")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    st = slv.stats()
    if st["error_bits"]:
        raise SystemExit(f"device error bits {st['error_bits']}")
    if st["order_unresolved"]:
        raise SystemExit(f"inter-robot CCD clamp: {st['order_unresolved']} segments whose pair order could not be replayed in the reference's tree order")

    if args.state_checksum:
        import hashlib
        stt = slv.get_state()
        u0, u1 = rank * scene["U"] // world, (rank + 1) * scene["U"] // world
        h = hashlib.sha256(np.ascontiguousarray(stt["spline"][u0:u1]).tobytes() + np.ascontiguousarray(stt["piece_time"][u0:u1]).tobytes()).hexdigest()
        print(f"CHECK {rank} {h}", flush=True)
    out = None
    if rank == 0:
        out = {"metric": "ADMM iterations/sec", "value": K / dt, "unit": "iters/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{scene['name']}: {scene['U']} UAVs crossing, {scene['tris'].shape[0] if scene.get('tris') is not None else scene['cloud'].shape[0]} obstacle {'triangles (fp64 narrow phase, fp32 outward-rounded BVH boxes)' if scene.get('tris') is not None else 'points'}, "
                                      f"{scene['P']} pieces x res 8 = {slv.S} segments/robot, {'coupled mode (decouple:0)' if args.coupled else 'decoupled mode (3D.json defaults)'}{', optimal_plane:1' if args.optimal_plane else ''}",
                          "parallelism": f"robots sharded over {world} GPU(s), one process per GPU, {'direct exchange (in-kernel pushes / waits through hipIpc-mapped receive blocks), six kernels per iteration and rank' if sharded and path['name'] == 'direct' else str(5 if args.coupled else 2) + ' RCCL all-gathers/iter on the library exchange buffers'}; expectation: {EXPECT.get(scene['name'].split('-coupled')[0], '')}" if world > 1 else (f"1 GPU, whole iteration resident on the device: a linear chain of 6 kernels on one queue (union kernels), enqueued ahead, no host sync" if args.coupled or os.environ.get("TJ_XS_ASYNC") == "0"
                                           else "1 GPU, whole iteration resident on the device: six kernels per iteration enqueued ahead, no host sync -- four in a chain on one queue; on a second queue the Newton solve next to the gradient kernel and the next iteration's plane queries next to the line search (in-kernel tickets / flags / commit flags; TJ_XS_ASYNC=0 TJ_FRONT_ASYNC=0: all six on one queue)"),
                          "iters_timed_from": "initial trajectory"}}
        if scene["name"] == "SCN-C" and not (args.coupled or args.optimal_plane):
            out["config"]["parity_pin"] = ("the timed SCN-C is pinned against the unmodified reference PER ITERATION (tests/golden/stages_scn_c.npz) and end to end only inside the reference's own "
                                           "1-ulp envelope (1.2e-2 on this scene: robots 0.25 apart cross inside each other's barrier range); north_star's end-to-end rel 1e-8 at 64 UAVs / 100 000 points "
                                           "is tested on SCN-C3 -- the same fleet and cloud stacked 0.29 apart, a spacing SEARCHED for so that the reference reproduces itself (envelope 8e-11): "
                                           "control points 1.3e-10, final energies 8.9e-8 (bar max(1e-8, 3 x the reference's own 5.8e-8 energy envelope)), tests/golden/e2e_scn_c3.npz")
        if sharded:
            tname = ("direct exchange: in-kernel pushes into hipIpc-mapped receive blocks of the peers, in-kernel waits (six kernels per iteration and rank, no collective)" if path["name"] == "direct"
                     else ("gloo through host memory (TEST ONLY)" if args.same_gpu else "RCCL all_gather_into_tensor on the library's exchange buffers (zero copy)"))
            out["group"] = {"launcher": "torch.distributed.run, one process per GPU", "backend": dist.get_backend(), "transport": tname,
                            "ranks": world, "rccl_ranks": (dist.get_world_size() if dist.get_backend() == "nccl" else 0),
                            "bitwise_equal_to_one_rank": validation["bitwise_equal_to_one_rank"], "validation": validation,
                            "expectation": EXPECT.get(scene["name"].split("-coupled")[0], "")}
    if world == 1:
        # per-kernel device time with hipEvents on the solver's stream, same K iterations
        slv.reset()
        prof = slv.profile_kernels(K)
        st2 = slv.stats()
        impl_bytes = kernel_bytes(st2, slv, K)           # this implementation's record sizes, caches included
        alg_bytes, alg_total = survey_bytes(st2, slv, K)   # the figure `achieved` is priced on
        per_launch_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in prof.items()}
        # dominant = largest share of the iteration (a kernel launched once per batch, like k_begin, is not a candidate)
        dom = max((k for k, v in prof.items() if v[1] >= K), key=lambda k: prof[k][0])
        ach = alg_bytes[dom] / (per_launch_ms[dom] * 1e-3) / 1e9
        impl_total = sum(impl_bytes[k] for k, v in prof.items() if v[1])
        # PMC traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_round.sh) is only quoted when the
        # profile was taken on THIS build (same source id) and scene; otherwise null
        pmc, pmc_note = None, "no PMC profile for this build: run tools/profile_round.sh on the GPU box"
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pj.get("scene") != scene["name"] or args.coupled or args.optimal_plane:
                pmc_note = f"profiles/pmc_latest.json is for {pj.get('scene')}, default mode"
            elif pj.get("source_id") != source_id():
                pmc_note = f"profiles/pmc_latest.json was measured on build {pj.get('source_id')}, this is {source_id()}: stale, not quoted"
            else:
                pmc = pj["kernels"].get(dom)
                pmc_note = (f"rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (separate passes) per launch of the same kernel on build {source_id()}; FETCH_SIZE raw -- "
                            "the accesses here are 8-byte scattered reads, which the guide lists as uncalibrated (x2 figure in traffic_detail)")
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "kernel": "tj::" + dom, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                           "traffic": (pmc["hbm_bytes_per_launch"] if pmc else None), "traffic_detail": pmc, "traffic_note": pmc_note,
                           "algorithmic_bytes_per_launch": alg_bytes[dom], "implementation_bytes_per_launch": impl_bytes[dom], "avg_launch_ms": per_launch_ms[dom],
                           "note": "latency-bound at this size: the working set is Infinity-Cache resident (DESIGN.md 5).  achieved = SURVEY 8(d) algorithmic bytes of the dominant kernel / its hipEvent time; "
                                   "implementation_bytes adds the hull / swept-hull caches this implementation moves through HBM",
                           "whole_iteration": {"algorithmic_bytes": alg_total, "achieved_GBps": alg_total / (dt / K) / 1e9, "frac": alg_total / (dt / K) / 1e9 / 8000.0,
                                               "implementation_bytes": impl_total, "implementation_frac": impl_total / (dt / K) / 1e9 / 8000.0},
                           "kernel_ms_per_launch": per_launch_ms, "source_id": source_id(),
                           "critical_path": critical_path(slv, st2, K, per_launch_ms, scene["tris"].shape[0] if scene.get("tris") is not None else scene["cloud"].shape[0], batch_us=1e6 * dt / K)}
        out["timed_window_ms"] = 1e3 * dt
        out["stats_per_iter"] = {k: (v / K if k not in ("error_bits", "order_ambiguous", "iters") else v) for k, v in st2.items()}
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(scene, K, args.optimal_plane)
        # BASELINE configs 3 (8 UAVs) and 5 (256 UAVs x 1M triangles, one GPU's share of the 8-GPU config) under the driver's eyes:
        # same K, same protocol, GPU only; the headline fields above are untouched
        if args.scene == "C" and not (args.no_extra or args.coupled or args.optimal_plane):
            # the headline scene beyond its timed window: 100 more iterations of the same run, the last 40 timed (the steady phase: every robot takes the full step)
            slv.iterate_async(60); slv.sync(); torch.cuda.synchronize()
            t1 = time.perf_counter(); slv.iterate_async(40); slv.sync(); torch.cuda.synchronize()
            steady_ms = 1e3 * (time.perf_counter() - t1) / 40
            slv.close()
            out["extra"] = {"configs": [extra_config(pkg, sc.scn_a(), K, W, local), extra_config(pkg, sc.scn_b(), K, W, local), extra_config(pkg, sc.scn_d_tri(), K, W, local),
                                        extra_config(pkg, dict(sc.scn_c(), mode=2, name="SCN-C-coupled"), K, W, local)],
                            "scn_c_steady_ms_per_step": steady_ms,
                            "note": "BASELINE configs 2 (SCN-A), 3 (SCN-B) and one GPU's view of 5 (SCN-D-tri), then the headline fleet in the coupled mode (decouple:0), timed after the headline line's window on the same "
                                    "process / GPU; scn_c_steady_ms_per_step: iterations 81 - 120 of the headline run; none of it is part of `value`"}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # make the JSON line the LAST thing on stdout: flush whatever native libraries (RCCL banner ...)
    # still hold in C stdio buffers first
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
