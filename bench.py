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


def stage_bytes(st, slv, iters):
    """Algorithmic HBM bytes per iteration of each stage (SURVEY 8d terms, device-counted where data
    dependent; this implementation's record sizes: BVH box 48 B, point 24 B, plane 32 B)."""
    U, S, P, T = slv.U, slv.S, slv.P, slv.T
    it = max(1, iters)
    hull_in = U * S * (18 * 8 + 36 * 8)            # 6 control points x 3 + 6x6 basis per (robot, segment)
    planes = (st["planes_obs"] + st["planes_self"]) / it
    b = {}
    b["planes_obs"] = st["nodes_dcd"] / it * 48 + st["cand_dcd"] / it * 24 + st["planes_obs"] / it * 32 + hull_in
    b["planes_self"] = st["pair_tests"] / it * 144 + st["planes_self"] / it * 32 + hull_in
    b["grad"] = planes * 32 + hull_in * 2 + U * P * (19 + 361) * 8 + U * P * (36 + 2 * 18) * 8
    b["xsolve"] = U * P * (19 + 361) * 8 + U * (3 * T + 4) * 8
    b["ccd_prep"] = U * S * (2 * 18 * 8 + 36 * 8) + U * S * 146 * 8
    b["ccd_obs"] = st["nodes_ccd"] / it * 48 + st["cand_ccd"] / it * 24 + U * S * 146 * 8
    b["ccd_self"] = S * (U * (U - 1) / 2) * 2 * (6 + 98) * 8 / 8  # boxes always, k-DOP intervals for ~1/8 of the pairs
    b["linesearch"] = st["energy_evals"] / it * (planes / U * 32 + 3 * T * 8 + S * 36 * 8) + U * 2 * 3 * T * 8
    b["slack"] = U * P * (3 * 18 * 8 * 2 + 36 * 8 * 2)
    b["begin"] = 64
    b["end"] = 8
    return b


def cpu_baseline(scene, steps):
    """Reference CPU path on this box's host cores: the unmodified reference (oracle/_ref/libref.so,
    prebuilt in the dev container) if present, else this repo's CPU restatement.  Single thread --
    the reference has no threading (no `#pragma omp` anywhere in its first-party code)."""
    from oracle import pyoracle
    kind = "reference" if pyoracle.available("ref") else "port"
    eng = pyoracle.Engine("ref" if kind == "reference" else "port", scene)
    n = min(steps, 20)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.iterate()
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "iters/s", "ms_per_iter": 1e3 * dt / n, "cores": 1, "kind": kind,
            "sample": f"the first {n} ADMM iterations of the same scene from the same initial trajectory (BVH build excluded)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="C", choices=["A", "B", "C", "D", "H8"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the sharded schedule + RCCL collectives even with one rank (self test)")
    args = ap.parse_args()

    pkg = importlib.import_module("traj-opt-admm_amd")
    sc = pkg.scenes
    scene = {"A": sc.scn_a, "B": sc.scn_b, "C": sc.scn_c, "D": sc.scn_d, "H8": lambda: sc.hard(8, 20000)}[args.scene]()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    dist = None
    sharded = world > 1 or args.force_dist
    torch.cuda.set_device(local)   # torch initialises the HIP runtime first; the library then shares it
    if sharded:
        import torch.distributed as dist
        if "TJ_KEEP_NCCL_DEBUG" not in os.environ:
            os.environ["NCCL_DEBUG"] = "WARN"   # no version banner on stdout next to the JSON line
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    if scene["U"] % world != 0:
        raise SystemExit("robot count must divide evenly over the ranks")
    slv = pkg.Solver(scene, device=local, rank=rank, world=world, stop=0.0)  # stop test off: time exactly K iterations
    K, W = args.steps, args.warmup

    if sharded:
        # kernels and collectives are ordered on one dedicated torch stream
        tstream = torch.cuda.Stream(device=local)
        torch.cuda.set_stream(tstream)
        slv.set_stream(tstream.cuda_stream)
        views = []
        for what in (0, 1):
            ptr, per, first, n = slv.exchange_buffer(what)
            full = torch.as_tensor(_DevView(ptr, per * slv.U), device=f"cuda:{local}")
            views.append((full, full[first * per:(first + n) * per]))

        sharding = importlib.import_module("traj-opt-admm_amd.sharding")

        class _Eng:
            @staticmethod
            def phase(k):
                slv.iterate_phase(k)

        def _gather(what):  # RCCL all-gather straight on the library's device buffers (in place)
            dist.all_gather_into_tensor(views[what][0], views[what][1])

        def run(n_it):
            sharding.run_sharded(_Eng, _gather, n_it)
    else:
        def run(n_it):
            slv.iterate_async(n_it)

    def barrier():
        if dist is not None:
            dist.barrier()
        slv.sync()
        torch.cuda.synchronize()

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

    out = None
    if rank == 0:
        out = {"metric": "ADMM iterations/sec", "value": K / dt, "unit": "iters/s", "n_gpus": world, "steps": K, "warmup": W,
               "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{scene['name']}: {scene['U']} UAVs crossing, {scene['cloud'].shape[0]} obstacle points, "
                                      f"{scene['P']} pieces x res 8 = {slv.S} segments/robot, decoupled mode (3D.json defaults)",
                          "parallelism": f"robots sharded over {world} GPU(s), 2 all-gathers/iter" if world > 1 else "1 GPU, whole iteration in one hipGraph",
                          "iters_timed_from": "initial trajectory"}}
    if world == 1:
        # per-kernel device time with hipEvents on the solver's stream, same K iterations
        slv.reset()
        prof = slv.profile_iterations(K)
        st2 = slv.stats()
        bytes_it = stage_bytes(st2, slv, K)
        launches = {k: v[1] for k, v in prof.items()}
        per_launch_ms = {k: (v[0] / max(1, v[1])) for k, v in prof.items()}
        dom = max(prof, key=lambda k: prof[k][0])
        kern_ms = prof[dom][0] / K
        ach = bytes_it[dom] / (kern_ms * 1e-3) / 1e9
        total_bytes = sum(bytes_it.values())
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                           "traffic": None, "algorithmic_bytes_per_launch": bytes_it[dom] / max(1, launches[dom] // K),
                           "avg_launch_ms": per_launch_ms[dom],
                           "whole_iteration": {"algorithmic_bytes": total_bytes, "achieved_GBps": total_bytes / (dt / K) / 1e9,
                                               "frac": total_bytes / (dt / K) / 1e9 / 8000.0},
                           "stage_ms_per_iter": {k: v[0] / K for k, v in prof.items()}}
        out["stats_per_iter"] = {k: (v / K if k not in ("error_bits", "order_ambiguous", "iters") else v) for k, v in st2.items()}
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(scene, K)
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
