#!/usr/bin/env python3
"""Rank-isolated replay (GPU, one device): what ONE rank's chain of a robot-sharded run costs at N = 2, 4, 8 -- measured on the one GPU a box has.

No scaling curve can be measured without an 8-GPU node; this can.  A one-context run records, per iteration, what the two exchanges of the sharded schedule carry
(every robot's control points before the iteration, every robot's direction record of the iteration: tj_exchange_buffer 0 / 1).  Then rank r of world N runs ALONE
on the device through the chained phase API (tj_iterate_phase_chained: the same six kernels per iteration a rank of tj_group / torchrun runs), and before each
consuming phase the OTHER ranks' slices are copied in from the recording -- a teacher-forced exchange without a peer.  The rank's owned robots must end bit for bit in
the one-context run's state (asserted); its time per iteration (with the two copies, one kernel each, standing where the exchanges stand) is the rank's chain cost; the copies are also timed alone, and the exchange latency measured
between two ranks on one device (profiles/round5b_*: 8.8 / 9.4 us per exchange) is the additive term of the PROJECTED strong-scaling table -- a prediction the first
real multi-GPU run can be checked against, with its assumptions in the output.

  python tools/rank_replay.py --scene C --worlds 1,2,4,8 --steps 20 --out gpurun_out/rank_replay_C.json
"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DevView:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="C")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warm", type=int, default=3, help="iterations replayed before the timed ones (they change the state: part of the recording)")
    ap.add_argument("--ranks", default="all", help="'all' or 'ends' (first, middle, last rank of each world)")
    ap.add_argument("--exchange-us", type=float, default=9.1, help="latency of one exchange between two ranks (mean of the two measured on one device in round 5)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    pkg = importlib.import_module("traj-opt-admm_amd")
    sc = pkg.scenes
    scene = {"B": sc.scn_b, "C": sc.scn_c, "D": sc.scn_d, "Dtri": sc.scn_d_tri, "E": sc.scn_e}[a.scene]()
    U = scene["U"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_it = a.warm + a.steps

    # ---- 1. the recording: one context, iteration by iteration ----
    ref = pkg.Solver(scene, stop=0.0)
    p0, per0, _, _ = ref.exchange_buffer(0); p1, per1, _, _ = ref.exchange_buffer(1)
    v0 = torch.as_tensor(DevView(p0, per0 * U), device=dev); v1 = torch.as_tensor(DevView(p1, per1 * U), device=dev)
    rec_x, rec_d = [], []
    for i in range(n_it):
        ref.sync()
        rec_x.append(v0.clone())
        ref.iterate(1)
        rec_d.append(v1.clone())
    ref_state = ref.get_state()
    # the one-context chain itself, timed like bench.py (K iterations in one batch, from the initial trajectory)
    ref.reset(); ref.iterate_async(a.warm); ref.sync()
    t0 = time.perf_counter(); ref.iterate_async(a.steps); ref.sync(); t_one = (time.perf_counter() - t0) / a.steps * 1e3
    ref.close()

    out = {"scene": scene["name"], "robots": U, "steps": a.steps, "warm": a.warm, "one_context_ms_per_iter": round(t_one, 4), "worlds": {},
           "method": "rank r of world N alone on the device, chained phases (tj_iterate_phase_chained), the other ranks' slices copied in from a one-context recording before "
                     "each consuming phase; owned robots bitwise equal to the one-context run (asserted); copy time measured alone and subtracted",
           "assumptions": ["the exchange between real peers costs what it cost between two ranks on one device (%.1f us each, two per iteration) and is not hidden" % a.exchange_us,
                           "ranks run their chains at the speed measured alone on a device of their own (no interference between devices)",
                           "an iteration is as long as its slowest rank's chain + the two exchanges (every rank waits for every peer twice per iteration)"]}
    for N in [int(x) for x in a.worlds.split(",")]:
        if U % N:
            continue
        ranks = list(range(N)) if a.ranks == "all" else sorted({0, N // 2, N - 1})
        rows = []
        for r in ranks:
            s = pkg.Solver(scene, rank=r, world=N, stop=0.0)
            stream = torch.cuda.Stream(device=dev)
            s.set_stream(stream.cuda_stream)
            q0, _, first, own = s.exchange_buffer(0); q1, _, _, _ = s.exchange_buffer(1)
            w0 = torch.as_tensor(DevView(q0, per0 * U), device=dev); w1 = torch.as_tensor(DevView(q1, per1 * U), device=dev)
            lo, hi = first, first + own

            def inject(dst, src, per):   # ONE copy per exchange: the whole buffer -- the owned slice of the recording holds the very bits this rank has just produced (asserted at the end)
                dst.copy_(src, non_blocking=True)

            def run(i0, i1, do_phases=True, do_copies=True):
                with torch.cuda.stream(stream):
                    for i in range(i0, i1):
                        if do_phases: s.iterate_phase(0)
                        if do_copies and N > 1: inject(w0, rec_x[i], per0)
                        if do_phases: s.iterate_phase(1)
                        if do_copies and N > 1: inject(w1, rec_d[i], per1)
                        if do_phases: s.iterate_phase(2, i + 1 < i1)
                stream.synchronize()

            run(0, a.warm)
            t0 = time.perf_counter(); run(a.warm, n_it); t_chain = (time.perf_counter() - t0) / a.steps * 1e3
            s.sync()
            st = s.get_state()
            for k in st:
                assert np.array_equal(st[k][lo:hi], ref_state[k][lo:hi]), f"world {N} rank {r}: {k} of the owned robots differs from the one-context run"
            err = s.stats()["error_bits"]
            assert err == 0, err
            # the copies alone (same stream, same sizes), for the subtraction
            t0 = time.perf_counter(); run(a.warm, n_it, do_phases=False); t_copy = (time.perf_counter() - t0) / a.steps * 1e3
            s.close()
            rows.append({"rank": r, "owned": own, "ms_per_iter_with_copies": round(t_chain, 4), "copies_alone_ms": round(t_copy, 4), "chain_ms": round(t_chain - t_copy, 4)})
            print(f"world {N} rank {r}: owned {own}  chain+copies {t_chain:.4f}  copies {t_copy:.4f}  chain {t_chain - t_copy:.4f} ms/iter  (bitwise == one context)", flush=True)
        slowest = max(x["chain_ms"] for x in rows)
        proj = slowest + (2 * a.exchange_us * 1e-3 if N > 1 else 0.0)
        out["worlds"][str(N)] = {"ranks": rows, "slowest_chain_ms": round(slowest, 4), "projected_ms_per_iter": round(proj, 4)}
    base = out["worlds"].get("1", {}).get("projected_ms_per_iter")
    for N, w in out["worlds"].items():
        w["projected_speedup_vs_one_context"] = round(t_one / w["projected_ms_per_iter"], 3)
        if base: w["projected_speedup_vs_one_rank_schedule"] = round(base / w["projected_ms_per_iter"], 3)
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
