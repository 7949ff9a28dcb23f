"""Robot sharding across ranks: who owns which robots and the per-iteration exchange schedule.

The reference is single-process.  Cross-robot data dependencies of one decoupled ADMM iteration
(SURVEY 3.3): `separate_self` needs every robot's control points (Optimization3D_multi.h:246-259),
`Step::self_step` needs every robot's control points AND search directions (Step.h:196-208), the
line search uses the LAST robot's `wolfe` (Optimization3D_multi.h:730 vs :792) and `gnorm` is the mean
of all |g| (:57,72).  Everything else is per robot.  Hence two all-gathers per iteration:

    phase 0   stop test                                    } gather 0 may run concurrently with phase 0
    gather 0  control points of all robots                 } (run_sharded's gather_begin)
    phase 1   obstacle planes of owned robots, robot-pair planes, gradient/Hessian, Newton direction (owned robots)
    gather 1  direction records (direction, t_direction, wolfe, |g|) of all robots
    phase 2   CCD clamps (pair clamp replicated on every rank), line search, slack + dual (owned)

`run_sharded` is the one schedule used both by bench.py (HIP engine, RCCL all-gather on device
buffers) and by the CPU test (oracle engine, gloo all-gather), so the N>1 path is covered without
a multi-GPU box.
"""


def owned_range(n_robots, rank, world):
    """Block partition: rank r owns robots [r*U/world, (r+1)*U/world) -- same formula as tj_create."""
    return (rank * n_robots) // world, ((rank + 1) * n_robots) // world


# Coupled mode ("decouple":0, one piece_time shared by all robots): the arrowhead Newton system, the single CCD step and the
# Armijo test on the summed energy each need a small contribution from every robot (Optimization3D_multi.h:508-639,
# Step.h:112-182), so an iteration has six phases and five exchanges:
#   phase 0 stop test | gather 0 control points | phase 1 planes, gradient, per-robot elimination | gather 2 Schur-corner terms |
#   phase 2 corner pivot + back substitution | gather 1 directions | phase 3 CCD clamps, shared step, gnorm |
#   gather 3 obstacle CCD exponents | phase 4 Armijo candidates | gather 4 their energies | phase 5 commit
COUPLED_SCHEDULE = ((0, 0), (1, 2), (2, 1), (3, 3), (4, 4), (5, None))
DECOUPLED_SCHEDULE = ((0, 0), (1, 1), (2, None))


def run_schedule(engine, gather, n_iters, schedule):
    """(phase, what to all-gather after it) pairs, n_iters times.
    Coupled schedule, engines that FOLLOW the Armijo search (engine.pending(), the HIP engine's tj_coupled_search_pending): while the search is pending after phase 5
    -- none of the candidates one exchange carries passed; the reference's loop has no bound (Optimization3D_multi.h:623) -- phase 4, gather 4 and phase 5 run again for
    the next candidates.  Every rank gets the same answer, so the ranks stay in step."""
    pending = getattr(engine, "pending", None)
    for _ in range(n_iters):
        for phase, what in schedule:
            engine.phase(phase)
            if what is not None:
                gather(what)
        if pending is not None and schedule is COUPLED_SCHEDULE:
            while pending():
                engine.phase(4); gather(4); engine.phase(5)


def run_sharded(engine, gather, n_iters, gather_begin=None):
    """engine.phase(k) runs phase k for the robots this rank owns; gather(what) all-gathers buffer
    `what` (0 = control points, 1 = direction records) in place.

    gather_begin(what) (optional) STARTS that all-gather and returns a callable that completes it.  Phase 0 touches no
    exchanged buffer and the control points are final when the previous phase 2 ends, so gather 0 may be started before
    phase 0 and completed after it."""
    chained = getattr(engine, "chained", False)   # engine.phase(k, more): the HIP engine folds the next iteration's begin into phase 2 (tj_iterate_phase_chained)
    for it in range(n_iters):
        finish = gather_begin(0) if gather_begin is not None else None
        engine.phase(0)
        if finish is not None:
            finish()
        else:
            gather(0)
        engine.phase(1)
        gather(1)
        if chained:
            engine.phase(2, it + 1 < n_iters)
        else:
            engine.phase(2)
