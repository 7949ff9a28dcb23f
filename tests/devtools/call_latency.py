import importlib, sys, time
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("traj-opt-admm_amd")
s = pkg.Solver(pkg.scenes.scn_c(), stop=0.0)
s.iterate(50)
for n in (1, 2, 5, 20):
    t0 = time.perf_counter()
    reps = 400 // n
    for _ in range(reps):
        s.iterate(n)
    dt = time.perf_counter() - t0
    print(f"tj_iterate({n}) x {reps}: {1e3 * dt / (reps * n):.4f} ms per iteration, {1e6 * dt / reps:.1f} us per call")
