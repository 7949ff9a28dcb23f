import importlib, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("traj-opt-admm_amd")
sc = pkg.scenes
t0 = time.time()
s = pkg.Solver(sc.scn_c(), stop=0.0)
g, it, conv = s.iterate(3000)
st = s.stats(); fin = s.get_state()
print("3000 iterations SCN-C: gnorm %.3e iters %d error_bits %d finite %s  %.2f s" % (g, it, st["error_bits"], bool(np.isfinite(fin["spline"]).all()), time.time() - t0))
s.close()
import ctypes as C
hip = C.CDLL("libamdhip64.so")
def free_mem():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
free0 = free_mem()
for k in range(30):
    x = pkg.Solver(sc.scn_b(), stop=0.0); x.iterate(3); x.close()
    y = pkg.Solver(dict(sc.scn_b(), mode=2), stop=0.0); y.iterate(3); y.close()
free1 = free_mem()
print("create/destroy x60: free memory before %.1f MB after %.1f MB" % (free0 / 2**20, free1 / 2**20))
# "optimal_plane":1 -- long run (the persistent tables only ever grow) and create/destroy with the planner in the loop
t0 = time.time()
s = pkg.Solver(sc.scn_c(), stop=0.0, optimal_plane=1)
g, it, conv = s.iterate(1500)
st = s.stats(); fin = s.get_state(); on, _ = s.get_pair_cache()
print("1500 iterations SCN-C optimal_plane: gnorm %.3e error_bits %d finite %s stored pair planes %d  %.2f s" % (g, st["error_bits"], bool(np.isfinite(fin["spline"]).all()), int(on.sum()), time.time() - t0))
s.close()
free0 = free_mem()
for k in range(20):
    x = pkg.Solver(sc.scn_b(), stop=0.0, optimal_plane=1); x.iterate(3)
    wp = x.plan_init([[-9.0, -3.0, 0.5]], [[9.0, 3.0, 0.5]], nodes=62)
    x.close()
    y = pkg.Solver(sc.scn_a(), stop=0.0, optimal_plane=1); y.iterate(3); y.get_obs_cache(); y.close()
free1 = free_mem()
print("create/destroy x40 with optimal_plane + planner: free memory before %.1f MB after %.1f MB" % (free0 / 2**20, free1 / 2**20))
# tj_group: uncached receive buffers / flag words / events are per group -- create, run both transports, destroy
# (the first cycles grow a pool of the HIP runtime once -- ~336 MB with three queues active on one device -- which later
#  cycles reuse: tests/devtools/group_leak.py separates the two; the loss is measured after a warm cycle)
for k in range(20):
    gr = pkg.Group(sc.scn_b(), [0, 0, 0], stop=0.0); gr.iterate(3); gr.close()
free0 = free_mem()
for k in range(30):
    gr = pkg.Group(sc.scn_b(), [0, 0, 0], stop=0.0); gr.iterate(3); gr.set_transport("flag"); gr.iterate(3); gr.profile_exchange(3); gr.close()
free1 = free_mem()
print("tj_group create/destroy x30 (3 ranks on device 0, both transports): free memory before %.1f MB after %.1f MB" % (free0 / 2**20, free1 / 2**20))
