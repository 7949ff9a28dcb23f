"""Where does device memory go over tj_group create / destroy cycles?  (development aid, GPU box)"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("traj-opt-admm_amd"); sc = importlib.import_module("traj-opt-admm_amd.scenes")
hip = C.CDLL("libamdhip64.so")
def free_mem():
    f = C.c_size_t(); t = C.c_size_t(); hip.hipDeviceSynchronize(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value
def report(tag, f0):
    print("%-60s lost %8.2f MB" % (tag, (f0 - free_mem()) / 2**20), flush=True)
s = pkg.Solver(sc.scn_b(), stop=0.0); s.iterate(2); s.close()
f0 = free_mem()
for k in range(200):
    p = C.c_void_p(); assert hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4096), C.c_uint(3)) == 0; hip.hipMemset(p, 0, C.c_size_t(4096)); hip.hipFree(p)
report("200 x hipExtMallocWithFlags(4 KB, uncached) + hipFree", f0)
f0 = free_mem()
for k in range(200):
    e = C.c_void_p(); assert hip.hipEventCreateWithFlags(C.byref(e), C.c_uint(0x2 | 0x80000000)) == 0; hip.hipEventDestroy(e)
report("200 x event create(ReleaseToSystem) + destroy", f0)
f0 = free_mem()
for k in range(20):
    g = pkg.Group(sc.scn_b(), [0, 0, 0], stop=0.0); g.close()
report("20 x group create + destroy (no iterations)", f0)
for rep in range(4):   # repeated: a pool of the runtime that grows once is not a leak, a loss per repetition is
    f0 = free_mem()
    for k in range(20):
        g = pkg.Group(sc.scn_b(), [0, 0, 0], stop=0.0); g.iterate(3); g.close()
    report("20 x group create + 3 iterations (event) + destroy, repetition %d" % rep, f0)
f0 = free_mem()
for k in range(20):
    g = pkg.Group(sc.scn_b(), [0, 0, 0], stop=0.0); g.set_transport("flag"); g.iterate(3); g.close()
report("20 x group create + 3 iterations (flag) + destroy", f0)
f0 = free_mem()
for k in range(20):
    g = pkg.Group(sc.scn_b(), [0, 0, 0], stop=0.0); g.profile_exchange(3); g.close()
report("20 x group create + profile_exchange + destroy", f0)
f0 = free_mem()
g = pkg.Group(sc.scn_b(), [0, 0, 0], stop=0.0)
for k in range(100): g.iterate(3)
report("one group, 100 x iterate(3)", f0)
g.close()
report("  ... after destroy", f0)
f0 = free_mem()
for k in range(60):
    s = pkg.Solver(sc.scn_b(), stop=0.0); s.iterate(3); s.close()
report("60 x solver create + 3 iterations + destroy", f0)
