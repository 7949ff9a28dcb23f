"""Python host-side mirror of the C ABI in include/trajadmm.h (libtrajadmm.so).

The reference is compiled C++ with no Python layer; this module exists for the test-suite,
bench.py and for Python callers, and is a thin ctypes binding -- all numerics run in the HIP
kernels behind the C ABI.  There is NO CPU fallback: importing works anywhere (so that
`-m "not gpu"` tests can check the exported symbols), but creating a `Solver` without the
built library or without a HIP device raises.

Method names follow the reference's driver vocabulary
(Optimization3D_multi::optimization_decouple and its stages, Optimization3D_multi.h:29-118).
"""
import ctypes as C
import os
import numpy as np

from . import scenes  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRAJADMM_LIB") or os.path.join(_HERE, "libtrajadmm.so")  # override: another BUILD of this library (e.g. `make timing`)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

EXPORTS = [
    "tj_default_params", "tj_create", "tj_destroy", "tj_last_error", "tj_set_cloud", "tj_set_mesh", "tj_init_state", "tj_get_state",
    "tj_set_state", "tj_iterate", "tj_iterate_async", "tj_sync", "tj_stream", "tj_run_stage", "tj_get_planes", "tj_get_candidates",
    "tj_set_planes", "tj_get_direction", "tj_set_direction", "tj_get_local_grad", "tj_get_steps", "tj_get_energy", "tj_get_stats", "tj_get_build_info", "tj_exchange_buffer",
    "tj_iterate_phase", "tj_iterate_phase_chained", "tj_set_coupled_follow", "tj_coupled_search_pending", "tj_launch_count", "tj_phase_count", "tj_xch_block", "tj_xch_ipc_export", "tj_xch_ipc_open", "tj_xch_attach", "tj_xch_enable", "tj_set_stream", "tj_host_tables", "tj_profile_kernels", "tj_kernel_count", "tj_kernel_name",
    "tj_get_obs_cache", "tj_set_obs_cache", "tj_get_pair_cache", "tj_set_pair_cache", "tj_edge_collision", "tj_plan_init",
    "tj_group_create", "tj_group_destroy", "tj_group_size", "tj_group_ctx", "tj_group_last_error", "tj_group_set_cloud", "tj_group_set_mesh",
    "tj_group_transport", "tj_group_set_transport", "tj_group_profile_exchange", "tj_rccl_available", "tj_group_rccl_ranks",
    "tj_group_init_state", "tj_group_iterate", "tj_group_get_state",
]

STAGES = dict(begin=0, planes_obs=1, planes_self=2, grad=3, xsolve=4, ccd_prep=5, ccd_obs=6, ccd_self=7, linesearch=8, slack=9, end=10)


class TjParams(C.Structure):
    _fields_ = [("mode", C.c_int), ("uav_num", C.c_int), ("piece_num", C.c_int), ("res", C.c_int),
                ("lambda_", C.c_double), ("margin", C.c_double), ("offset", C.c_double), ("mu", C.c_double),
                ("vel_limit", C.c_double), ("acc_limit", C.c_double), ("ks", C.c_double), ("kt", C.c_double),
                ("stop", C.c_double), ("device", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("cap_obs", C.c_int), ("cap_self", C.c_int), ("cap_pairs", C.c_int), ("optimal_plane", C.c_int)]


class TjStats(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("iters", "nodes_dcd", "nodes_ccd", "cand_dcd", "cand_ccd", "planes_obs",
                                             "planes_self", "energy_evals", "pair_tests", "llt_fail_piece", "llt_fail_robot", "newton_iters", "pair_solves")] + \
               [("order_ambiguous", C.c_int), ("error_bits", C.c_int), ("order_unresolved", C.c_int), ("head_starts", C.c_int), ("gjk_max_sum", C.c_ulonglong),
                ("ls_giveups", C.c_int), ("ls_helper_timeouts", C.c_int), ("async_fallbacks", C.c_int)]


class TrajAdmmError(RuntimeError):
    pass


# the known-answer hooks (include/trajadmm_kat.h) live in a TEST build of the same translation unit, never in the product library
KAT_EXPORTS = ["tj_kat_gjk", "tj_kat_gjk_wave", "tj_kat_gjk_wave_split", "tj_kat_planes", "tj_kat_ccd", "tj_kat_tri", "tj_kat_query", "tj_kat_linalg"]
KAT_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libtrajadmm_kat.so")
_lib = None
_kat_lib = None


def _open(path):
    if not os.path.exists(path):
        raise TrajAdmmError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.tj_last_error.restype = C.c_char_p
    lib.tj_stream.restype = C.c_void_p
    lib.tj_group_last_error.restype = C.c_char_p
    lib.tj_group_ctx.restype = C.c_void_p
    return lib


def load_library(kat=False):
    """dlopen libtrajadmm.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if absent.  kat=True: the test build
    libtrajadmm_kat.so (same sources + the tj_kat_* hooks), unless TRAJADMM_LIB points somewhere else."""
    global _lib, _kat_lib
    if kat and "TRAJADMM_LIB" not in os.environ:
        if _kat_lib is None:
            _kat_lib = _open(KAT_LIB_PATH)
        return _kat_lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def host_tables(piece_num, res=8):
    """(convert[P,6,6], mdyn[6,6], basis[P*res,6,6], kdop[49,3]) as the library precomputes them on the host
    (tj_host_tables; needs no GPU).  Row-major [row, col]."""
    lib = load_library()
    conv = np.zeros((piece_num, 6, 6)); M = np.zeros((6, 6)); basis = np.zeros((piece_num * res, 6, 6)); kd = np.zeros((49, 3))
    rc = lib.tj_host_tables(C.c_int(piece_num), C.c_int(res), _d(conv), _d(M), _d(basis), _d(kd))
    if rc < 0:
        raise TrajAdmmError(f"tj_host_tables failed ({rc})")
    return conv, M, basis, kd


def _i(a):
    return a.ctypes.data_as(_ip)


class Solver:
    """One ADMM problem resident on one GPU (one `tj_ctx`)."""

    def __init__(self, scene, params=None, device=0, rank=0, world=1, stop=None, kat=False, **caps):
        self.lib = load_library(kat)          # kat=True: the test build with the known-answer hooks (kat_* methods)
        p = dict(scenes.DEFAULT_PARAMS)
        if params:
            p.update(params)
        self.params = p
        self.mode, self.U, self.P = scene["mode"], scene["U"], scene["P"]
        self.res = p["res"]
        self.S, self.T = self.P * self.res, 3 * self.P + 3
        self.box_bytes, self.prim_vertices = 24, 1   # BVH box record; vertices per obstacle primitive (bench.py's byte model)
        tp = TjParams()
        self.lib.tj_default_params(C.byref(tp), self.mode, self.U, self.P)
        tp.res = self.res
        tp.lambda_, tp.margin, tp.offset, tp.mu = p["lam"], p["margin"], p["offset"], p["mu"]
        tp.vel_limit, tp.acc_limit, tp.ks, tp.kt = p["vel_limit"], p["acc_limit"], scene["ks"], p["kt"]
        tp.stop = p["stop"] if stop is None else stop
        tp.device, tp.rank, tp.world = device, rank, world
        for k, v in caps.items():
            setattr(tp, k, v)
        self._ctx = C.c_void_p()
        rc = self.lib.tj_create(C.byref(tp), C.byref(self._ctx))
        self._check(rc)
        if scene.get("tris") is not None:   # obstacle triangles [N][3][3] (BASELINE config 5): tj_set_mesh with an unshared vertex list
            verts = np.ascontiguousarray(scene["tris"], dtype=np.float64).reshape(-1, 3)
            self.N = verts.shape[0] // 3
            faces = np.arange(3 * self.N, dtype=np.int32).reshape(-1, 3)
            self._check(self.lib.tj_set_mesh(self._ctx, _d(verts), C.c_int(3 * self.N), _i(faces), C.c_int(self.N)))
            self.prim_vertices = 3
        else:
            cloud = np.ascontiguousarray(scene["cloud"], dtype=np.float64).reshape(-1, 3)
            self.N = cloud.shape[0]
            self._check(self.lib.tj_set_cloud(self._ctx, _d(cloud), C.c_int(self.N)))
        self._wp = np.ascontiguousarray(scene["waypoints"], dtype=np.float64)
        self._pt0 = float(p["piece_time0"])
        self.reset()

    def reset(self):
        """init_variable: back to the initial trajectory, iteration counter 0."""
        self._check(self.lib.tj_init_state(self._ctx, _d(self._wp), C.c_double(self._pt0)))

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self.lib.tj_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            msg = self.lib.tj_last_error(self._ctx).decode() if self._ctx.value else "tj_create failed"
            raise TrajAdmmError(f"libtrajadmm error {rc}: {msg}")
        return rc

    # ---- state ---------------------------------------------------------------------------
    def get_state(self):
        U, P, T = self.U, self.P, self.T
        st = dict(spline=np.zeros((U, 3, T)), p_slack=np.zeros((U, 3, 6 * P)), p_lambda=np.zeros((U, 3, 6 * P)),
                  t_slack=np.zeros((U, P)), t_lambda=np.zeros((U, P)), piece_time=np.zeros(U))
        for u in range(U):
            pt = C.c_double()
            self._check(self.lib.tj_get_state(self._ctx, u, _d(st["spline"][u]), _d(st["p_slack"][u]), _d(st["p_lambda"][u]),
                                              _d(st["t_slack"][u]), _d(st["t_lambda"][u]), C.byref(pt)))
            st["piece_time"][u] = pt.value
        return st

    def set_state(self, st):
        for u in range(self.U):
            a = [np.ascontiguousarray(st[k][u], dtype=np.float64) for k in ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda")]
            self._check(self.lib.tj_set_state(self._ctx, u, _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(a[4]), C.c_double(float(st["piece_time"][u]))))

    # ---- hot path ------------------------------------------------------------------------
    def iterate(self, n=1):
        """n ADMM iterations on the device; returns (gnorm, iter, converged)."""
        g, it, cv = C.c_double(), C.c_int(), C.c_int()
        self._check(self.lib.tj_iterate(self._ctx, C.c_int(n), C.byref(g), C.byref(it), C.byref(cv)))
        return g.value, it.value, bool(cv.value)

    def iterate_async(self, n=1):
        self._check(self.lib.tj_iterate_async(self._ctx, C.c_int(n)))

    def sync(self):
        self._check(self.lib.tj_sync(self._ctx))

    def stream(self):
        return self.lib.tj_stream(self._ctx)

    def set_stream(self, hip_stream):
        self._check(self.lib.tj_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def profile_kernels(self, n):
        """{kernel name: (device ms summed over n iterations, launches)} measured with hipEvents on the solver's stream"""
        self.lib.tj_kernel_name.restype = C.c_char_p
        nk = self.lib.tj_kernel_count()
        ms = np.zeros(nk); ln = np.zeros(nk, dtype=np.int32)
        self._check(self.lib.tj_profile_kernels(self._ctx, C.c_int(n), _d(ms), _i(ln)))
        return {self.lib.tj_kernel_name(i).decode(): (float(ms[i]), int(ln[i])) for i in range(nk)}

    def run_stage(self, name):
        self._check(self.lib.tj_run_stage(self._ctx, C.c_int(STAGES[name])))

    def phase_count(self):
        return self._check(self.lib.tj_phase_count(self._ctx))

    def launch_count(self):
        self.lib.tj_launch_count.restype = C.c_longlong
        return int(self.lib.tj_launch_count(self._ctx))

    def iterate_phase(self, phase, more=0):
        """more: another iteration follows in this batch (decoupled schedules then fold its begin into this one's line search)"""
        self._check(self.lib.tj_iterate_phase_chained(self._ctx, C.c_int(phase), C.c_int(1 if more else 0)))

    def set_coupled_follow(self, on=True):
        """coupled mode, sharded context: follow the Armijo search beyond the candidates one exchange carries (ask coupled_search_pending() after phase 5)"""
        self._check(self.lib.tj_set_coupled_follow(self._ctx, C.c_int(1 if on else 0)))

    def coupled_search_pending(self):
        p = C.c_int()
        self._check(self.lib.tj_coupled_search_pending(self._ctx, C.byref(p)))
        return bool(p.value)

    # ---- direct exchange between processes (include/trajadmm.h tj_xch_*) ----
    def xch_ipc_export(self):
        h = (C.c_ubyte * 64)()
        self._check(self.lib.tj_xch_ipc_export(self._ctx, h))
        return bytes(h)

    def xch_attach_ipc(self, handles, poll_in_kernel=True):
        """handles: {rank: 64-byte handle} of every OTHER rank; opens them, attaches and switches the direct exchange on"""
        ranks = sorted(handles)
        bases = (C.c_void_p * len(ranks))()
        for i, r in enumerate(ranks):
            b = C.c_void_p()
            hb = (C.c_ubyte * 64).from_buffer_copy(handles[r])
            self._check(self.lib.tj_xch_ipc_open(self._ctx, hb, C.byref(b)))
            bases[i] = b.value
        rk = np.ascontiguousarray(ranks, dtype=np.int32)
        self._check(self.lib.tj_xch_attach(self._ctx, C.c_int(len(ranks)), _i(rk), bases))
        self._check(self.lib.tj_xch_enable(self._ctx, C.c_int(1), C.c_int(1 if poll_in_kernel else 0)))

    def xch_enable(self, on, poll_in_kernel=True):
        self._check(self.lib.tj_xch_enable(self._ctx, C.c_int(1 if on else 0), C.c_int(1 if poll_in_kernel else 0)))

    # ---- stage-level views (same shapes as oracle.pyoracle.Engine) ---------------------------
    def stage_planes(self):
        self.run_stage("begin")
        self.run_stage("planes_obs")
        self.run_stage("planes_self")
        return self.get_planes()

    def get_planes(self):
        counts = np.zeros((self.U, self.S), dtype=np.int32)
        chunks = []
        for u in range(self.U):
            co = np.zeros(self.S, dtype=np.int32); cs = np.zeros(self.S, dtype=np.int32)
            n = self._check(self.lib.tj_get_planes(self._ctx, u, _i(co), _i(cs), None, 0))
            buf = np.zeros((max(n, 1), 4))
            self._check(self.lib.tj_get_planes(self._ctx, u, _i(co), _i(cs), _d(buf), C.c_int(n)))
            counts[u] = co + cs
            chunks.append(buf[:n])
        return counts, np.concatenate(chunks, axis=0)

    def get_candidates(self, u, cap=4096):
        """per segment: (obstacle ids that passed box query + k-DOP cull, number the box query alone returned since reset)"""
        out = []
        for tr in range(self.S):
            ids = np.zeros(cap, dtype=np.int32); nb = C.c_int()
            n = self._check(self.lib.tj_get_candidates(self._ctx, C.c_int(u), C.c_int(tr), C.c_int(cap), _i(ids), C.byref(nb)))
            out.append((ids[:n].copy(), nb.value))
        return out

    def set_planes(self, counts, planes):
        counts = np.ascontiguousarray(counts, dtype=np.int32).reshape(self.U, self.S)
        planes = np.ascontiguousarray(planes, dtype=np.float64).reshape(-1, 4)
        w = 0
        for u in range(self.U):
            n = int(counts[u].sum())
            blk = np.ascontiguousarray(planes[w:w + n]) if n else np.zeros((1, 4))
            self._check(self.lib.tj_set_planes(self._ctx, u, _i(np.ascontiguousarray(counts[u])), _d(blk)))
            w += n

    def stage_direction(self):
        self.run_stage("grad")
        self.run_stage("xsolve")
        out = dict(direction=np.zeros((self.U, 3, self.T)), t_direction=np.zeros(self.U), wolfe=np.zeros(self.U), gn=np.zeros(self.U))
        for u in range(self.U):
            a, b, c = C.c_double(), C.c_double(), C.c_double()
            self._check(self.lib.tj_get_direction(self._ctx, u, _d(out["direction"][u]), C.byref(a), C.byref(b), C.byref(c)))
            out["t_direction"][u], out["wolfe"][u], out["gn"][u] = a.value, b.value, c.value
        out["gnorm"] = float(np.sum(out["gn"]) / self.U) if self.mode == 1 else float(out["gn"][0])
        return out

    def set_direction(self, u, direction, t_direction, wolfe, gn):
        d = np.ascontiguousarray(direction, dtype=np.float64)
        self._check(self.lib.tj_set_direction(self._ctx, C.c_int(u), _d(d), C.c_double(t_direction), C.c_double(wolfe), C.c_double(gn)))

    def local_grad(self, u, sp):
        g = np.zeros(19); h = np.zeros((19, 19))
        self._check(self.lib.tj_get_local_grad(self._ctx, u, sp, _d(g), _d(h)))
        return g, h

    def stage_steps(self):
        self.run_stage("ccd_prep")
        self.run_stage("ccd_obs")
        self.run_stage("ccd_self")
        a = np.zeros(self.U); b = np.zeros(self.U)
        self._check(self.lib.tj_get_steps(self._ctx, _d(a), _d(b), None))
        return a, b

    def stage_linesearch(self):
        self.run_stage("linesearch")
        s = np.zeros(self.U)
        self._check(self.lib.tj_get_steps(self._ctx, None, None, _d(s)))
        return s

    def last_armijo_steps(self):
        """accepted Armijo step of every robot in the last iteration (diagnostic; tj_get_steps)"""
        s = np.zeros(self.U)
        self._check(self.lib.tj_get_steps(self._ctx, None, None, _d(s)))
        return s

    def stage_slack(self):
        self.run_stage("slack")
        self.run_stage("end")

    def stage_update_spline(self):
        """coupled mode (scene mode 2): Optimization3D_multi::update_spline as one stage -- arrowhead Newton
        system, CCD clamps, Armijo search on the summed energy, commit.  Returns (gnorm, wolfe)."""
        for st in ("grad", "xsolve", "ccd_prep", "ccd_obs", "ccd_self", "linesearch"):
            self.run_stage(st)
        t, w, g = C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.tj_get_direction(self._ctx, 0, None, C.byref(t), C.byref(w), C.byref(g)))
        return g.value, w.value

    # ---- known-answer hooks (device primitives on caller batches) -------------------------------
    def kat_gjk(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        n = a.shape[0]; v = np.zeros((n, 3))
        self._check(self.lib.tj_kat_gjk(self._ctx, C.c_int(n), C.c_int(a.shape[1]), _d(a), C.c_int(b.shape[1]), _d(b), _d(v)))
        return v

    def kat_gjk_wave(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        n = a.shape[0]; v = np.zeros((n, 3))
        self._check(self.lib.tj_kat_gjk_wave(self._ctx, C.c_int(n), C.c_int(a.shape[1]), _d(a), C.c_int(b.shape[1]), _d(b), _d(v)))
        return v

    def kat_gjk_wave_split(self, a, b, k_stop):
        """gjk_wave interrupted after k_stop iterations and continued from the state it left in memory: (witness vectors, iterations)"""
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        n = a.shape[0]; out = np.zeros((n, 4))
        self._check(self.lib.tj_kat_gjk_wave_split(self._ctx, C.c_int(n), C.c_int(a.shape[1]), _d(a), C.c_int(b.shape[1]), _d(b), C.c_int(k_stop), _d(out)))
        return out[:, :3], out[:, 3].astype(int)

    def kat_planes(self, what, P, Q, dist):
        P = np.ascontiguousarray(P, dtype=np.float64); Q = np.ascontiguousarray(Q, dtype=np.float64)
        n = P.shape[0]; out = np.zeros((n, 5))
        self._check(self.lib.tj_kat_planes(self._ctx, C.c_int(what), C.c_int(n), _d(P), _d(Q), C.c_double(dist), _d(out)))
        return out

    def kat_refine_planes(self, what, P, Q, cd):
        """what 5: Optimal_plane::optimal_cd (Q = points), 6: self_optimal_cd (Q = hulls); cd[n][4] = planes to refine.
        Returns (finished[n], refined cd[n][4])."""
        P = np.ascontiguousarray(P, dtype=np.float64); Q = np.ascontiguousarray(Q, dtype=np.float64)
        n = P.shape[0]; out = np.zeros((n, 5)); out[:, 1:] = cd
        self._check(self.lib.tj_kat_planes(self._ctx, C.c_int(what), C.c_int(n), _d(P), _d(Q), C.c_double(0.0), _d(out)))
        return out[:, 0] != 0, out[:, 1:].copy()

    # ---- "optimal_plane":1 : the persistent plane tables ----
    def get_obs_cache(self, u=0, cap=4096):
        """single UAV: per segment (ids into the scene's cloud, planes (c, d)) in insertion order"""
        out = []
        for tr in range(self.S):
            ids = np.zeros(cap, dtype=np.int32); cd = np.zeros((cap, 4))
            n = self.lib.tj_get_obs_cache(self._ctx, C.c_int(u), C.c_int(tr), C.c_int(cap), _i(ids), _d(cd))
            self._check(n)
            assert n <= cap
            out.append((ids[:n].copy(), cd[:n].copy()))
        return out

    def set_obs_cache(self, cache, u=0):
        for tr, (ids, cd) in enumerate(cache):
            ids = np.ascontiguousarray(ids, dtype=np.int32); cd = np.ascontiguousarray(cd, dtype=np.float64).reshape(-1, 4)
            self._check(self.lib.tj_set_obs_cache(self._ctx, C.c_int(u), C.c_int(tr), C.c_int(len(ids)), _i(ids), _d(cd)))

    def get_pair_cache(self):
        """multi UAV: flags [S][U][U] (p0 < p1) and planes [S][U][U][4] = (c, d) before the offset/2 split"""
        fl = np.zeros((self.S, self.U, self.U), dtype=np.int32); cd = np.zeros((self.S, self.U, self.U, 4))
        self._check(self.lib.tj_get_pair_cache(self._ctx, _i(fl), _d(cd)))
        return fl, cd

    def set_pair_cache(self, flags, cd):
        fl = np.ascontiguousarray(flags, dtype=np.int32); cd = np.ascontiguousarray(cd, dtype=np.float64)
        self._check(self.lib.tj_set_pair_cache(self._ctx, _i(fl), _d(cd)))

    # ---- initial-trajectory planner (replaces ompl_init / simplify_path / edge_collision) ----
    def edge_collision(self, edges, prior=None, d=None):
        """the reference's motion validator on a batch of straight edges [n][2][3] -> bool[n]"""
        edges = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 6)
        prior = np.zeros((0, 6)) if prior is None else np.ascontiguousarray(prior, dtype=np.float64).reshape(-1, 6)
        d = self.params["offset"] + 0.5 * self.params["margin"] if d is None else d
        hit = np.zeros(max(len(edges), 1), dtype=np.int32)
        self._check(self.lib.tj_edge_collision(self._ctx, C.c_int(len(edges)), _d(edges), C.c_int(len(prior)), _d(prior), C.c_double(d), _i(hit)))
        return hit[:len(edges)].astype(bool)

    def plan_init(self, starts, goals, nodes=0, min_waypoints=0, bound_scale=0.0, cap=256):
        """way points [n_robots][n][3] from start/goal pairs (tj_plan_init); n is common to all robots"""
        starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3); goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
        U = len(starts); wp = np.zeros((U, cap, 3)); n = C.c_int()
        self._check(self.lib.tj_plan_init(self._ctx, C.c_int(U), _d(starts), _d(goals), C.c_double(bound_scale), C.c_int(nodes), C.c_int(min_waypoints), C.c_int(cap), _d(wp), C.byref(n)))
        return wp[:, :n.value].copy()

    def kat_ccd(self, P, D, Q, E, q, tu, d):
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (P, D, Q, E, q, tu)]
        n = arrs[0].shape[0]; out = np.zeros((n, 2))
        self._check(self.lib.tj_kat_ccd(self._ctx, C.c_int(n), *[_d(x) for x in arrs], C.c_double(d), _d(out)))
        return out

    def kat_query(self, boxes, margin, cap=2048, sort=True):
        """raw broad-phase candidate SETS of caller-supplied query boxes [nq][6] (lo, hi): list of sorted id arrays"""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        nq = boxes.shape[0]; counts = np.zeros(nq, dtype=np.int32); ids = np.zeros((nq, cap), dtype=np.int32)
        self._check(self.lib.tj_kat_query(self._ctx, C.c_int(nq), _d(boxes), C.c_double(margin), C.c_int(cap), _i(counts), _i(ids)))
        return [np.sort(ids[q, :counts[q]]) if sort else ids[q, :counts[q]].copy() for q in range(nq)]

    def kat_tri(self, P, D, tri, t, dist, off):
        arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (P, D, tri, t)]
        n = arrs[0].shape[0]; out = np.zeros((n, 8))
        self._check(self.lib.tj_kat_tri(self._ctx, C.c_int(n), *[_d(x) for x in arrs], C.c_double(dist), C.c_double(off), _d(out)))
        return out

    def kat_linalg(self, mats):
        mats = np.ascontiguousarray(mats, dtype=np.float64)
        out = np.zeros((mats.shape[0], 2))
        self._check(self.lib.tj_kat_linalg(self._ctx, C.c_int(mats.shape[0]), C.c_int(mats.shape[1]), _d(mats), _d(out)))
        return out

    def energy(self):
        """Energy_admm::spline_energy of every owned robot at the current state (planes of the last iteration)"""
        e = np.zeros(self.U)
        self._check(self.lib.tj_get_energy(self._ctx, _d(e)))
        return e

    def build_info(self):
        ms, dev = C.c_double(), C.c_int()
        self._check(self.lib.tj_get_build_info(self._ctx, C.byref(ms), C.byref(dev)))
        return dict(bvh_build_ms=ms.value, on_device=bool(dev.value))

    def stats(self):
        s = TjStats()
        self._check(self.lib.tj_get_stats(self._ctx, C.byref(s)))
        return {n: getattr(s, n) for n, _ in TjStats._fields_}

    def exchange_buffer(self, what):
        ptr, per, first, n = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.tj_exchange_buffer(self._ctx, what, C.byref(ptr), C.byref(per), C.byref(first), C.byref(n)))
        return ptr.value, per.value, first.value, n.value


class Group:
    """The same problem sharded over several devices by the library itself (`tj_group`, csrc/tj_group.h): one context per rank,
    peer stores + events between them, no torch on the path.  `devices` may repeat (several ranks on one GPU)."""

    def __init__(self, scene, devices, params=None, stop=None, **caps):
        self.lib = load_library()
        p = dict(scenes.DEFAULT_PARAMS)
        if params:
            p.update(params)
        self.mode, self.U, self.P = scene["mode"], scene["U"], scene["P"]
        self.res = p["res"]
        self.S, self.T = self.P * self.res, 3 * self.P + 3
        tp = TjParams()
        self.lib.tj_default_params(C.byref(tp), self.mode, self.U, self.P)
        tp.res = self.res
        tp.lambda_, tp.margin, tp.offset, tp.mu = p["lam"], p["margin"], p["offset"], p["mu"]
        tp.vel_limit, tp.acc_limit, tp.ks, tp.kt = p["vel_limit"], p["acc_limit"], scene["ks"], p["kt"]
        tp.stop = p["stop"] if stop is None else stop
        for k, v in caps.items():
            setattr(tp, k, v)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self.n = len(dev)
        self._g = C.c_void_p()
        rc = self.lib.tj_group_create(C.byref(tp), C.c_int(self.n), _i(dev), C.byref(self._g))
        if rc < 0:
            raise TrajAdmmError(f"tj_group_create error {rc}: {self.lib.tj_group_last_error(None).decode()}")
        if scene.get("tris") is not None:
            verts = np.ascontiguousarray(scene["tris"], dtype=np.float64).reshape(-1, 3)
            n = verts.shape[0] // 3
            faces = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
            self._check(self.lib.tj_group_set_mesh(self._g, _d(verts), C.c_int(3 * n), _i(faces), C.c_int(n)))
        else:
            cloud = np.ascontiguousarray(scene["cloud"], dtype=np.float64).reshape(-1, 3)
            self._check(self.lib.tj_group_set_cloud(self._g, _d(cloud), C.c_int(cloud.shape[0])))
        self._wp = np.ascontiguousarray(scene["waypoints"], dtype=np.float64)
        self._pt0 = float(p["piece_time0"])
        self.reset()

    def reset(self):
        self._check(self.lib.tj_group_init_state(self._g, _d(self._wp), C.c_double(self._pt0)))

    def _check(self, rc):
        if rc < 0:
            raise TrajAdmmError(f"libtrajadmm group error {rc}: {self.lib.tj_group_last_error(self._g).decode()}")
        return rc

    def iterate(self, n=1):
        """n iterations on every rank; returns (gnorm, iterations so far, converged)"""
        g, it, cv = C.c_double(), C.c_int(), C.c_int()
        self._check(self.lib.tj_group_iterate(self._g, C.c_int(n), C.byref(g), C.byref(it), C.byref(cv)))
        return g.value, it.value, bool(cv.value)

    @property
    def transport(self):
        self.lib.tj_group_transport.restype = C.c_char_p
        return self.lib.tj_group_transport(self._g).decode()

    def set_transport(self, name):
        """"flag" | "event" | "rccl" (csrc/tj_group.h); between batches only"""
        self._check(self.lib.tj_group_set_transport(self._g, name.encode()))

    @property
    def rccl_ranks(self):
        """ranks RCCL's communicator reports for this group (0 unless the rccl transport is selected)"""
        return max(0, int(self.lib.tj_group_rccl_ranks(self._g)))

    def launch_counts(self):
        """kernels each rank's context has enqueued so far (tj_launch_count)"""
        self.lib.tj_launch_count.restype = C.c_longlong
        self.lib.tj_group_ctx.restype = C.c_void_p
        return [int(self.lib.tj_launch_count(C.c_void_p(self.lib.tj_group_ctx(self._g, C.c_int(r))))) for r in range(self.n)]

    def profile_exchange(self, reps=50):
        """event-timed microseconds of one exchange of each buffer kind (slowest rank's average)"""
        us = np.zeros(5)
        self._check(self.lib.tj_group_profile_exchange(self._g, C.c_int(reps), _d(us)))
        return us

    def get_state(self):
        """every robot's state from the rank that owns it"""
        U, P, T = self.U, self.P, self.T
        st = dict(spline=np.zeros((U, 3, T)), p_slack=np.zeros((U, 3, 6 * P)), p_lambda=np.zeros((U, 3, 6 * P)),
                  t_slack=np.zeros((U, P)), t_lambda=np.zeros((U, P)), piece_time=np.zeros(U))
        for u in range(U):
            pt = C.c_double()
            self._check(self.lib.tj_group_get_state(self._g, u, _d(st["spline"][u]), _d(st["p_slack"][u]), _d(st["p_lambda"][u]),
                                                    _d(st["t_slack"][u]), _d(st["t_lambda"][u]), C.byref(pt)))
            st["piece_time"][u] = pt.value
        return st

    def close(self):
        if getattr(self, "_g", None) and self._g.value:
            self.lib.tj_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
