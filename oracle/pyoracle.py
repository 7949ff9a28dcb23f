"""ctypes front-end for the two CPU checkers -- TEST INFRASTRUCTURE ONLY.

  Engine("port")  -> oracle/liboracle.so  (this repo's own CPU restatement, prefix orc_)
  Engine("ref")   -> oracle/_ref/libref.so (the unmodified reference sources compiled by
                     oracle/Makefile from /root/reference; prefix ref_)

Both libraries export the same stage-level C interface so that tests can run the same
script against either.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product (traj-opt-admm_amd) never does.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATHS = {"port": os.path.join(_HERE, "liboracle.so"), "ref": os.path.join(_HERE, "_ref", "libref.so")}
_PREFIX = {"port": "orc_", "ref": "ref_"}
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def available(kind):
    return os.path.exists(_PATHS[kind])


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class Engine:
    """One solver instance (the reference keeps its state in globals, so only one `ref`
    engine can be live per process; the port mirrors that restriction for symmetry)."""

    def __init__(self, kind, scene, params=None):
        from importlib import import_module
        self.kind = kind
        self.lib = C.CDLL(_PATHS[kind])
        self.px = _PREFIX[kind]
        p = dict(import_module("traj-opt-admm_amd.scenes").DEFAULT_PARAMS)
        if params:
            p.update(params)
        self.params = p
        self.mode, self.U, self.P = scene["mode"], scene["U"], scene["P"]
        self.res = p["res"]
        self.S = self.P * self.res
        self.T = 3 * self.P + 3
        pr = np.array([p["lam"], p["margin"], p["offset"], p["mu"], p["vel_limit"], p["acc_limit"], scene["ks"], p["kt"]], dtype=np.float64)
        if scene.get("tris") is not None:
            # obstacle TRIANGLES [N][3][3] (an extension: the reference's live path reads point clouds only);
            # only the port implements it
            if kind != "port":
                raise ValueError("triangle obstacles exist in the port only: the reference's triangle path is dead code")
            tris = np.ascontiguousarray(scene["tris"], dtype=np.float64).reshape(-1, 9)
            self.N = tris.shape[0]
            self._f("setup_prim", C.c_int)(C.c_int(self.mode), C.c_int(self.U), C.c_int(self.P), C.c_int(self.res), _d(pr), _d(tris), C.c_int(self.N), C.c_int(3))
        else:
            cloud = np.ascontiguousarray(scene["cloud"], dtype=np.float64)
            self.N = cloud.shape[0]
            self._f("setup", C.c_int)(C.c_int(self.mode), C.c_int(self.U), C.c_int(self.P), C.c_int(self.res), _d(pr), _d(cloud), C.c_int(self.N))
        wp = np.ascontiguousarray(scene["waypoints"], dtype=np.float64)
        self._f("init_state", C.c_int)(_d(wp), C.c_double(p["piece_time0"]))
        self.iters = 0

    def _f(self, name, restype=None):
        f = getattr(self.lib, self.px + name)
        f.restype = restype
        return f

    # ---- state -------------------------------------------------------------------
    def get_state(self):
        U, P, T = self.U, self.P, self.T
        st = dict(spline=np.zeros((U, 3, T)), p_slack=np.zeros((U, 3, 6 * P)), p_lambda=np.zeros((U, 3, 6 * P)),
                  t_slack=np.zeros((U, P)), t_lambda=np.zeros((U, P)), piece_time=np.zeros(U))
        f = self._f("get_state")
        for u in range(U):
            pt = C.c_double()
            f(C.c_int(u), _d(st["spline"][u]), _d(st["p_slack"][u]), _d(st["p_lambda"][u]), _d(st["t_slack"][u]), _d(st["t_lambda"][u]), C.byref(pt))
            st["piece_time"][u] = pt.value
        return st  # arrays are column-major T x 3 stored as [3][T]

    def set_state(self, st):
        f = self._f("set_state")
        for u in range(self.U):
            a = [np.ascontiguousarray(st[k][u], dtype=np.float64) for k in ("spline", "p_slack", "p_lambda", "t_slack", "t_lambda")]
            f(C.c_int(u), _d(a[0]), _d(a[1]), _d(a[2]), _d(a[3]), _d(a[4]), C.c_double(float(st["piece_time"][u])))

    def iterate(self):
        self.iters += 1
        return self._f("iterate", C.c_double)()

    def tables(self):
        conv = np.zeros((self.P, 6, 6)); M = np.zeros((6, 6)); basis = np.zeros((self.S, 6, 6))
        self._f("get_tables")(_d(conv), _d(M), _d(basis))
        # column-major 6x6 -> numpy [row, col]
        return conv.transpose(0, 2, 1).copy(), M.T.copy(), basis.transpose(0, 2, 1).copy()

    def kdop_axes(self):
        a = np.zeros((49, 3))
        self._f("get_kdop")(_d(a))
        return a

    # ---- stages ------------------------------------------------------------------
    def stage_planes(self):
        n = self._f("stage_planes", C.c_int)()
        counts = np.zeros(self.U * self.S, dtype=np.int32); planes = np.zeros((max(n, 1), 4))
        self._f("get_planes")(_i(counts), _d(planes))
        return counts.reshape(self.U, self.S), planes[:n]

    def set_planes(self, counts, planes):
        c = np.ascontiguousarray(counts, dtype=np.int32).ravel(); p = np.ascontiguousarray(planes, dtype=np.float64)
        self._f("set_planes")(_i(c), _d(p))

    def stage_direction(self):
        gn = self._f("stage_direction", C.c_double)()
        out = dict(gnorm=gn, direction=np.zeros((self.U, 3, self.T)), t_direction=np.zeros(self.U), wolfe=np.zeros(self.U), gn=np.zeros(self.U))
        f = self._f("get_direction")
        for u in range(self.U):
            a, b, c = C.c_double(), C.c_double(), C.c_double()
            f(C.c_int(u), _d(out["direction"][u]), C.byref(a), C.byref(b), C.byref(c))
            out["t_direction"][u], out["wolfe"][u], out["gn"][u] = a.value, b.value, c.value
        return out

    def set_direction(self, u, direction, t_direction, wolfe, gn):
        d = np.ascontiguousarray(direction, dtype=np.float64)
        self._f("set_direction")(C.c_int(u), _d(d), C.c_double(t_direction), C.c_double(wolfe), C.c_double(gn))

    def local_grad(self, u, sp):
        g = np.zeros(19); h = np.zeros((19, 19))
        self._f("local_grad")(C.c_int(u), C.c_int(sp), _d(g), _d(h))
        return g, h

    def global_grad(self, u):
        n = 3 * self.T + 1
        g = np.zeros(n); h = np.zeros((n, n))
        self._f("global_grad")(C.c_int(u), _d(g), _d(h))
        return g, h

    def stage_steps(self):
        a = np.zeros(self.U); b = np.zeros(self.U)
        self._f("stage_steps")(_d(a), _d(b))
        return a, b

    def stage_linesearch(self):
        a = np.zeros(self.U)
        self._f("stage_linesearch")(_d(a))
        return a

    def stage_slack(self):
        self._f("stage_slack")()

    def stage_update_spline(self):
        """coupled mode (scene mode 2): Optimization3D_multi::update_spline as one stage -> (gnorm, wolfe)"""
        w = C.c_double()
        g = self._f("stage_update_spline", C.c_double)(C.byref(w))
        return g, w.value

    # ---- "optimal_plane":1 -----------------------------------------------------------
    def set_optimal_plane(self, on=True):
        """switch the persistent-plane branch on (empties the caches); call right after construction"""
        self._f("set_optimal_plane")(C.c_int(int(on)))

    def get_obs_cache(self, cap=4096):
        """single-UAV path: per segment (ids ascending, planes (c, d))"""
        out = []
        f = self._f("get_obs_cache", C.c_int)
        for tr in range(self.S):
            ids = np.zeros(cap, dtype=np.int32); cd = np.zeros((cap, 4))
            n = f(C.c_int(tr), C.c_int(cap), _i(ids), _d(cd))
            assert n <= cap
            out.append((ids[:n].copy(), cd[:n].copy()))
        return out

    def set_obs_cache(self, cache):
        f = self._f("set_obs_cache")
        for tr, (ids, cd) in enumerate(cache):
            ids = np.ascontiguousarray(ids, dtype=np.int32); cd = np.ascontiguousarray(cd, dtype=np.float64)
            f(C.c_int(tr), C.c_int(len(ids)), _i(ids), _d(cd))

    def get_pair_cache(self):
        """multi-UAV paths: flags [S][U][U] (p0 < p1), planes [S][U][U][4] = (c, d) before the offset/2 split"""
        fl = np.zeros((self.S, self.U, self.U), dtype=np.int32); cd = np.zeros((self.S, self.U, self.U, 4))
        self._f("get_pair_cache")(_i(fl), _d(cd))
        return fl, cd

    def set_pair_cache(self, flags, cd):
        fl = np.ascontiguousarray(flags, dtype=np.int32); cd = np.ascontiguousarray(cd, dtype=np.float64)
        self._f("set_pair_cache")(_i(fl), _d(cd))

    def edge_collision(self, edges, prior=None, d=None):
        """the planner's motion validator on a batch of straight edges [n][2][3] -> bool[n]"""
        edges = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 6)
        prior = np.zeros((0, 6)) if prior is None else np.ascontiguousarray(prior, dtype=np.float64).reshape(-1, 6)
        d = self.params["offset"] + 0.5 * self.params["margin"] if d is None else d
        hit = np.zeros(len(edges), dtype=np.int32)
        self._f("edge_collision")(C.c_int(len(edges)), _d(edges), C.c_int(len(prior)), _d(prior), C.c_double(d), _i(hit))
        return hit.astype(bool)

    def spline_energy(self, u):
        return self._f("spline_energy", C.c_double)(C.c_int(u))

    def candidates(self, u, use_dir, d, cap=1 << 20):
        counts = np.zeros(self.S, dtype=np.int32); ids = np.zeros(cap, dtype=np.int32)
        n = self._f("candidates", C.c_int)(C.c_int(u), C.c_int(use_dir), C.c_double(d), _i(counts), _i(ids), C.c_int(cap))
        assert n <= cap
        return counts, ids[:n]


class Prims:
    """Stateless known-answer primitives (GJK, k-DOP, planes, pair order, LLT)."""

    def __init__(self, kind):
        self.lib = C.CDLL(_PATHS[kind]); self.px = _PREFIX[kind]
        # both libraries keep parameters (offset, margin, k-DOP axes) in process-wide state that a
        # setup call initialises with the shipped 3D.json values
        Engine(kind, _dummy_scene())

    def _f(self, name, restype=C.c_int):
        f = getattr(self.lib, self.px + name); f.restype = restype; return f

    def gjk(self, p1, p2):
        p1 = np.ascontiguousarray(p1, dtype=np.float64); p2 = np.ascontiguousarray(p2, dtype=np.float64)
        v = np.zeros(3)
        self._f("gjk", None)(C.c_int(p1.shape[0]), _d(p1), C.c_int(p2.shape[0]), _d(p2), _d(v))
        return v

    @staticmethod
    def _cm(P):  # rows x 3 -> column-major buffer
        return np.ascontiguousarray(np.asarray(P, dtype=np.float64).T)

    def plane_obs(self, P, q, dist):
        cd = np.zeros(4); q = np.ascontiguousarray(q, dtype=np.float64)
        ok = self._f("plane_obs")(_d(self._cm(P)), _d(q), C.c_double(dist), _d(cd))
        return bool(ok), cd

    def plane_self(self, P, Q, dist, refine=True):
        cd = np.zeros(4)
        ok = self._f("plane_self")(_d(self._cm(P)), _d(self._cm(Q)), C.c_double(dist), C.c_int(int(refine)), _d(cd))
        return bool(ok), cd

    def query_kat(self, verts, boxes, d):
        """ref only: raw broad-phase candidate sets of query boxes [nq][6] on the reference's own tree over points
        (verts [n][3], BVH::InitPointcloud) or triangles (verts [n][3][3], BVH::InitObstacle): list of sorted id arrays"""
        verts = np.ascontiguousarray(verts, dtype=np.float64)
        prim = 1 if verts.ndim == 2 else 3
        n = verts.shape[0]
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        nq = boxes.shape[0]; counts = np.zeros(nq, dtype=np.int32)
        cap = 1 << 22
        ids = np.zeros(cap, dtype=np.int32)
        tot = self._f("query_kat")(C.c_int(prim), _d(verts), C.c_int(n), C.c_int(nq), _d(boxes), C.c_double(d), _i(counts), _i(ids), C.c_int(cap))
        assert tot <= cap
        out, w = [], 0
        for q in range(nq):
            out.append(np.sort(ids[w:w + counts[q]])); w += counts[q]
        return out

    def sparse_llt_solve(self, H, b):
        """Eigen::SimplicialLLT (AMD ordering) on H.sparseView(), as the single-UAV Newton solve uses it
        (Optimization3D_admm.h:470-475) -> (ok, x, permutationPinv indices)"""
        H = np.asfortranarray(H, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        n = H.shape[0]; x = np.zeros(n); order = np.zeros(n, dtype=np.int32)
        ok = self._f("sparse_llt_solve")(C.c_int(n), _d(H), _d(b), _d(x), _i(order))
        return bool(ok), x, order

    def kdop_general(self, A, B, d):
        """CCD::KDOPDCD on row-major point sets of any size (6 or 12 rows vs 1 or 3 rows)"""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64).reshape(-1, 3)
        return bool(self._f("kdop_general")(C.c_int(A.shape[0]), _d(A), C.c_int(B.shape[0]), _d(B), C.c_double(d)))

    def gjk_dcd_general(self, A, B, d):
        """CCD::GJKDCD on row-major point sets of any size"""
        A = np.ascontiguousarray(A, dtype=np.float64); B = np.ascontiguousarray(B, dtype=np.float64).reshape(-1, 3)
        return bool(self._f("gjk_dcd_general")(C.c_int(A.shape[0]), _d(A), C.c_int(B.shape[0]), _d(B), C.c_double(d)))

    def plane_tri(self, P, tri, dist):
        """port only: Separate::opengjk with a 3-vertex obstacle body (d0 = min over the vertices)"""
        cd = np.zeros(4); tri = np.ascontiguousarray(tri, dtype=np.float64).reshape(9)
        ok = self._f("plane_tri")(_d(self._cm(P)), _d(tri), C.c_double(dist), _d(cd))
        return bool(ok), cd

    def min_eig_small(self, H):
        """eigenvalues()(0) of Eigen::SelfAdjointEigenSolver on a fixed-size 2x2 / 3x3 matrix"""
        H = np.ascontiguousarray(H, dtype=np.float64)
        return self._f("kat_min_eig_small", C.c_double)(C.c_int(H.shape[0]), _d(H))

    def optimal_cd(self, P, q, cd):
        """Optimal_plane::optimal_cd on one (hull, obstacle point) plane -> refined (c, d)"""
        out = np.ascontiguousarray(cd, dtype=np.float64).copy(); q = np.ascontiguousarray(q, dtype=np.float64)
        self._f("kat_optimal_cd", None)(_d(self._cm(P)), _d(q), _d(out))
        return out

    def self_optimal_cd(self, P, Q, cd):
        """Optimal_plane::self_optimal_cd on one (hull, hull) plane -> refined (c, d)"""
        out = np.ascontiguousarray(cd, dtype=np.float64).copy()
        self._f("kat_self_optimal_cd", None)(_d(self._cm(P)), _d(self._cm(Q)), _d(out))
        return out

    def kdop_dcd(self, P, q, d):
        q = np.ascontiguousarray(q, dtype=np.float64)
        return bool(self._f("kdop_dcd")(_d(self._cm(P)), _d(q), C.c_double(d)))

    def kdop_self_dcd(self, P, Q, d):
        return bool(self._f("kdop_self_dcd")(_d(self._cm(P)), _d(self._cm(Q)), C.c_double(d)))

    def kdop_ccd(self, P, D, q, d, t0, t1):
        q = np.ascontiguousarray(q, dtype=np.float64)
        return bool(self._f("kdop_ccd")(_d(self._cm(P)), _d(self._cm(D)), _d(q), C.c_double(d), C.c_double(t0), C.c_double(t1)))

    def gjk_ccd(self, P, D, q, d, t0, t1):
        q = np.ascontiguousarray(q, dtype=np.float64)
        return bool(self._f("gjk_ccd")(_d(self._cm(P)), _d(self._cm(D)), _d(q), C.c_double(d), C.c_double(t0), C.c_double(t1)))

    def self_kdop_ccd(self, P, D, Q, E, d, t1, u1):
        return bool(self._f("self_kdop_ccd")(_d(self._cm(P)), _d(self._cm(D)), _d(self._cm(Q)), _d(self._cm(E)), C.c_double(d), C.c_double(t1), C.c_double(u1)))

    def self_gjk_ccd(self, P, D, Q, E, d, t1, u1):
        return bool(self._f("self_gjk_ccd")(_d(self._cm(P)), _d(self._cm(D)), _d(self._cm(Q)), _d(self._cm(E)), C.c_double(d), C.c_double(t1), C.c_double(u1)))

    def self_pairs(self, lo, hi, d):
        lo = np.ascontiguousarray(lo, dtype=np.float64); hi = np.ascontiguousarray(hi, dtype=np.float64)
        n = lo.shape[0]; cap = n * n
        pr = np.zeros((cap, 2), dtype=np.int32)
        m = self._f("self_pairs")(C.c_int(n), _d(lo), _d(hi), C.c_double(d), _i(pr), C.c_int(cap))
        return pr[:m]

    def llt_fails(self, H):
        H = np.ascontiguousarray(H, dtype=np.float64)
        return bool(self._f("llt_fails")(C.c_int(H.shape[0]), _d(H)))

    def min_eig(self, H):
        H = np.ascontiguousarray(H, dtype=np.float64)
        return self._f("min_eig", C.c_double)(C.c_int(H.shape[0]), _d(H))


def _dummy_scene():
    return dict(mode=0, U=1, P=2, ks=1e-8, cloud=np.array([[9.0, 9.0, 9.0], [8.0, 8.0, 8.0]]),
                waypoints=np.array([[[0.0, 0, 0], [1.0, 0, 0], [2.0, 0, 0]]]))
