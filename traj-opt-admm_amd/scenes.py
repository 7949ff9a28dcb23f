"""Seeded synthetic scenes.  The reference ships no data files
(`bridge.obj`, `cross.obj`, `init/*.txt` are an external download, reference
README.md:24-32), so every scene here is generated, in *solver units* (i.e. after the
x5 scaling that Main/multiPathPlanning3D.cpp:107,536 applies to multi-UAV inputs).

Each scene is a dict:
  mode       0 = single-UAV path (admmPathPlanning3D), 1 = multi-UAV decoupled
  U, P       robots, pieces (= waypoints - 1)
  waypoints  float64 [U, P+1, 3]
  cloud      float64 [N, 3]   obstacle point cloud
  ks         1e-8 (single, admmPathPlanning3D.cpp:477) / 1e-3 (multi, multiPathPlanning3D.cpp:596)
"""
import numpy as np


def scn_a(n_points=32768, seed=12345, pieces=5):
    """SCN-A (BASELINE configs 1-2): one UAV through a tube in a uniform cloud."""
    rng = np.random.default_rng(seed)
    pts = np.empty((0, 3))
    while pts.shape[0] < n_points:
        c = rng.uniform(-3.0, 3.0, size=(2 * n_points, 3))
        keep = (c[:, 1] - 0.8 * np.sin(c[:, 0])) ** 2 + c[:, 2] ** 2 > 0.45 ** 2
        pts = np.concatenate([pts, c[keep]], axis=0)
    cloud = np.ascontiguousarray(pts[:n_points])
    xs = np.linspace(2.7, -2.7, pieces + 1)
    wp = np.stack([xs, 0.8 * np.sin(xs), np.zeros_like(xs)], axis=1)[None]
    return dict(name="SCN-A", mode=0, U=1, P=pieces, waypoints=np.ascontiguousarray(wp), cloud=cloud, ks=1e-8)


def crossing(U, n_points, seed=777, pieces=5, name=None, dz=0.25):
    """SCN-B/C/D family: U robots on a circle of radius 10 fly to the antipodal point,
    stacked dz (0.25) apart in z; cloud = two slabs just below / above the robot layer."""
    rng = np.random.default_rng(seed)
    wp = np.zeros((U, pieces + 1, 3))
    for u in range(U):
        th = np.pi * u / U
        a = np.array([10 * np.cos(th), 10 * np.sin(th), dz * u])
        b = np.array([-10 * np.cos(th), -10 * np.sin(th), dz * u])
        for k in range(pieces + 1):
            wp[u, k] = a + (b - a) * (k / pieces)
    xy = rng.uniform(-12.0, 12.0, size=(n_points, 2))
    r = rng.uniform(0.0, 1.0, size=n_points)
    top = dz * (U - 1) + 0.16 + 0.3 * r
    bot = -0.16 - 0.3 * r
    z = np.where(np.arange(n_points) % 2 == 0, bot, top)
    cloud = np.ascontiguousarray(np.concatenate([xy, z[:, None]], axis=1))
    return dict(name=name or f"crossing-U{U}-N{n_points}", mode=1, U=U, P=pieces,
                waypoints=wp, cloud=cloud, ks=1e-3)


def scn_b():
    return crossing(8, 20000, seed=777, name="SCN-B")


def scn_c():
    return crossing(64, 100000, seed=777, name="SCN-C")


def scn_c3():
    """SCN-C3: SCN-C's fleet (64 UAVs, 100 000 points) stacked 0.29 instead of 0.25 apart.  On SCN-C itself the unmodified reference
    moves its own final control points by 1.2e-2 under a ONE-ULP change of its input; here by ~2e-10 (tests/golden/envelope_scn_c3.npz; at exactly 0.3 -- the barrier range -- the reference backs off ~400 times in iteration 0, beyond this library's loop cap),
    so north_star's "final trajectory within 1e-8 of the reference" can be tested literally at 64 UAVs."""
    return crossing(64, 100000, seed=777, name="SCN-C3", dz=0.29)


def scn_d():
    return crossing(256, 1000000, seed=777, name="SCN-D")


def scn_e():
    """SCN-E (stress, not a BASELINE config): 64 UAVs crossing THROUGH a cloud of 1M points that fills the flight volume
    (kept 0.22 away from the straight initial paths): ~10^2 broad-phase candidates per segment, the BVH / k-DOP / GJK
    front end becomes throughput bound."""
    s = hard(U=64, n_points=1_000_000, seed=9, radius=10.0, dz=0.25, clear=0.22)
    s["name"] = "SCN-E"
    return s


def triangulate(scene, size=0.06, seed=11, degenerate=False):
    """The same scene with TRIANGLE obstacles (BASELINE config 5's geometry type; Solver -> tj_set_mesh): every cloud point
    becomes a small randomly oriented triangle that contains it (vertices within `size` of the point), tris [N][3][3].
    degenerate=True: three EQUAL vertices -- such a scene must reproduce the point-cloud results bit for bit."""
    pts = scene["cloud"]
    n = pts.shape[0]
    if degenerate:
        tris = np.repeat(pts[:, None, :], 3, axis=1)
    else:
        rng = np.random.default_rng(seed)
        off = rng.normal(0.0, 1.0, size=(n, 3, 3))
        off -= off.mean(axis=1, keepdims=True)                       # centroid stays at the cloud point
        off *= size / np.maximum(1e-12, np.linalg.norm(off, axis=2).max(axis=1))[:, None, None]
        tris = pts[:, None, :] + off
    out = dict(scene)
    out["tris"] = np.ascontiguousarray(tris)
    out["name"] = scene["name"] + ("-tri0" if degenerate else "-tri")
    return out


def scn_d_tri():
    """BASELINE config 5 as stated: 256 UAVs, 1M obstacle TRIANGLES"""
    s = triangulate(scn_d(), size=0.05)
    s["name"] = "SCN-D-tri"
    return s


def tiny(mode=1, U=3, n_points=600, seed=5):
    """Small scene for fast oracle-vs-HIP unit tests."""
    if mode == 0:
        s = scn_a(n_points=n_points, seed=seed)
        s["name"] = "tiny-single"
        return s
    s = crossing(U, n_points, seed=seed, name="tiny-multi")
    return s


# Shipped Config_File/3D.json values (reference "Config File/3D.json")
DEFAULT_PARAMS = dict(res=8, vel_limit=2.0, acc_limit=2.0, lam=10.0, margin=0.1, offset=0.1,
                      stop=1e-2, mu=0.1, kt=1.0, piece_time0=20.0)


def write_reference_files(scene, root, mesh_name):
    """Emit the scene in the reference's on-disk formats (OBJ `v` lines, init file), so the
    CLIs can be exercised: model/{single,multiple}/<mesh>, init/<mesh>_init_file.txt.
    Multi-UAV inputs are divided by 5 because the reader multiplies by 5."""
    import os
    multi = scene["mode"] == 1
    sub = "multiple" if multi else "single"
    os.makedirs(os.path.join(root, "model", sub), exist_ok=True)
    os.makedirs(os.path.join(root, "init"), exist_ok=True)
    os.makedirs(os.path.join(root, "result"), exist_ok=True)
    scale = 0.2 if multi else 1.0
    with open(os.path.join(root, "model", sub, mesh_name), "w") as f:
        if scene.get("tris") is not None:   # triangle obstacles: unshared vertices + `f` lines (the CLIs' --triangles front end)
            for p in scene["tris"].reshape(-1, 3) * scale:
                f.write("v %.17g %.17g %.17g\n" % (p[0], p[1], p[2]))
            for i in range(scene["tris"].shape[0]):
                f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
        else:
            for p in scene["cloud"] * scale:
                f.write("v %.17g %.17g %.17g\n" % (p[0], p[1], p[2]))
    with open(os.path.join(root, "init", mesh_name + "_init_file.txt"), "w") as f:
        wp = scene["waypoints"] * scale
        for k in range(wp.shape[1]):
            f.write(" ".join("%.17g" % v for u in range(wp.shape[0]) for v in wp[u, k]) + "\n")


def hard(U=4, n_points=4000, seed=3, pieces=5, radius=4.0, dz=0.13, clear=0.13):
    """Stress scene for tests: robots cross almost in one plane (offset < dz < offset+2*margin, so the start is feasible but tight) and the
    cloud fills the volume around the straight-line initial paths (kept `clear` away from them),
    so obstacle planes, inter-robot planes and both CCD step clamps become active."""
    rng = np.random.default_rng(seed)
    wp = np.zeros((U, pieces + 1, 3))
    for u in range(U):
        th = np.pi * u / U + 0.1
        a = np.array([radius * np.cos(th), radius * np.sin(th), dz * u])
        b = -a.copy(); b[2] = dz * u
        for k in range(pieces + 1):
            wp[u, k] = a + (b - a) * (k / pieces)
    pts = np.empty((0, 3))
    while pts.shape[0] < n_points:
        c = rng.uniform([-radius, -radius, -0.6], [radius, radius, 0.6 + dz * U], size=(4 * n_points, 3))
        keep = np.ones(len(c), dtype=bool)
        for u in range(U):
            a, b = wp[u, 0], wp[u, -1]
            ab = b - a
            t = np.clip(((c - a) @ ab) / (ab @ ab), 0, 1)
            dist = np.linalg.norm(c - (a + t[:, None] * ab), axis=1)
            keep &= dist > clear
        pts = np.concatenate([pts, c[keep]], axis=0)
    return dict(name=f"hard-U{U}", mode=1, U=U, P=pieces, waypoints=wp, cloud=np.ascontiguousarray(pts[:n_points]), ks=1e-3)
