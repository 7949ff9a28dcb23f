import importlib, sys, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("traj-opt-admm_amd")
g = np.load('/root/repo/tests/golden/prims_kat.npz')
s = pkg.Solver(pkg.scenes.tiny(mode=1, U=2, n_points=200), stop=0.0, kat=True)
out = s.kat_linalg(g["llt_mats"])
scale = np.abs(g["llt_mats"]).max(axis=(1, 2))
err = np.abs(out[:, 1] - g["min_eig"]) / np.maximum(1.0, scale)
print("n", len(err), "max", err.max(), "median", np.median(err), "p99", np.quantile(err, 0.99))
# more matrices: random piece-like Hessians with a few negative directions
rng = np.random.default_rng(5)
mats = []
for k in range(512):
    q, _ = np.linalg.qr(rng.standard_normal((19, 19)))
    ev = np.concatenate([-10.0 ** rng.uniform(-6, 1, 2), 10.0 ** rng.uniform(-4, 5, 17)])
    a = (q * ev) @ q.T
    mats.append(0.5 * (a + a.T))
mats = np.array(mats)
out = s.kat_linalg(mats)
want = np.array([np.linalg.eigvalsh(m)[0] for m in mats])
sc = np.abs(mats).max(axis=(1, 2))
e2 = np.abs(out[:, 1] - want) / sc
print("random: max", e2.max(), "median", np.median(e2), "p99", np.quantile(e2, 0.99), "disagree flags", int((out[:, 0] == 2).sum()))
