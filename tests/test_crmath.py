"""csrc/dev_crmath.h (log / sin / cos that round like glibc's) against 80-digit decimal arithmetic, on the CPU: the header
compiles for the host, and the device executes the same IEEE operations (explicit fma, -ffp-contract=off)."""
import math
import os
import subprocess
from decimal import Decimal, getcontext

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
getcontext().prec = 70


def _dsin(x):
    term = x; s = x; k = 1
    while abs(term) > Decimal(10) ** -65:
        term = -term * x * x / ((2 * k) * (2 * k + 1)); s += term; k += 1
    return s


def _dcos(x):
    term = Decimal(1); s = term; k = 1
    while abs(term) > Decimal(10) ** -65:
        term = -term * x * x / ((2 * k - 1) * (2 * k)); s += term; k += 1
    return s


def _ulps(a, b):
    return np.abs(np.ascontiguousarray(a).view(np.int64) - np.ascontiguousarray(b).view(np.int64))


def test_crmath_is_correctly_rounded_where_glibc_is(tmp_path):
    exe = str(tmp_path / "crmath_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-o", exe, os.path.join(ROOT, "tests", "devtools", "crmath_check.cpp")])
    rng = np.random.default_rng(11)
    n = 6000
    # logarithm arguments: dist / margin in (0, 1) incl. values next to 1 and tiny ones, plus a spread of magnitudes
    xl = np.concatenate([rng.uniform(1e-6, 1.0, n), 1.0 - 10.0 ** rng.uniform(-14, -1, 600), 1.0 + 10.0 ** rng.uniform(-14, -1, 300),
                         10.0 ** rng.uniform(-300, 300, 600), [1.0, 0.5, 2.0, 0.75, 1.5, 1.4999999999999998]])
    xa = np.concatenate([rng.uniform(-1.5, 1.5, n), 10.0 ** rng.uniform(-300, -1, 300), [0.0, 1.0 / 64, 1.5, -1.5, 0.95 * math.pi / 2, -0.95 * math.pi / 2, 1.0 / 128, 3.0 / 128]])
    x = np.concatenate([xl, xa])
    (tmp_path / "in.bin").write_bytes(x.tobytes())
    subprocess.check_call([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    o = np.frombuffer((tmp_path / "out.bin").read_bytes()).reshape(-1, 6)
    want_log = np.array([float(Decimal(float(v)).ln()) for v in xl])
    want_sin = np.array([float(_dsin(Decimal(float(v)))) for v in xa])
    want_cos = np.array([float(_dcos(Decimal(float(v)))) for v in xa])
    nl = len(xl)
    for name, got, glibc, want in (("log", o[:nl, 0], o[:nl, 3], want_log), ("sin", o[nl:, 1], o[nl:, 4], want_sin), ("cos", o[nl:, 2], o[nl:, 5], want_cos)):
        u, ug = _ulps(got, want), _ulps(glibc, want)
        # correctly rounded in all but (at most) a handful of cases, never more than one ulp off; at least as often as glibc itself
        assert u.max() <= 1, (name, int(u.max()))
        assert (u > 0).sum() <= max(3, (ug > 0).sum()), (name, int((u > 0).sum()), int((ug > 0).sum()))
        # and therefore equal to glibc wherever glibc is correctly rounded (all but ~0.1 %)
        assert (_ulps(got, glibc) > 0).mean() < 5e-3, name
