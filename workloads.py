"""Seeded synthetic workloads for the batched Kalman log-likelihood path.

Bench / test infrastructure (pure numpy; also runs under the python3.9 that
generates the golden vectors).  Generators follow SURVEY.md section 8(d); the
matrix STRUCTURE of each family restates what the reference's model classes
build (cited per function) -- only the structure, nothing is imported from or
copied out of the reference.

Every generator returns a dict with reference layouts plus a leading draw axis B:
  y (n,p)   a0 (B,m)   P0 (B,m,m)   T (B,m,m)   Z (B,p,m)   R (B,m,r)
  H (B,p,p)   Q (B,r,r)   and ints B,n,p,m,r, str name.
"""
from __future__ import annotations

import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NILE_CSV = os.path.join(_HERE, "tests", "golden", "nile.csv")

SEED_SERIES = 1337  # seed of the daily-seasonality notebook's first cell
SEED_DRAWS = 2024


def nile():
    """The Nile series (reference data/nile.csv, 100 annual flows)."""
    return np.loadtxt(NILE_CSV, skiprows=1, dtype=np.float64)


def _pack(name, y, a0, P0, T, Z, R, H, Q):
    B, m = a0.shape
    return dict(name=name, y=np.ascontiguousarray(y), a0=a0, P0=P0, T=T, Z=Z, R=R, H=H, Q=Q,
                B=B, n=y.shape[0], p=Z.shape[1], m=m, r=R.shape[2])


def simulate_series(T, Z, R, H, Q, n, rng, burn=200):
    """One series y (n,p) from x' = T x + R eta, y = Z x + eps (single parameter set)."""
    m, r, p = T.shape[0], R.shape[1], Z.shape[0]
    LQ = np.linalg.cholesky(Q + 1e-12 * np.eye(r))
    Hs = np.sqrt(np.clip(np.diag(H), 0.0, None))
    x = np.zeros(m)
    y = np.empty((n, p))
    for t in range(-burn, n):
        x = T @ x + R @ (LQ @ rng.standard_normal(r))
        if t >= 0:
            y[t] = Z @ x + Hs * rng.standard_normal(p)
    return y


def standardize(y):
    return (y - np.nanmean(y, axis=0)) / np.nanstd(y, axis=0)


def stationary_cov(T, RQR):
    """Solve P = T P T' + RQR' per draw: vec P = (I - T(x)T)^-1 vec RQR'.

    Host-side stand-in for the reference's solve_discrete_lyapunov call
    (models/SARIMAX.py:100-107)."""
    B, m, _ = T.shape
    kron = np.einsum("bij,bkl->bikjl", T, T).reshape(B, m * m, m * m)
    A = np.eye(m * m)[None] - kron
    P = np.linalg.solve(A, RQR.reshape(B, m * m, 1)).reshape(B, m, m)
    return 0.5 * (P + np.swapaxes(P, 1, 2))


# --------------------------------------------------------------------------- #
# C1: Nile local level (p = m = r = 1)
# --------------------------------------------------------------------------- #
def nile_local_level(B=1, seed=SEED_DRAWS + 1):
    """BASELINE config 1.  Draw 0 is the known-answer point of SURVEY 8(c):
    H=15099, Q=1469.1, a0=0, P0=1e7 (logp = -641.5855784594156)."""
    rng = np.random.default_rng(seed)
    y = nile()[:, None]
    one = np.ones((B, 1, 1))
    H = 15099.0 * one * np.exp(0.3 * rng.standard_normal((B, 1, 1)))
    Q = 1469.1 * one * np.exp(0.3 * rng.standard_normal((B, 1, 1)))
    H[0], Q[0] = 15099.0, 1469.1
    return _pack("nile_local_level", y, np.zeros((B, 1)), 1e7 * one, one.copy(), one.copy(),
                 one.copy(), H, Q)


def nile_local_linear_trend(B=1, seed=SEED_DRAWS + 9):
    """The reference's only numeric parity case (tests/utilities/test_helpers.py:110-121):
    p=1, m=r=2, a0=0, P0=1e6 I, Q=diag(.5,.01), H=.8 at draw 0; jittered after."""
    rng = np.random.default_rng(seed)
    y = nile()[:, None]
    T = np.broadcast_to(np.array([[1.0, 1.0], [0.0, 1.0]]), (B, 2, 2)).copy()
    Z = np.broadcast_to(np.array([[1.0, 0.0]]), (B, 1, 2)).copy()
    R = np.broadcast_to(np.eye(2), (B, 2, 2)).copy()
    Q = np.zeros((B, 2, 2))
    Q[:, 0, 0] = 0.5 * np.exp(0.3 * rng.standard_normal(B))
    Q[:, 1, 1] = 0.01 * np.exp(0.3 * rng.standard_normal(B))
    H = 0.8 * np.exp(0.3 * rng.standard_normal((B, 1, 1)))
    Q[0] = np.diag([0.5, 0.01])
    H[0] = 0.8
    P0 = np.broadcast_to(1e6 * np.eye(2), (B, 2, 2)).copy()
    return _pack("nile_llt", y, np.zeros((B, 2)), P0, T, Z, R, H, Q)


# --------------------------------------------------------------------------- #
# C2: ARMA(1,1), m = 2  (structure: reference models/SARIMAX.py:30-49)
# --------------------------------------------------------------------------- #
def arma11(B=4096, n=1000, seed=SEED_DRAWS + 2):
    rng_s = np.random.default_rng(SEED_SERIES)
    rho0, th0, s0 = 0.8, 0.3, 1.0
    T0 = np.array([[rho0, 1.0], [0.0, 0.0]])
    R0 = np.array([[1.0], [th0]])
    Z0 = np.array([[1.0, 0.0]])
    y = standardize(simulate_series(T0, Z0, R0, np.zeros((1, 1)), np.array([[s0 ** 2]]), n, rng_s))

    rng = np.random.default_rng(seed)
    rho = rng.uniform(-0.95, 0.95, B)
    theta = np.clip(0.5 * rng.standard_normal(B), -0.95, 0.95)
    sigma = np.abs(rng.standard_normal(B)) + 0.05
    T = np.zeros((B, 2, 2))
    T[:, 0, 0] = rho
    T[:, 0, 1] = 1.0
    R = np.zeros((B, 2, 1))
    R[:, 0, 0] = 1.0
    R[:, 1, 0] = theta
    Z = np.zeros((B, 1, 2))
    Z[:, 0, 0] = 1.0
    Q = (sigma ** 2)[:, None, None].copy()
    H = np.zeros((B, 1, 1))
    RQR = R @ Q @ np.swapaxes(R, 1, 2)
    P0 = stationary_cov(T, RQR)
    return _pack("arma11", y, np.zeros((B, 2)), P0, T, Z, R, H, Q)


# --------------------------------------------------------------------------- #
# C3 / north-star headline: structural model, m = 10, r = 4, p = 1
#   local linear trend (2) + dummy seasonal s=7 (6) + AR(2) cycle (2)
# --------------------------------------------------------------------------- #
def _structural_matrices(phi1, phi2, s_lvl, s_trd, s_sea, s_cyc, s_eps):
    B = phi1.shape[0]
    m, r = 10, 4
    T = np.zeros((B, m, m))
    T[:, 0, 0] = 1.0
    T[:, 0, 1] = 1.0
    T[:, 1, 1] = 1.0
    T[:, 2, 2:8] = -1.0  # dummy seasonal: first row -1s, then shifted identity
    for i in range(5):
        T[:, 3 + i, 2 + i] = 1.0
    T[:, 8, 8] = phi1
    T[:, 8, 9] = phi2
    T[:, 9, 8] = 1.0
    R = np.zeros((B, m, r))
    R[:, 0, 0] = 1.0
    R[:, 1, 1] = 1.0
    R[:, 2, 2] = 1.0
    R[:, 8, 3] = 1.0
    Z = np.zeros((B, 1, m))
    Z[:, 0, [0, 2, 8]] = 1.0
    Q = np.zeros((B, r, r))
    for i, s in enumerate((s_lvl, s_trd, s_sea, s_cyc)):
        Q[:, i, i] = s
    H = s_eps[:, None, None].copy()
    return T, Z, R, H, Q


def structural(B=4096, n=1000, seed=SEED_DRAWS + 3):
    """Headline shape of BASELINE.json north_star (4096 draws x T=1000 x m=10);
    C3 is the same family at B=8192, n=2000."""
    rng_s = np.random.default_rng(SEED_SERIES)
    one = np.ones(1)
    T0, Z0, R0, H0, Q0 = _structural_matrices(1.2 * one, -0.5 * one, 0.02 * one, 0.0005 * one,
                                              0.05 * one, 0.3 * one, 0.2 * one)
    y = standardize(simulate_series(T0[0], Z0[0], R0[0], H0[0], Q0[0], n, rng_s))

    rng = np.random.default_rng(seed)
    # stationary AR(2) via partial autocorrelations
    r1 = rng.uniform(-0.9, 0.9, B)
    r2 = rng.uniform(-0.9, 0.9, B)
    phi2 = r2
    phi1 = r1 * (1.0 - r2)
    hn = lambda: 0.5 * np.abs(rng.standard_normal(B)) + 0.01  # noqa: E731
    T, Z, R, H, Q = _structural_matrices(phi1, phi2, hn(), hn(), hn(), hn(), hn())
    P0 = np.broadcast_to(np.eye(10), (B, 10, 10)).copy()
    return _pack("structural_m10", y, np.zeros((B, 10)), P0, T, Z, R, H, Q)


# --------------------------------------------------------------------------- #
# C4: VARMAX companion form, 5 series (structure: reference models/VARMAX.py:26-80)
# --------------------------------------------------------------------------- #
def varmax(B=16384, n=1000, k=5, order=(2, 1), seed=SEED_DRAWS + 4):
    pa, q = order
    k_order = max(pa, 1) + q
    m = k * k_order
    rng = np.random.default_rng(seed)

    def draw(nb, rng):
        T = np.zeros((nb, m, m))
        ar = 0.5 / max(pa, 1) * rng.standard_normal((nb, k, k * pa))
        T[:, :k, : k * pa] = ar
        if pa > 1:
            T[:, k: k * pa, : k * (pa - 1)] = np.eye(k * (pa - 1))
        # rescale the AR companion block to spectral radius <= 0.9
        comp = T[:, : k * pa, : k * pa]
        rad = np.abs(np.linalg.eigvals(comp)).max(axis=-1)
        scale = np.minimum(1.0, 0.9 / rad)
        for lag in range(pa):
            T[:, :k, lag * k: (lag + 1) * k] *= (scale ** (lag + 1))[:, None, None]
        if q > 0:
            T[:, :k, k * max(1, pa):] = 0.3 * rng.standard_normal((nb, k, k * q))
        if q > 1:
            T[:, -k * (q - 1):, -k * q: -k] = np.eye(k * (q - 1))
        R = np.zeros((nb, m, k))
        R[:, :k, :] = np.eye(k)
        if q > 0:
            end = -k * (q - 1) if q > 1 else None
            R[:, k * -q: end, :] = np.eye(k)
        Z = np.zeros((nb, k, m))
        Z[:, :, :k] = np.eye(k)
        L = np.tril(0.3 * rng.standard_normal((nb, k, k)), -1)
        L[:, np.arange(k), np.arange(k)] = 0.3 + np.abs(rng.standard_normal((nb, k)))
        Q = L @ np.swapaxes(L, 1, 2)
        H = np.zeros((nb, k, k))
        H[:, np.arange(k), np.arange(k)] = 0.1 * np.abs(rng.standard_normal((nb, k))) + 1e-3
        return T, Z, R, H, Q

    rng_s = np.random.default_rng(SEED_SERIES)
    T0, Z0, R0, H0, Q0 = draw(1, rng_s)
    y = standardize(simulate_series(T0[0], Z0[0], R0[0], H0[0], Q0[0], n, rng_s))
    T, Z, R, H, Q = draw(B, rng)
    P0 = np.broadcast_to(np.eye(m), (B, m, m)).copy()
    return _pack(f"varmax_k{k}_{pa}{q}", y, np.zeros((B, m)), P0, T, Z, R, H, Q)


# --------------------------------------------------------------------------- #
# C5: daily seasonality, m = 367 = local linear trend (2) + 365-state dummy seasonal
# --------------------------------------------------------------------------- #
def daily_seasonal(B=1024, n=730, s=366, seed=SEED_DRAWS + 5):
    m, r = 2 + (s - 1), 3
    rng_s = np.random.default_rng(SEED_SERIES)

    def mats(nb, s_lvl, s_trd, s_sea, s_eps):
        T = np.zeros((nb, m, m))
        T[:, 0, 0] = 1.0
        T[:, 0, 1] = 1.0
        T[:, 1, 1] = 1.0
        T[:, 2, 2:] = -1.0
        idx = np.arange(s - 2)
        T[:, 3 + idx, 2 + idx] = 1.0
        R = np.zeros((nb, m, r))
        R[:, 0, 0] = R[:, 1, 1] = R[:, 2, 2] = 1.0
        Z = np.zeros((nb, 1, m))
        Z[:, 0, 0] = Z[:, 0, 2] = 1.0
        Q = np.zeros((nb, r, r))
        Q[:, 0, 0], Q[:, 1, 1], Q[:, 2, 2] = s_lvl, s_trd, s_sea
        H = np.asarray(s_eps, dtype=np.float64).reshape(nb, 1, 1).copy()
        return T, Z, R, H, Q

    T0, Z0, R0, H0, Q0 = mats(1, 0.02, 0.0005, 0.01, 0.2)
    y = standardize(simulate_series(T0[0], Z0[0], R0[0], H0[0], Q0[0], n, rng_s, burn=0))
    rng = np.random.default_rng(seed)
    hn = lambda: 0.5 * np.abs(rng.standard_normal(B)) + 0.01  # noqa: E731
    T, Z, R, H, Q = mats(B, hn(), hn(), hn(), hn())
    P0 = np.broadcast_to(np.eye(m), (B, m, m)).copy()
    return _pack(f"daily_seasonal_m{m}", y, np.zeros((B, m)), P0, T, Z, R, H, Q)


# --------------------------------------------------------------------------- #
# generic dense random stable model (tests): any (p, m, r)
# --------------------------------------------------------------------------- #
def random_dense(B, n, p, m, r, seed=0, h_diag=True, h_zero=False):
    rng = np.random.default_rng(seed)

    def draw(nb, rng):
        T = rng.standard_normal((nb, m, m)) / np.sqrt(m)
        rad = np.abs(np.linalg.eigvals(T)).max(axis=-1)
        T *= (np.minimum(1.0, 0.9 / rad))[:, None, None]
        Z = rng.standard_normal((nb, p, m))
        R = rng.standard_normal((nb, m, r))
        L = np.tril(0.3 * rng.standard_normal((nb, r, r)), -1)
        L[:, np.arange(r), np.arange(r)] = 0.3 + np.abs(rng.standard_normal((nb, r)))
        Q = L @ np.swapaxes(L, 1, 2)
        if h_zero:
            H = np.zeros((nb, p, p))
        elif h_diag:
            H = np.zeros((nb, p, p))
            H[:, np.arange(p), np.arange(p)] = 0.2 + np.abs(rng.standard_normal((nb, p)))
        else:
            Lh = np.tril(0.3 * rng.standard_normal((nb, p, p)), -1)
            Lh[:, np.arange(p), np.arange(p)] = 0.4 + np.abs(rng.standard_normal((nb, p)))
            H = Lh @ np.swapaxes(Lh, 1, 2)
        return T, Z, R, H, Q

    rng_s = np.random.default_rng(SEED_SERIES + seed)
    T0, Z0, R0, H0, Q0 = draw(1, rng_s)
    Hsim = H0[0] if not h_zero else 0.1 * np.eye(p)
    y = standardize(simulate_series(T0[0], Z0[0], R0[0], np.diag(np.diag(Hsim)), Q0[0], n, rng_s))
    T, Z, R, H, Q = draw(B, rng)
    a0 = 0.1 * rng.standard_normal((B, m))
    Lp = np.tril(0.2 * rng.standard_normal((B, m, m)), -1)
    Lp[:, np.arange(m), np.arange(m)] = 0.7 + np.abs(rng.standard_normal((B, m)))
    P0 = Lp @ np.swapaxes(Lp, 1, 2)
    return _pack(f"dense_p{p}m{m}r{r}", y, a0, P0, T, Z, R, H, Q)


def reference_test_inputs(p, m, r, n, missing_rows=()):
    """Deterministic inputs of the reference's shape tests
    (tests/utilities/test_helpers.py:62-76), single draw, B axis added."""
    y = np.arange(n * p, dtype=np.float64).reshape(n, p)
    for t in missing_rows:
        y[t] = np.nan
    T = np.eye(m, k=-1)
    T[0, :] = 1.0 / m
    return _pack(
        f"reftest_p{p}m{m}r{r}", y, np.zeros((1, m)), np.eye(m)[None], T[None],
        np.eye(m)[:p, :][None], np.eye(m)[:, :r][None], np.eye(p)[None], np.eye(r)[None],
    )


WORKLOADS = {
    "c1": lambda **kw: nile_local_level(**kw),
    "c2": lambda **kw: arma11(**{**dict(B=4096, n=1000), **kw}),
    "headline": lambda **kw: structural(**{**dict(B=4096, n=1000), **kw}),
    "c3": lambda **kw: structural(**{**dict(B=8192, n=2000), **kw}),
    "c4": lambda **kw: varmax(**{**dict(B=16384, n=1000, order=(2, 1)), **kw}),
    "c4_m10": lambda **kw: varmax(**{**dict(B=16384, n=1000, order=(2, 0)), **kw}),
    "c5": lambda **kw: daily_seasonal(**{**dict(B=1024, n=730), **kw}),
    # C5's shape with a DENSE transition matrix (no model of the reference has one at this size; the line SURVEY 8(d) prices on the fp64
    # MFMA roofline: two dense m x m products per step, no CSR gather, no all-zero blocks to skip)
    "c5_dense": lambda **kw: random_dense(**{**dict(B=1024, n=730, p=1, m=367, r=3, seed=SEED_DRAWS + 6), **kw}),
    # not a BASELINE config: level + trend + 52-week seasonal (m = 53), the shape of the workgroup kernels
    "weekly": lambda **kw: daily_seasonal(**{**dict(B=4096, n=520, s=52), **kw}),
    # level + trend + 24-period seasonal (m = 25): block-tile kernels
    "hourly": lambda **kw: daily_seasonal(**{**dict(B=4096, n=1000, s=24), **kw}),
}
