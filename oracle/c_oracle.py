"""ctypes wrapper of oracle/_build/libkf_oracle.so (plain C restatement, see kf_oracle.c).

TEST INFRASTRUCTURE ONLY -- never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libkf_oracle.so")
LIB_NATIVE = os.path.join(_HERE, "_build", "libkf_oracle_native.so")
FILTERS = {"standard": 0, "cholesky": 1, "univariate": 2, "single": 3}
_lib = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def use_native():
    """bench.py's cpu_baseline leg: (re)build with -O3 -march=native ON THIS HOST and load that
    library from now on.  Returns a label of the flavour in use; falls back to the portable build
    when the host has no compiler."""
    global _lib
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
    except Exception:  # noqa: BLE001
        load()
        return "-O2 portable build"
    _lib = None
    load(LIB_NATIVE)
    return "-O3 -march=native, built on this host"


def load(path=None):
    global _lib
    if _lib is None:
        if path is None:
            path = LIB
            if not os.path.exists(LIB):
                build()
        _lib = C.CDLL(path)
        dp = C.c_void_p
        _lib.kf_oracle_filter.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_uint] + [dp] * 17 + [C.c_int]
        _lib.kf_oracle_filter.restype = C.c_int
        _lib.kf_oracle_smoother.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int] + [dp] * 7 + [C.c_int]
        _lib.kf_oracle_smoother.restype = C.c_int
        _lib.kf_oracle_max_threads.restype = C.c_int
    return _lib


def _p(x):
    return None if x is None else x.ctypes.data_as(C.c_void_p)


def _c(x):
    return None if x is None else np.ascontiguousarray(x, dtype=np.float64)


def max_threads():
    return load().kf_oracle_max_threads()


def kalman_filter(y, a0, P0, T, Z, R, H, Q, c=None, d=None, filter_type="standard",
                  mvn_const="reference", full=True, nthreads=0, out=None):
    """Batched inputs as in include/kf.h (leading B). Returns dict like kalman_numpy.
    ``out``: a dict returned by an earlier call with the same shapes, to reuse its buffers."""
    lib = load()
    y, a0, P0, T, Z, R, H, Q, c, d = map(_c, (y, a0, P0, T, Z, R, H, Q, c, d))
    B, m = a0.shape
    p, r = Z.shape[1], R.shape[2]
    per_draw = y.ndim == 3
    n = y.shape[-2]
    flags = (1 if mvn_const == "exact" else 0) | (2 if per_draw else 0)
    if out is not None:
        logp, info, ll = out["logp"], out["info"], out["ll_obs"]
        fa, pa, fP, pP = (out["filtered_states"], out["predicted_states"], out["filtered_covs"],
                          out["predicted_covs"])
        assert logp.shape == (B,) and (not full or fP.shape == (B, n, m, m))
    else:
        logp = np.empty(B)
        info = np.zeros(B, dtype=np.int32)
        ll = np.empty((B, n)) if full else None
        fa = np.empty((B, n, m)) if full else None
        pa = np.empty((B, n + 1, m)) if full else None
        fP = np.empty((B, n, m, m)) if full else None
        pP = np.empty((B, n + 1, m, m)) if full else None
    rc = lib.kf_oracle_filter(B, n, p, m, r, FILTERS[filter_type], flags, _p(y), _p(a0), _p(P0),
                              _p(T), _p(Z), _p(R), _p(H), _p(Q), _p(c), _p(d), _p(logp), _p(ll),
                              _p(fa), _p(pa), _p(fP), _p(pP), _p(info), nthreads)
    if rc != 0:
        raise ValueError(f"kf_oracle_filter rc={rc}")
    return dict(logp=logp, ll_obs=ll, filtered_states=fa, predicted_states=pa, filtered_covs=fP,
                predicted_covs=pP, info=info)


def kalman_smoother(T, R, Q, filtered_states, filtered_covs, nthreads=0):
    lib = load()
    T, R, Q, fa, fP = map(_c, (T, R, Q, filtered_states, filtered_covs))
    B, n, m = fa.shape
    r = R.shape[2]
    sa, sP = np.empty_like(fa), np.empty_like(fP)
    rc = lib.kf_oracle_smoother(B, n, m, r, _p(T), _p(R), _p(Q), _p(fa), _p(fP), _p(sa), _p(sP),
                                nthreads)
    if rc != 0:
        raise ValueError(f"kf_oracle_smoother rc={rc}")
    return sa, sP
