"""Batched parameter-vector -> state-space-matrix assembly.

The reference's model classes write slices of the flat parameter vector ``theta`` into named
matrices one draw at a time (``update(theta)``: reference models/SARIMAX.py:59-107,
models/VARMAX.py:95-150) and, for a stationary start, solve a discrete Lyapunov equation for
P0.  Here the model STRUCTURE is turned once, on the host, into a constant template plus two
index maps; ``kf_assemble`` then scatters a whole batch of draws on the GPU and ``kf_lyapunov``
solves all the P0 at once.  Output layouts are exactly what ``kf_filter_forward`` takes.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np

ORDER = ("a0", "P0", "T", "Z", "R", "H", "Q")


@dataclass
class Layout:
    name: str
    p: int
    m: int
    r: int
    param_names: List[str]
    stationary: bool
    n_theta: int = 0
    template: Dict[str, np.ndarray] = field(default_factory=dict)
    dst: List[int] = field(default_factory=list)
    src: List[int] = field(default_factory=list)

    @property
    def shapes(self) -> Dict[str, Tuple[int, ...]]:
        p, m, r = self.p, self.m, self.r
        return dict(a0=(m,), P0=(m, m), T=(m, m), Z=(p, m), R=(m, r), H=(p, p), Q=(r, r))

    @property
    def offsets(self) -> Dict[str, int]:
        off, out = 0, {}
        for k in ORDER:
            out[k] = off
            off += int(np.prod(self.shapes[k]))
        out["_total"] = off
        return out

    def flat_template(self) -> np.ndarray:
        return np.concatenate([np.asarray(self.template[k], dtype=np.float64).ravel() for k in ORDER])

    def put(self, mat: str, index, n: int):
        """theta[cursor : cursor+n] -> mat[index] (numpy fancy index, C order), advance cursor."""
        flat = np.arange(int(np.prod(self.shapes[mat]))).reshape(self.shapes[mat])[index].ravel()
        assert flat.size == n, (mat, flat.size, n)
        base = self.offsets[mat]
        self.dst += [base + int(i) for i in flat]
        self.src += list(range(self.n_theta, self.n_theta + n))
        self.n_theta += n

    def apply_numpy(self, theta: np.ndarray) -> Dict[str, np.ndarray]:
        """Host evaluation of the same maps (used by the CPU tests of the host logic)."""
        theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        out = np.repeat(self.flat_template()[None], theta.shape[0], axis=0)
        if self.dst:
            out[:, np.asarray(self.dst)] = theta[:, np.asarray(self.src)]
        return self.split(out)

    def split(self, out):
        off, res = self.offsets, {}
        for k in ORDER:
            n = int(np.prod(self.shapes[k]))
            res[k] = out[:, off[k]: off[k] + n].reshape((out.shape[0],) + self.shapes[k])
        return res


def arma_layout(order: Tuple[int, int], stationary_initialization: bool = True) -> Layout:
    """BayesianARMA (reference models/SARIMAX.py:10-107): Harvey form, k_states = max(p, q+1),
    k_posdef = 1; theta = [x0, (P0), sigma_state, rho, theta]."""
    p_ar, q_ma = order
    m = max(p_ar, q_ma + 1)
    names = ["x0", "P0", "sigma_state", "rho", "theta"]
    if stationary_initialization:
        names.remove("P0")
    lay = Layout("arma", 1, m, 1, names, stationary_initialization)
    Z = np.zeros((1, m)); Z[0, 0] = 1.0                       # :30
    T = np.eye(m, k=1)                                        # :32
    R = np.zeros((m, 1)); R[0, 0] = 1.0                       # :34
    lay.template = dict(a0=np.zeros(m), P0=np.eye(m), T=T, Z=Z, R=R, H=np.zeros((1, 1)),
                        Q=np.zeros((1, 1)))
    lay.put("a0", slice(None), m)                             # :70-73
    if not stationary_initialization:
        lay.put("P0", (slice(None), slice(None)), m * m)      # :75-81
    lay.put("Q", (np.array([0]), np.array([0])), 1)           # :83-86
    lay.put("T", (np.arange(p_ar), np.zeros(p_ar, dtype=int)), p_ar)          # :88-91
    lay.put("R", (np.arange(1, q_ma + 1), np.zeros(q_ma, dtype=int)), q_ma)   # :93-96
    return lay


def varmax_layout(k_obs: int, order: Tuple[int, int], stationary_initialization: bool = True,
                  measurement_error: bool = True) -> Layout:
    """BayesianVARMAX (reference models/VARMAX.py:11-150): companion form, k_states =
    k_obs*(max(p,1)+q), k_posdef = k_obs; theta = [x0, (P0), AR, MA, state_cov, (obs_cov)]."""
    p_ar, q_ma = order
    k_order = max(p_ar, 1) + q_ma
    m, r = k_obs * k_order, k_obs
    names = ["x0", "P0", "ar_params", "ma_params", "state_cov", "obs_cov"]
    if stationary_initialization:
        names.remove("P0")
    if not measurement_error:
        names.remove("obs_cov")
    if p_ar == 0:
        names.remove("ar_params")
    if q_ma == 0:
        names.remove("ma_params")
    lay = Layout("varmax", k_obs, m, r, names, stationary_initialization)
    Z = np.zeros((k_obs, m)); Z[np.arange(k_obs), np.arange(k_obs)] = 1.0       # :43
    T = np.zeros((m, m))
    if p_ar > 1:                                                                  # :52-54
        T[k_obs: k_obs * p_ar, 0: k_obs * (p_ar - 1)] = np.eye(k_obs * (p_ar - 1))
    if q_ma > 1:                                                                  # :56-58
        T[-k_obs * (q_ma - 1):, -k_obs * q_ma: -k_obs] = np.eye(k_obs * (q_ma - 1))
    R = np.zeros((m, k_obs)); R[:k_obs, :] = np.eye(k_obs)                        # :62-63
    if q_ma > 0:                                                                  # :64-66
        end = -k_obs * (q_ma - 1) if q_ma > 1 else None
        R[k_obs * -q_ma: end, :] = np.eye(k_obs)
    lay.template = dict(a0=np.zeros(m), P0=np.zeros((m, m)), T=T, Z=Z, R=R,
                        H=np.zeros((k_obs, k_obs)), Q=np.zeros((r, r)))
    lay.put("a0", slice(None), m)                                                 # :108-109
    if not stationary_initialization:
        lay.put("P0", (slice(None), slice(None)), m * m)                          # :111-116
    if p_ar > 0:                                                                  # :119-122
        lay.put("T", (slice(0, k_obs), slice(0, k_obs * p_ar)), k_obs * k_obs * p_ar)
    if q_ma > 0:                                                                  # :125-128
        lay.put("T", (slice(0, k_obs), slice(k_obs * max(1, p_ar), None)), k_obs * k_obs * q_ma)
    lay.put("Q", (slice(None), slice(None)), r * r)                               # :131-135
    if measurement_error:                                                         # :138-143
        lay.put("H", (np.arange(k_obs), np.arange(k_obs)), k_obs)
    return lay


class BatchedStateSpace:
    """theta (B, n_theta) in HBM -> dict(a0, P0, T, Z, R, H, Q) in HBM, ready for the filter."""

    def __init__(self, layout: Layout, engine=None):
        from .engine import get_engine

        self.layout = layout
        self.engine = engine or get_engine()
        self._template = self.engine.to_device(layout.flat_template())
        self._dst = np.asarray(layout.dst, dtype=np.int32)
        self._src = np.asarray(layout.src, dtype=np.int32)

    def __call__(self, theta):
        out = self.engine.assemble(theta, self._template, self._dst, self._src)
        mats = {k: v.contiguous() for k, v in self.layout.split(out).items()}
        if self.layout.stationary:
            P0, info = self.engine.lyapunov(mats["T"], mats["R"], mats["Q"])
            mats["P0"] = P0
            mats["lyapunov_info"] = info
        return mats
