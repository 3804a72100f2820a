"""Host-side driver of the C-ABI (include/kf.h): torch tensors in HBM in, torch tensors out.

torch is plumbing here -- device memory, the current HIP stream, ``torch.distributed`` -- the
arithmetic is entirely in libkf_hip.so.  All tensors are fp64, contiguous, on the engine's
device, with the batch-of-draws axis leading (layouts: include/kf.h).
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib

FILTER_TYPES = ("standard", "cholesky", "univariate", "single")


class KalmanError(RuntimeError):
    """Argument / HIP error reported by the library (negative return code)."""


@dataclass
class FilterResult:
    logp: torch.Tensor                       # (B,)
    ll_obs: Optional[torch.Tensor]           # (B,n)
    filtered_states: Optional[torch.Tensor]  # (B,n,m)
    predicted_states: Optional[torch.Tensor]  # (B,n+1,m)
    filtered_covs: Optional[torch.Tensor]    # (B,n,m,m)
    predicted_covs: Optional[torch.Tensor]   # (B,n+1,m,m)
    info: torch.Tensor                       # (B,) int32


def smoother_from_filter_applies(p, m, ndims, r=None):
    """Should the smoothed moments come from the filter's PREDICTED tapes (kf_smoother_from_filter) rather than from the
    reference's recursion on the filtered ones (kf_smoother, kalman_smoother.py:56-104)?  The one rule every caller that holds
    a filter's result uses: the fused PyTensor node (op.perform_smoother_fused), ``KalmanEngine.smooth_result``,
    ``KalmanSmoother.__call__(..., filter_result=...)``, ``BaseFilter.filter_and_smooth`` and ``ShardedKalman.filter_and_smooth``.
    p == 1 and every system matrix static (``ndims``: the per-draw ranks of T, Z, H, R, Q -- and of d when given as a sixth
    entry, 1), and m <= 1024: every range has an inverse-free kernel now --
      * m <= 4   one lane per draw (round 6, kf_smoother_bf_lane.hip): an ARMA model (r = 1, H = 0) puts the prediction covariance on
                 pinv's cut-off at almost every step and the RTS lane kernel's Jacobi fallback agrees with statsmodels only to 1e-2
                 there, this recursion to 1e-9 (tests/golden/arma11.npz);
      * 5 <= m <= 15   the block-tile kernel (kf_smoother_bf.hip);
      * 16 <= m <= 79  16 x 16 fp64 MFMA tiles, one workgroup per draw (round 6, kf_smoother_bf_wg.hip): 1024 draws x 300 steps,
                 m = 20 / 32 / 48 / 63: 1.9 / 8.2 / 13.0 / 13.3 ms whatever the rank of the state noise, against 3.6 / 7.8 / 22 / 42 ms of
                 the RTS tile kernels with full-rank noise and 0.19 / 2.0 / 7.3 / 37.7 SECONDS with r = 1 (every ARMA / SARIMA model:
                 most draws go to the pinv kernel), 34 / 43 / 58 / 77 ms on round 5's workspace path (profiles/r06/smooth_wg_time.txt);
      * 64 < m <= 1024  (beyond 79 on the HBM workspace) no factorisation: C5, 128 draws, m = 367: 0.80 against 2.78 ms per step.
    The one exception is m == 64 exactly (the vector carrier needs a 65th tile index: five wavefronts on four SIMDs, 49.5 ms): the RTS
    workgroup kernel is the faster one there with full-rank noise (42.7 ms), so m == 64 keeps round 5's rank rule, 12 r <= m + 11.
    A non-zero state intercept c is the CALLER's business: the reference's smoother predicts with a_hat = T a_f, without c
    (kalman_smoother.py:99-104), while the filter's predicted tapes contain it, so the two recursions differ by O(c) and a caller that
    wants the reference's numbers must keep kf_smoother (``smooth_result`` and the fused node check c themselves)."""
    nds = list(ndims)
    if p != 1 or not all(nd == 2 for nd in nds[:5]) or any(nd != 1 for nd in nds[5:]):
        return False
    if m == 64:
        return r is not None and 12 * r <= m + 11
    return 1 <= m <= 1024


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class KalmanEngine:
    """One ``kf_handle`` bound to one GPU."""

    def __init__(self, device=None, lanes_per_draw: int = 0):
        if not torch.cuda.is_available():
            raise KalmanError("KalmanEngine needs a HIP device (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.type != "cuda":
            raise KalmanError(f"device must be a cuda/HIP device, got {self.device}")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        h = C.c_void_p()
        rc = self.lib.kf_create(C.byref(h), idx)
        if rc != 0:
            raise KalmanError(f"kf_create failed ({rc}): {self.lib.kf_last_error(None).decode()}")
        self._h = h
        if lanes_per_draw:
            self.set_lanes_per_draw(lanes_per_draw)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.kf_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ utils
    def _check(self, rc, what):
        if rc != 0:
            raise KalmanError(f"{what} failed ({rc}): {self.lib.kf_last_error(self._h).decode()}")

    def set_lanes_per_draw(self, lanes: int):
        self._check(self.lib.kf_set_lanes_per_draw(self._h, int(lanes)), "kf_set_lanes_per_draw")

    def set_option(self, name: str, value):
        """kf_set_option: a KF_* path switch on THIS engine's handle -- it wins over the process environment for
        every call made through this engine.  ``value`` None removes the override, '' means 'as if unset'."""
        v = None if value is None else str(value).encode()
        self._check(self.lib.kf_set_option(self._h, name.encode(), v), "kf_set_option")

    def get_option(self, name: str):
        """The value a call through this engine would see for the switch (handle first, then the environment); None = unset."""
        v = self.lib.kf_get_option(self._h, name.encode())
        return None if v is None else v.decode()

    def set_forward_kernel(self, kernel: str):
        """'auto' | 'lds' (generic LDS-resident) | 'row' (register-resident: p == 1, m <= 10; or
        m <= 16, p <= 8 with the Univariate filter / diagonal H) | 'lane' (one lane per draw,
        p == 1, m <= 4) | 'mfma' (covariance in MFMA accumulator layout, one wavefront per draw:
        m <= 16, p <= 8, same eligibility as the multivariate row kernel; 16 < m <= 48 as blocks of
        16 x 16) | 'b4' (v_mfma_f64_4x4x4_4b_f64 blocks, mean carried in the tile: p == 1, m <= 14; 2 <= p <= 8 scalar
        observations -- Univariate filter / diagonal H -- with m <= 15)."""
        code = {"auto": 0, "lds": 1, "row": 2, "lane": 3, "mfma": 4, "b4": 5}[kernel]
        self._check(self.lib.kf_set_forward_kernel(self._h, code), "kf_set_forward_kernel")

    def set_exact_recursion(self, on: bool = True):
        """Default of ``forward(..., exact_recursion=None)`` for this engine (the PyTensor Ops use the shared engine):
        True passes KF_FLAG_EXACT_RECURSION, i.e. the forward kernels never switch to their steady-state phase."""
        self._exact_recursion = bool(on)

    def last_forward_kernel(self) -> str:
        """kf_last_forward_kernel: which kernel carried the recursion of the last ``forward`` on the workspace path (m > 64) --
        "bsp_step_kernel" (T sparse: CSR gather) or "big_gemm_kernel" (fp64 MFMA products); "" before the first."""
        return self.lib.kf_last_forward_kernel(self._h).decode()

    def forward_kernel_name(self, B, n, p, m, r, filter_type="standard", h_diagonal=False, time_varying=()) -> str:
        """Kernel ``forward`` launches for this shape (``forward`` always passes the workspace)."""
        dims = self._dims(B, n, p, m, r, filter_type)
        if h_diagonal:
            dims.flags |= _lib.KF_FLAG_H_DIAGONAL
        for name in time_varying:
            dims.flags |= _lib.KF_FLAG_TV[name]
        return self.lib.kf_forward_kernel_name(self._h, C.byref(dims)).decode()

    def _workspace(self, dims, which=_lib.KF_WS_ALL):
        """Scratch for the entry point about to be called (kf_workspace_bytes_for): only what THAT
        call needs, cached per (entry point, stream) and grown on demand -- a forward-only caller at
        m = 367 never holds the gradient's 7 B m^2 + 2 B m T doubles, and calls on different streams
        never share scratch.  ``release_workspace()`` drops the cache."""
        nbytes = C.c_size_t(0)
        self._check(self.lib.kf_workspace_bytes_for(self._h, C.byref(dims), which, C.byref(nbytes)),
                    "kf_workspace_bytes_for")
        if nbytes.value == 0:
            return None, 0
        cache = self.__dict__.setdefault("_ws", {})
        key = (which, torch.cuda.current_stream(self.device).cuda_stream)
        ws = cache.get(key)
        if ws is None or ws.numel() < nbytes.value:
            cache.pop(key, None)
            ws = None  # free the old block before asking for the larger one
            cache[key] = ws = torch.empty(nbytes.value, dtype=torch.uint8, device=self.device)
        return ws, ws.numel()

    def release_workspace(self):
        self.__dict__.pop("_ws", None)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def to_device(self, x, shape=None):
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=torch.float64)
        else:
            t = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=self.device)
        t = t.contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    @staticmethod
    def _dims(B, n, p, m, r, filter_type="standard", mvn_const="reference", y_per_draw=False):
        if filter_type not in FILTER_TYPES:
            raise NotImplementedError(
                "The following are valid filter types: " + ", ".join(FILTER_TYPES))
        flags = 0
        if mvn_const == "exact":
            flags |= _lib.KF_FLAG_MVN_CONST_EXACT
        elif mvn_const != "reference":
            raise ValueError("mvn_const must be 'reference' or 'exact'")
        if y_per_draw:
            flags |= _lib.KF_FLAG_Y_PER_DRAW
        return _lib.KfDims(B, n, p, m, r, _lib.KF_FILTER[filter_type], flags)

    # ---------------------------------------------------------------- forward
    def forward(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, filter_type="standard",
                mvn_const="reference", store_ll_obs=True, store_filtered=True,
                store_predicted=True, time_varying=(), h_diagonal: Optional[bool] = None,
                exact_recursion: Optional[bool] = None,
                out: Optional[FilterResult] = None) -> FilterResult:
        """kf_filter_forward.  y (n,p) shared or (B,n,p); a0 (B,m); P0,T (B,m,m); Z (B,p,m);
        R (B,m,r); H (B,p,p); Q (B,r,r); c (B,m); d (B,p).  Names listed in ``time_varying``
        (subset of T,Z,R,H,Q,c,d) carry an extra time axis after the draw axis: (B,n,...).

        ``h_diagonal``: True asserts every draw's H is diagonal (KF_FLAG_H_DIAGONAL: lets the
        Standard/Cholesky filters with p > 1, m <= 48 use the register-resident kernels);
        None checks H on the device (one small reduction + host sync) when it matters.

        ``exact_recursion``: True sets KF_FLAG_EXACT_RECURSION (the covariance recursion at every step, as the reference), False
        sets KF_FLAG_STEADY_STATE (stop recomputing it once the predicted covariance has converged and no observation is
        missing ahead, wherever the selected kernel can), None leaves the choice to the library (include/kf.h: lane kernel
        m <= 4 and the multi-observation block kernel use the phase, the p == 1 block kernel from 1536 steps on) unless
        ``set_exact_recursion(True)`` made exact the engine's default.  Results agree to ~1e-13 relative."""
        if exact_recursion is None:
            exact_recursion = True if getattr(self, "_exact_recursion", False) else None
        tv = set(time_varying)
        if not tv <= set(_lib.KF_FLAG_TV):
            raise ValueError(f"time_varying must be a subset of {sorted(_lib.KF_FLAG_TV)}")
        P0 = self.to_device(P0)
        if P0.ndim != 3 or P0.shape[1] != P0.shape[2]:
            raise ValueError(f"P0 must be (B,m,m), got {tuple(P0.shape)}")
        B, m = P0.shape[0], P0.shape[1]
        Z = self.to_device(Z)
        if Z.ndim != (4 if "Z" in tv else 3) or Z.shape[0] != B or Z.shape[-1] != m:
            raise ValueError(f"Z must be (B,[n,]p,m) with B={B}, m={m}; got {tuple(Z.shape)}")
        p = Z.shape[-2]
        R = self.to_device(R)
        if R.ndim != (4 if "R" in tv else 3) or R.shape[0] != B or R.shape[-2] != m:
            raise ValueError(f"R must be (B,[n,]m,r), got {tuple(R.shape)}")
        r = R.shape[-1]
        y = self.to_device(y)
        y_per_draw = y.ndim == 3
        if y.ndim not in (2, 3) or y.shape[-1] != p or (y_per_draw and y.shape[0] != B):
            raise ValueError(f"y must be (n,{p}) or ({B},n,{p}), got {tuple(y.shape)}")
        n = y.shape[-2]

        def shp(name, *core):
            return (B, n) + core if name in tv else (B,) + core

        def tvcheck(name, x, *core):
            # assert_time_varying_dim_correct, reference filters/kalman_filter.py:20-23,:79-90
            if name in tv and x is not None and x.ndim == 2 + len(core) and x.shape[1] != n:
                raise AssertionError(
                    "The first dimension of a time varying matrix (the time dimension) must be "
                    "equal to the first dimension of the data (the time dimension).")
            return self.to_device(x, shp(name, *core)) if x is not None else None

        a0 = self.to_device(a0, (B, m))
        T = tvcheck("T", self.to_device(T), m, m)
        Z = tvcheck("Z", Z, p, m)
        R = tvcheck("R", R, m, r)
        H = tvcheck("H", self.to_device(H), p, p)
        Q = tvcheck("Q", self.to_device(Q), r, r)
        c = tvcheck("c", self.to_device(c), m)
        d = tvcheck("d", self.to_device(d), p)
        if filter_type == "single" and p != 1:
            # assert_data_is_1d, reference filters/kalman_filter.py:19,:329
            raise AssertionError("UnivariateTimeSeries filter requires data be at most 1-dimensional")

        dims = self._dims(B, n, p, m, r, filter_type, mvn_const, y_per_draw)
        for name in tv:
            dims.flags |= _lib.KF_FLAG_TV[name]
        if h_diagonal is None:
            h_diagonal = False
            if p > 1 and m <= 64 and p <= 8 and not tv and filter_type in ("standard", "cholesky"):
                # the check costs a few small launches and a host sync: remember the answer per tensor
                # (keyed on the live tensor OBJECT and its version counter -- not on the address: the
                # caching allocator hands the same block to the next, different H)
                ref = getattr(self, "_hdiag_ref", None)
                if ref is None or ref() is not H or self._hdiag_ver != H._version:
                    off = H - torch.diag_embed(torch.diagonal(H, dim1=-2, dim2=-1))
                    self._hdiag_val = not bool(torch.any(off != 0))
                    self._hdiag_ref, self._hdiag_ver = weakref.ref(H), H._version
                h_diagonal = self._hdiag_val
        if h_diagonal:
            dims.flags |= _lib.KF_FLAG_H_DIAGONAL
        if exact_recursion:
            dims.flags |= _lib.KF_FLAG_EXACT_RECURSION
        elif exact_recursion is not None:  # False: the caller asks for the steady-state phase (KF_FLAG_STEADY_STATE)
            dims.flags |= _lib.KF_FLAG_STEADY_STATE
        dev, f64 = self.device, torch.float64
        if out is None:
            out = FilterResult(
                logp=torch.empty(B, device=dev, dtype=f64),
                ll_obs=torch.empty(B, n, device=dev, dtype=f64) if store_ll_obs else None,
                filtered_states=torch.empty(B, n, m, device=dev, dtype=f64) if store_filtered else None,
                predicted_states=torch.empty(B, n + 1, m, device=dev, dtype=f64) if store_predicted else None,
                filtered_covs=torch.empty(B, n, m, m, device=dev, dtype=f64) if store_filtered else None,
                predicted_covs=torch.empty(B, n + 1, m, m, device=dev, dtype=f64) if store_predicted else None,
                info=torch.empty(B, device=dev, dtype=torch.int32),
            )
        ws, ws_bytes = self._workspace(dims, _lib.KF_WS_FORWARD)
        rc = self.lib.kf_filter_forward(
            self._h, C.byref(dims), _ptr(y), _ptr(a0), _ptr(P0), _ptr(T), _ptr(Z), _ptr(R), _ptr(H),
            _ptr(Q), _ptr(c), _ptr(d), _ptr(out.logp), _ptr(out.ll_obs), _ptr(out.filtered_states),
            _ptr(out.predicted_states), _ptr(out.filtered_covs), _ptr(out.predicted_covs),
            _ptr(out.info), _ptr(ws), ws_bytes, self._stream())
        self._check(rc, "kf_filter_forward")
        return out

    # --------------------------------------------------------------- backward
    def backward(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, predicted_states,
                 predicted_covs, g_logp=None, g_ll_obs=None, g_filtered_states=None, g_predicted_states=None,
                 g_filtered_covs=None, g_predicted_covs=None, filter_type="standard",
                 mvn_const="reference", want_cd=False, h_diagonal=False, time_varying=()):
        """kf_filter_backward -> dict of per-draw gradients (a0,P0,T,Z,R,H,Q[,c,d]).

        ``g_filtered_states`` (B,n,m), ``g_predicted_states`` (B,n+1,m), ``g_filtered_covs`` (B,n,m,m),
        ``g_predicted_covs`` (B,n+1,m,m): cotangents of the forward's tape outputs -- with any of them the call is
        kf_filter_backward_tapes (the LDS-resident adjoint; KF_ERR_UNSUPPORTED beyond m ~ 45).

        ``h_diagonal=True`` (KF_FLAG_H_DIAGONAL) lets Standard/Cholesky with p > 1, m <= 48 use the
        scalar-observation adjoint kernel: g["H"] then carries the diagonal only and g["P0"] the
        symmetric part of the entrywise gradient (all a symmetric / diagonal parameterisation sees).
        Unlike ``forward`` this is never auto-detected: a dense-H model that happens to be diagonal
        at the current point still needs its off-diagonal partials.

        ``time_varying`` as in ``forward``; the gradient of a matrix named there has its (B,n,.,.) shape."""
        tv = set(time_varying)
        if not tv <= set(_lib.KF_FLAG_TV):
            raise ValueError(f"time_varying must be a subset of {sorted(_lib.KF_FLAG_TV)}")
        P0 = self.to_device(P0)
        B, m = P0.shape[0], P0.shape[1]
        Z = self.to_device(Z)
        p = Z.shape[-2]
        R = self.to_device(R)
        r = R.shape[-1]
        y = self.to_device(y)
        y_per_draw = y.ndim == 3
        n = y.shape[-2]

        def shp(name, *core):
            return (B, n) + core if name in tv else (B,) + core

        a0 = self.to_device(a0, (B, m))
        T = self.to_device(T, shp("T", m, m))
        Z = self.to_device(Z, shp("Z", p, m))
        R = self.to_device(R, shp("R", m, r))
        H = self.to_device(H, shp("H", p, p))
        Q = self.to_device(Q, shp("Q", r, r))
        c = self.to_device(c, shp("c", m)) if c is not None else None
        d = self.to_device(d, shp("d", p)) if d is not None else None
        pa = self.to_device(predicted_states, (B, n + 1, m))
        pP = self.to_device(predicted_covs, (B, n + 1, m, m))
        g_logp = self.to_device(g_logp, (B,)) if g_logp is not None else None
        g_ll = self.to_device(g_ll_obs, (B, n)) if g_ll_obs is not None else None
        g_fa = self.to_device(g_filtered_states, (B, n, m)) if g_filtered_states is not None else None
        g_pa = self.to_device(g_predicted_states, (B, n + 1, m)) if g_predicted_states is not None else None
        g_fP = self.to_device(g_filtered_covs, (B, n, m, m)) if g_filtered_covs is not None else None
        g_pP = self.to_device(g_predicted_covs, (B, n + 1, m, m)) if g_predicted_covs is not None else None
        tapes = any(x is not None for x in (g_fa, g_pa, g_fP, g_pP))
        dims = self._dims(B, n, p, m, r, filter_type, mvn_const, y_per_draw)
        for name in tv:
            dims.flags |= _lib.KF_FLAG_TV[name]
        if h_diagonal:
            dims.flags |= _lib.KF_FLAG_H_DIAGONAL
        dev, f64 = self.device, torch.float64

        def empty(name, *core):
            return torch.empty(shp(name, *core), device=dev, dtype=f64)

        g = dict(
            a0=torch.empty(B, m, device=dev, dtype=f64), P0=torch.empty(B, m, m, device=dev, dtype=f64),
            T=empty("T", m, m), Z=empty("Z", p, m), R=empty("R", m, r), H=empty("H", p, p), Q=empty("Q", r, r),
        )
        if want_cd:
            g["c"] = empty("c", m)
            g["d"] = empty("d", p)
        ws, ws_bytes = self._workspace(dims, _lib.KF_WS_BACKWARD)  # large m only (kf_backward_big.hip)
        outs = (_ptr(g["a0"]), _ptr(g["P0"]), _ptr(g["T"]), _ptr(g["Z"]), _ptr(g["R"]), _ptr(g["H"]),
                _ptr(g["Q"]), _ptr(g.get("c")), _ptr(g.get("d")), _ptr(ws), ws_bytes, self._stream())
        ins = (self._h, C.byref(dims), _ptr(y), _ptr(a0), _ptr(P0), _ptr(T), _ptr(Z), _ptr(R), _ptr(H),
               _ptr(Q), _ptr(c), _ptr(d), _ptr(pa), _ptr(pP), _ptr(g_logp), _ptr(g_ll))
        if tapes:
            rc = self.lib.kf_filter_backward_tapes(*ins, _ptr(g_fa), _ptr(g_pa), _ptr(g_fP), _ptr(g_pP), *outs)
            self._check(rc, "kf_filter_backward_tapes")
        else:
            rc = self.lib.kf_filter_backward(*ins, *outs)
            self._check(rc, "kf_filter_backward")
        return g

    def value_and_grad(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, g_logp=None, filter_type="standard",
                       mvn_const="reference", h_diagonal=False, exact_recursion: Optional[bool] = None):
        """kf_filter_value_and_grad -> (logp (B,), info (B,) int32, dict of gradients a0,P0,T,Z,R,H,Q,c,d) of
        sum_b g_logp[b] logp[b] (default: ones) in ONE call: the predicted tapes live in the workspace only.
        ``exact_recursion`` as in ``forward`` (None: the engine's default, ``set_exact_recursion``)."""
        if exact_recursion is None:
            exact_recursion = True if getattr(self, "_exact_recursion", False) else None
        P0 = self.to_device(P0)
        B, m = P0.shape[0], P0.shape[1]
        Z = self.to_device(Z, None)
        p = Z.shape[-2]
        R = self.to_device(R)
        r = R.shape[-1]
        y = self.to_device(y)
        y_per_draw = y.ndim == 3
        n = y.shape[-2]
        a0 = self.to_device(a0, (B, m))
        T = self.to_device(T, (B, m, m))
        Z = self.to_device(Z, (B, p, m))
        R = self.to_device(R, (B, m, r))
        H = self.to_device(H, (B, p, p))
        Q = self.to_device(Q, (B, r, r))
        c = self.to_device(c, (B, m)) if c is not None else None
        d = self.to_device(d, (B, p)) if d is not None else None
        g_logp = self.to_device(g_logp, (B,)) if g_logp is not None else None
        dims = self._dims(B, n, p, m, r, filter_type, mvn_const, y_per_draw)
        if h_diagonal:
            dims.flags |= _lib.KF_FLAG_H_DIAGONAL
        if exact_recursion:
            dims.flags |= _lib.KF_FLAG_EXACT_RECURSION
        elif exact_recursion is not None:  # False: the caller asks for the steady-state phase (KF_FLAG_STEADY_STATE)
            dims.flags |= _lib.KF_FLAG_STEADY_STATE
        dev, f64 = self.device, torch.float64
        logp = torch.empty(B, device=dev, dtype=f64)
        info = torch.empty(B, device=dev, dtype=torch.int32)
        g = dict(a0=torch.empty(B, m, device=dev, dtype=f64), P0=torch.empty(B, m, m, device=dev, dtype=f64),
                 T=torch.empty(B, m, m, device=dev, dtype=f64), Z=torch.empty(B, p, m, device=dev, dtype=f64),
                 R=torch.empty(B, m, r, device=dev, dtype=f64), H=torch.empty(B, p, p, device=dev, dtype=f64),
                 Q=torch.empty(B, r, r, device=dev, dtype=f64), c=torch.empty(B, m, device=dev, dtype=f64),
                 d=torch.empty(B, p, device=dev, dtype=f64))
        ws, ws_bytes = self._workspace(dims, _lib.KF_WS_VALUE_AND_GRAD)
        rc = self.lib.kf_filter_value_and_grad(
            self._h, C.byref(dims), _ptr(y), _ptr(a0), _ptr(P0), _ptr(T), _ptr(Z), _ptr(R), _ptr(H), _ptr(Q), _ptr(c),
            _ptr(d), _ptr(g_logp), _ptr(logp), _ptr(info), _ptr(g["a0"]), _ptr(g["P0"]), _ptr(g["T"]), _ptr(g["Z"]),
            _ptr(g["R"]), _ptr(g["H"]), _ptr(g["Q"]), _ptr(g["c"]), _ptr(g["d"]), _ptr(ws), ws_bytes, self._stream())
        self._check(rc, "kf_filter_value_and_grad")
        return logp, info, g

    # --------------------------------------------------------------- smoother
    def smooth(self, T, R, Q, filtered_states, filtered_covs, time_varying=(), pairing="filter",
               exact_recursion: Optional[bool] = None):
        """kf_smoother -> (smoothed_states (B,n,m), smoothed_covs (B,n,m,m)).  Names in ``time_varying`` (subset of
        T, R, Q) carry a time axis after the draw axis, (B,n,.,.): the step that smooths t uses slice t (``pairing="filter"``,
        the forward filter's pairing) or slice t + 1 (``pairing="reference"``: what the reference's own scan hands its step,
        kalman_smoother.py:62-77 -- for parity with a model ported from upstream).  ``exact_recursion=True``
        (KF_FLAG_EXACT_RECURSION): the prediction covariance is inverted at every step, also where the filtered covariance on the
        tape is bit for bit the next step's (5 <= m <= 15, m <= 2: the smoothers reuse the last step's gain there by default);
        None: the engine's default (``set_exact_recursion``), as in ``forward`` / ``value_and_grad``."""
        if exact_recursion is None:
            exact_recursion = bool(getattr(self, "_exact_recursion", False))
        if pairing not in ("filter", "reference"):
            raise ValueError("pairing must be 'filter' or 'reference'")
        tv = set(time_varying)
        if not tv <= {"T", "R", "Q"}:
            raise ValueError("time_varying must be a subset of {'T', 'R', 'Q'}")
        T = self.to_device(T)
        B, m = T.shape[0], T.shape[-1]
        R = self.to_device(R)
        r = R.shape[-1]
        fa = self.to_device(filtered_states)
        n = fa.shape[1]
        fa = self.to_device(fa, (B, n, m))
        fP = self.to_device(filtered_covs, (B, n, m, m))
        T = self.to_device(T, (B, n, m, m) if "T" in tv else (B, m, m))
        R = self.to_device(R, (B, n, m, r) if "R" in tv else (B, m, r))
        Q = self.to_device(Q, (B, n, r, r) if "Q" in tv else (B, r, r))
        dims = self._dims(B, n, 1, m, r)
        for name in tv:
            dims.flags |= _lib.KF_FLAG_TV[name]
        if tv and pairing == "reference":
            dims.flags |= _lib.KF_FLAG_TV_SMOOTHER_REFERENCE_PAIRING
        if exact_recursion:
            dims.flags |= _lib.KF_FLAG_EXACT_RECURSION
        sa = torch.empty_like(fa)
        sP = torch.empty_like(fP)
        ws, ws_bytes = self._workspace(dims, _lib.KF_WS_SMOOTHER)  # large m only (kf_smoother_big.hip)
        rc = self.lib.kf_smoother(self._h, C.byref(dims), _ptr(T), _ptr(R), _ptr(Q), _ptr(fa),
                                  _ptr(fP), _ptr(sa), _ptr(sP), _ptr(ws), ws_bytes, self._stream())
        self._check(rc, "kf_smoother")
        return sa, sP


    def smooth_from_filter_ok(self, p: int, m: int, time_varying=()) -> bool:
        """Does kf_smoother_from_filter take this shape (p == 1, m <= 1024, static matrices: one lane per draw up to m = 4, on chip up
        to m = 15, on the HBM workspace beyond)?  (Whether it is the FASTER recursion: ``smoother_from_filter_applies``.)"""
        return p == 1 and 1 <= m <= 1024 and not tuple(time_varying)

    def smooth_from_filter(self, data, T, Z, H, predicted_states, predicted_covs, d=None, filter_type: str = "standard"):
        """kf_smoother_from_filter -> (smoothed_states (B,n,m), smoothed_covs (B,n,m,m)) from the FILTER's predicted tapes
        (B,n+1,m), (B,n+1,m,m) and the observation side of the model it ran on (data, Z, H, d) -- the same moments
        ``smooth`` computes from the filtered tapes (kalman_smoother.py:56-104) up to rounding, without inverting the prediction
        covariance (include/kf.h).  p == 1, static matrices, m <= 4 one lane per draw, m <= 15 on chip and 15 < m <= 1024 on the workspace path;
        ``filter_type="single"`` keeps that filter's sign of d."""
        if filter_type not in _lib.KF_FILTER:
            raise ValueError(f"unknown filter_type {filter_type!r}")
        T = self.to_device(T)
        B, m = T.shape[0], T.shape[-1]
        T = self.to_device(T, (B, m, m))
        pa = self.to_device(predicted_states)
        n = pa.shape[1] - 1
        pa = self.to_device(pa, (B, n + 1, m))
        pP = self.to_device(predicted_covs, (B, n + 1, m, m))
        Z = self.to_device(Z, (B, 1, m))
        H = self.to_device(H, (B, 1, 1))
        d = None if d is None else self.to_device(d, (B, 1))
        y = self.to_device(data)
        per_draw = y.ndim == 3  # (n, 1) shared or (B, n, 1), as in forward
        y = self.to_device(y, (B, n, 1) if per_draw else (n, 1))
        dims = self._dims(B, n, 1, m, 1)
        dims.filter = _lib.KF_FILTER[filter_type]
        if per_draw:
            dims.flags |= _lib.KF_FLAG_Y_PER_DRAW
        sa = torch.empty(B, n, m, device=self.device, dtype=torch.float64)
        sP = torch.empty(B, n, m, m, device=self.device, dtype=torch.float64)
        ws, ws_bytes = self._workspace(dims, _lib.KF_WS_SMOOTHER_FROM_FILTER)   # (m > 15 only: 5 B m^2 doubles + vectors)
        rc = self.lib.kf_smoother_from_filter(self._h, C.byref(dims), _ptr(y), _ptr(T), _ptr(Z), _ptr(H), _ptr(d),
                                              _ptr(pa), _ptr(pP), _ptr(sa), _ptr(sP), _ptr(ws), ws_bytes, self._stream())
        self._check(rc, "kf_smoother_from_filter")
        return sa, sP


    def smooth_result(self, res: FilterResult, data, T, Z, R, H, Q, c=None, d=None, *, filter_type: str = "standard",
                      time_varying=(), route: str = "auto", **smooth_kw):
        """Smoothed moments of the model whose ``forward`` produced ``res`` -- build_statespace_graph + build_smoother_graph
        (core/statespace.py:164-197) for a caller that holds the filter's result: from the PREDICTED tapes (kf_smoother_from_filter: no
        pinv, every draw at the same cost) wherever ``smoother_from_filter_applies`` says so, the tapes are in ``res`` and the state
        intercept c is absent or all zero (one device reduction + host sync when c is given); else the reference's recursion on the
        filtered tapes (kf_smoother).  ``route``: "auto" | "from_filter" (raise if it does not apply) | "rts".  Returns
        (smoothed_states (B,n,m), smoothed_covs (B,n,m,m))."""
        if route not in ("auto", "from_filter", "rts"):
            raise ValueError("route must be 'auto', 'from_filter' or 'rts'")
        tv = tuple(time_varying)
        T_ = self.to_device(T)
        Z_ = self.to_device(Z)
        R_ = self.to_device(R)
        m, p, r = T_.shape[-1], Z_.shape[-2], R_.shape[-1]
        ok = route != "rts" and not tv and res.predicted_states is not None and res.predicted_covs is not None
        if ok:
            ok = smoother_from_filter_applies(p, m, (2,) * 5, r=r) if route == "auto" else self.smooth_from_filter_ok(p, m)
        if ok and c is not None:
            ok = not bool(torch.any(self.to_device(c) != 0))
        if ok:
            dv = None if d is None else self.to_device(d).reshape(T_.shape[0], p)
            return self.smooth_from_filter(data, T_, Z_, H, res.predicted_states, res.predicted_covs, d=dv,
                                           filter_type=filter_type)
        if route == "from_filter":
            raise KalmanError("smooth_result(route='from_filter'): needs p == 1, static matrices, m <= 1024, the predicted tapes in "
                              "the filter result and a zero state intercept c")
        if res.filtered_states is None or res.filtered_covs is None:
            raise KalmanError("smooth_result: the filter result holds no filtered tapes (forward(..., store_filtered=True))")
        return self.smooth(T_, R_, Q, res.filtered_states, res.filtered_covs,
                           time_varying=tuple(nm for nm in tv if nm in ("T", "R", "Q")), **smooth_kw)

    # ------------------------------------------------- before the filter (section 8f row 1)
    def lyapunov(self, T, R=None, Q=None, S=None):
        """kf_lyapunov: P (B,m,m) with P = T P T' + S, S given or R Q R'.  Returns (P, info)."""
        T = self.to_device(T)
        B, m = T.shape[0], T.shape[1]
        if S is not None:
            S = self.to_device(S, (B, m, m))
            r = 0
        else:
            R = self.to_device(R)
            r = R.shape[2]
            R = self.to_device(R, (B, m, r))
            Q = self.to_device(Q, (B, r, r))
        P = torch.empty(B, m, m, device=self.device, dtype=torch.float64)
        info = torch.empty(B, device=self.device, dtype=torch.int32)
        rc = self.lib.kf_lyapunov(self._h, B, m, r, _ptr(T), _ptr(R), _ptr(Q), _ptr(S), _ptr(P),
                                  _ptr(info), self._stream())
        self._check(rc, "kf_lyapunov")
        return P, info

    def assemble(self, theta, template, dst, src):
        """kf_assemble: out (B,total) = template with out[:, dst] = theta[:, src]."""
        theta = self.to_device(theta)
        B, n_theta = theta.shape
        template = self.to_device(template)
        total = template.numel()
        dst = torch.as_tensor(np.asarray(dst, dtype=np.int32), device=self.device)
        src = torch.as_tensor(np.asarray(src, dtype=np.int32), device=self.device)
        out = torch.empty(B, total, device=self.device, dtype=torch.float64)
        rc = self.lib.kf_assemble(self._h, B, n_theta, _ptr(theta), total, _ptr(template),
                                  dst.numel(), _ptr(dst), _ptr(src), _ptr(out), self._stream())
        self._check(rc, "kf_assemble")
        return out


    # -------------------------------------------------- after the filter (section 8f row 2)
    def conditional_simulation(self, mus, covs, n_simulations=100, jitter=1e-10, generator=None, z=None):
        """kf_conditional_simulation: mus (B,n,k), covs (B,n,k,k) -> (B*n_simulations, n, k), ordered
        like the reference (sample-major: simulation j of sample i at row i*n_simulations + j)."""
        mus = self.to_device(mus)
        B, n, k = mus.shape
        covs = self.to_device(covs, (B, n, k, k))
        if z is None:
            z = torch.randn(B, n_simulations, n, k, device=self.device, dtype=torch.float64,
                            generator=generator)
        else:
            z = self.to_device(z, (B, n_simulations, n, k))
        out = torch.empty(B * n_simulations, n, k, device=self.device, dtype=torch.float64)
        info = torch.empty(B, device=self.device, dtype=torch.int32)
        rc = self.lib.kf_conditional_simulation(self._h, B, n, k, n_simulations, _ptr(mus), _ptr(covs),
                                                _ptr(z), float(jitter), _ptr(out), _ptr(info),
                                                self._stream())
        self._check(rc, "kf_conditional_simulation")
        return out, info

    # ------------------------------------------------- SURVEY section 8f row 4
    def dare(self, T, Z, R, H, Q, P_init=None, max_iter=100000, tol=1e-14):
        """kf_dare: steady-state predicted covariance per draw, (P (B,m,m), info (B,) int32)."""
        T = self.to_device(T)
        B, m = T.shape[0], T.shape[1]
        Z = self.to_device(Z)
        p = Z.shape[1]
        R = self.to_device(R)
        r = R.shape[2]
        Z = self.to_device(Z, (B, p, m))
        H = self.to_device(H, (B, p, p))
        Q = self.to_device(Q, (B, r, r))
        P_init = self.to_device(P_init, (B, m, m)) if P_init is not None else None
        P = torch.empty(B, m, m, device=self.device, dtype=torch.float64)
        info = torch.empty(B, device=self.device, dtype=torch.int32)
        rc = self.lib.kf_dare(self._h, B, m, p, r, _ptr(T), _ptr(Z), _ptr(R), _ptr(H), _ptr(Q), _ptr(P_init),
                              _ptr(P), _ptr(info), int(max_iter), float(tol), self._stream())
        self._check(rc, "kf_dare")
        return P, info

    def dare_backward(self, T, Z, R, H, Q, P, g_P, into=None):
        """kf_dare_backward: gradient of <g_P, P_steady> w.r.t. (T, Z, R, H, Q) per draw -> (dict, info).
        ``into``: a dict of gradient tensors (as ``backward`` returns) to ADD the result to, in place."""
        T = self.to_device(T)
        B, m = T.shape[0], T.shape[1]
        Z = self.to_device(Z)
        p = Z.shape[1]
        R = self.to_device(R)
        r = R.shape[2]
        H = self.to_device(H, (B, p, p))
        Q = self.to_device(Q, (B, r, r))
        P = self.to_device(P, (B, m, m))
        g_P = self.to_device(g_P, (B, m, m))
        dev, f64 = self.device, torch.float64
        if into is None:
            g = dict(T=torch.empty(B, m, m, device=dev, dtype=f64), Z=torch.empty(B, p, m, device=dev, dtype=f64),
                     R=torch.empty(B, m, r, device=dev, dtype=f64), H=torch.empty(B, p, p, device=dev, dtype=f64),
                     Q=torch.empty(B, r, r, device=dev, dtype=f64))
        else:
            g = into
            for k, shape in (("T", (B, m, m)), ("Z", (B, p, m)), ("R", (B, m, r)), ("H", (B, p, p)), ("Q", (B, r, r))):
                if tuple(g[k].shape) != shape or not g[k].is_contiguous():
                    raise ValueError(f"into[{k!r}] must be a contiguous {shape} tensor")
        info = torch.empty(B, device=dev, dtype=torch.int32)
        rc = self.lib.kf_dare_backward(self._h, B, m, p, r, _ptr(T), _ptr(Z), _ptr(R), _ptr(H), _ptr(Q), _ptr(P),
                                       _ptr(g_P), _ptr(g["T"]), _ptr(g["Z"]), _ptr(g["R"]), _ptr(g["H"]), _ptr(g["Q"]),
                                       1 if into is not None else 0, _ptr(info), self._stream())
        self._check(rc, "kf_dare_backward")
        return g, info

    def simulate_statespace(self, T, Z, R, H, Q, n_steps, n_simulations=1, x0=None, generator=None,
                            z_state=None, z_obs=None):
        """kf_simulate_statespace: (states (B*n_simulations, n_steps, m), obs (B*n_simulations, n_steps, p),
        info (B,)).  Standard normals z_state (B,n_sims,n_steps,r) / z_obs (B,n_sims,n_steps,p) are drawn
        with ``generator`` unless given."""
        T = self.to_device(T)
        B, m = T.shape[0], T.shape[1]
        Z = self.to_device(Z)
        p = Z.shape[1]
        R = self.to_device(R)
        r = R.shape[2]
        H = self.to_device(H, (B, p, p))
        Q = self.to_device(Q, (B, r, r))
        x0 = self.to_device(x0, (B, m)) if x0 is not None else None
        dev, f64 = self.device, torch.float64
        if z_state is None:
            z_state = torch.randn(B, n_simulations, n_steps, r, device=dev, dtype=f64, generator=generator)
        else:
            z_state = self.to_device(z_state, (B, n_simulations, n_steps, r))
        if z_obs is None:
            z_obs = torch.randn(B, n_simulations, n_steps, p, device=dev, dtype=f64, generator=generator)
        else:
            z_obs = self.to_device(z_obs, (B, n_simulations, n_steps, p))
        states = torch.empty(B * n_simulations, n_steps, m, device=dev, dtype=f64)
        obs = torch.empty(B * n_simulations, n_steps, p, device=dev, dtype=f64)
        info = torch.empty(B, device=dev, dtype=torch.int32)
        rc = self.lib.kf_simulate_statespace(self._h, B, n_simulations, n_steps, p, m, r, _ptr(T), _ptr(Z),
                                             _ptr(R), _ptr(H), _ptr(Q), _ptr(x0), _ptr(z_state), _ptr(z_obs),
                                             _ptr(states), _ptr(obs), _ptr(info), self._stream())
        self._check(rc, "kf_simulate_statespace")
        return states, obs, info


_engines = {}


def get_engine(device=None) -> KalmanEngine:
    """Process-wide engine per device (one kf_handle per (process, device))."""
    if not torch.cuda.is_available():
        raise KalmanError("no HIP device visible; the MI355X path has no CPU fallback")
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _engines:
        _engines[idx] = KalmanEngine(torch.device("cuda", idx))
    return _engines[idx]
