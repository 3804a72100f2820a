"""Host-side mirror of ``pymc_statespace.filters`` backed by the MI355X kernels.

Same class names, same ``build_graph`` signatures and output order as the reference
(reference pymc_statespace/filters/__init__.py:1-17, kalman_filter.py:126-163,
kalman_smoother.py:56-85), so the classes slot into ``FILTER_FACTORY``
(core/statespace.py:25-31).  Two ways in:

* ``flt.build_graph(data, a0, P0, T, Z, R, H, Q, c=None, d=None, mode=None)`` -- symbolic,
  needs PyTensor; returns the reference's six TensorVariables, computed by one Op whose
  ``perform``/``grad`` call the C-ABI (see ``op.py``).
* ``flt(data, a0, P0, T, Z, R, H, Q, c=None, d=None)`` -- eager.  numpy in -> numpy out,
  torch(HBM) in -> torch out.  Per-draw (2-D) matrices give the reference's shapes; matrices
  with one extra LEADING axis B are a batch of draws and give B-leading outputs.

Reference behaviours kept: ``ValueError`` for matrices that are not 2-D/3-D
(filters/utilities.py:16-18), ``AssertionError`` for SingleTimeseriesFilter with p > 1
(kalman_filter.py:19,:329), ``numpy.linalg.LinAlgError`` when F is not positive definite
(what SciPy's ``dposv``/``dpotrf`` raise inside the reference graph).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from .engine import KalmanEngine, get_engine

MVN_CONST = float(np.log(2.0 * np.pi))
PARAM_NAMES = ["c", "d", "T", "Z", "R", "H", "Q"]  # reference kalman_filter.py:17


def _is_torch(*xs):
    return any(isinstance(x, torch.Tensor) for x in xs if x is not None)


def _np_or_t(x):
    return x if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64)


def _check_ndim(params, names, batched):
    """split_vars_into_seq_and_nonseq, reference filters/utilities.py:1-20: per draw a 2-D
    matrix is static, a 3-D one is time varying (time on axis 0).  With ``batched`` the draw
    axis leads.  Returns the names of the time-varying matrices."""
    seq = []
    for prm, name in zip(params, names):
        nd = prm.ndim - (1 if batched else 0)
        if nd == 2:
            continue
        if nd == 3:
            seq.append(name)
            continue
        raise ValueError(
            f"Matrix {name} has {prm.ndim}, it should either 2 (static) or 3 (time varying).")
    return seq


class BaseFilter:
    filter_type: str = ""

    def __init__(self, mode=None, mvn_const: str = "reference", on_error: str = "raise",
                 engine: Optional[KalmanEngine] = None, h_diagonal: bool = False):
        self.mode = mode
        self.mvn_const = mvn_const
        self.h_diagonal = h_diagonal  # H diagonal by construction: see include/kf.h KF_FLAG_H_DIAGONAL
        self.on_error = on_error
        self._engine = engine
        self.seq_names: List[str] = []
        self.non_seq_names: List[str] = list(PARAM_NAMES)

    @property
    def engine(self) -> KalmanEngine:
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    # -- symbolic (PyTensor) ---------------------------------------------------------------
    def build_graph(self, data, a0, P0, T, Z, R, H, Q, c=None, d=None, mode=None, outputs="all", batched=False):
        """Reference signature (kalman_filter.py:126-128).  Returns
        [filtered_states, predicted_states, filtered_covariances, predicted_covariances,
         log_likelihood, ll_obs]; the node computes and copies back only the outputs the compiled graph
        consumes.  ``outputs="logp"`` (extension) returns the log-likelihood alone from the fused
        value-and-gradient Op (see op.py) -- what a model that registers no tape Deterministics needs.
        ``batched=True`` (extension): a0 .. Q (and c, d) carry ONE leading axis of B parameter draws -- the graph
        ``pytensor.graph.replace.vectorize_graph`` makes of the per-draw one, built directly; every output gains the leading
        axis and ``outputs="logp"`` is logp (B,) whose gradient costs ONE kf_filter_value_and_grad call for all draws."""
        from .op import kalman_filter_batched_graph, kalman_filter_graph  # imports pytensor; fails loudly without it

        self.mode = mode
        build = kalman_filter_batched_graph if batched else kalman_filter_graph
        return build(data, a0, P0, T, Z, R, H, Q, c=c, d=d, filter_type=self.filter_type, mvn_const=self.mvn_const,
                     h_diagonal=self.h_diagonal, outputs=outputs)

    def update(self, *args, **kwargs):
        # reference BaseFilter.update raises (kalman_filter.py:225-229); the fused kernels
        # have no separately callable update step.
        raise NotImplementedError

    # -- eager ---------------------------------------------------------------------------------
    def _forward(self, data, a0, P0, T, Z, R, H, Q, c, d, outputs):
        """The eager call's common part: (FilterResult in HBM, finisher, the batch-leading inputs as handed to the engine)."""
        as_torch = _is_torch(data, a0, P0, T, Z, R, H, Q)
        T_, Z_, R_, H_, Q_ = map(_np_or_t, (T, Z, R, H, Q))
        batched = P0 is not None and _np_or_t(P0).ndim == 3
        seq = _check_ndim([T_, Z_, R_, H_, Q_], ["T", "Z", "R", "H", "Q"], batched)
        self.seq_names = seq
        self.non_seq_names = [nm for nm in PARAM_NAMES if nm not in seq]

        def lead(x, core_ndim):
            x = _np_or_t(x)
            if x.ndim > core_ndim + (1 if batched else 0) and x.shape[-1] == 1 and core_ndim == 1:
                x = x[..., 0]  # (m,1) column vectors of the reference
            return x if batched else x[None]

        data_ = _np_or_t(data)
        p = Z_.shape[-2]
        if data_.ndim == 3 and data_.shape[-1] == 1 and data_.shape[-2] == p:
            data_ = data_[..., 0]  # reference data layout (n,p,1)
        if data_.ndim == 2 and data_.shape[1] != p:
            raise ValueError(f"data has {data_.shape[1]} columns but Z has {p} rows")

        eng = self.engine
        args = dict(data=data_, a0=lead(a0, 1), P0=lead(P0, 2), T=lead(T_, 2), Z=lead(Z_, 2), R=lead(R_, 2), H=lead(H_, 2),
                    Q=lead(Q_, 2), c=None if c is None else lead(c, 1), d=None if d is None else lead(d, 1))
        res = eng.forward(
            args["data"], args["a0"], args["P0"], args["T"], args["Z"], args["R"], args["H"], args["Q"], args["c"], args["d"],
            filter_type=self.filter_type, mvn_const=self.mvn_const,
            store_ll_obs=outputs == "all", store_filtered=outputs == "all",
            store_predicted=outputs == "all", time_varying=seq,
            h_diagonal=True if self.h_diagonal else None)
        info = res.info.cpu().numpy()
        if self.on_error == "raise" and np.any(info != 0):
            bad = int(np.flatnonzero(info)[0])
            raise np.linalg.LinAlgError(
                f"draw {bad}: {info[bad] & 0xff}-th leading minor of the innovation covariance F "
                f"not positive definite at time step {(info[bad] >> 8) - 1}")

        def fin(x, col=False):
            if col:
                x = x[..., None]
            if not batched:
                x = x[0]
            return x if as_torch else x.cpu().numpy()

        return res, fin, args

    def __call__(self, data, a0, P0, T, Z, R, H, Q, c=None, d=None, *, outputs="all"):
        """Evaluate on the GPU.  Returns the reference's list
        [filtered_states, predicted_states, filtered_covs, predicted_covs, logp, ll_obs]
        (or only ``[logp]`` for ``outputs="logp"``)."""
        res, fin, _ = self._forward(data, a0, P0, T, Z, R, H, Q, c, d, outputs)
        if outputs != "all":
            return [fin(res.logp)]
        return [fin(res.filtered_states, True), fin(res.predicted_states, True),
                fin(res.filtered_covs), fin(res.predicted_covs), fin(res.logp), fin(res.ll_obs)]

    def filter_and_smooth(self, data, a0, P0, T, Z, R, H, Q, c=None, d=None, *, route="auto"):
        """Filter AND smoother of one model in one eager call -- build_statespace_graph + build_smoother_graph
        (core/statespace.py:164-197): the reference's six filter outputs followed by (smoothed_states, smoothed_covariances).
        The tapes stay in HBM between the two kernels, and because the filter's own result is at hand the smoother takes the
        recursion on the PREDICTED tapes wherever it is the cheaper one (``engine.smoother_from_filter_applies``: p == 1, static
        matrices, zero c -- no pinv(P_hat), every draw at the same cost; an ARMA model's smoothed covariances then agree with
        statsmodels to 1e-9 instead of the pinv recursion's 1e-2), else ``kf_smoother`` on the filtered ones.
        ``route``: "auto" | "from_filter" | "rts" (``KalmanEngine.smooth_result``)."""
        res, fin, a = self._forward(data, a0, P0, T, Z, R, H, Q, c, d, "all")
        sa, sP = self.engine.smooth_result(res, a["data"], a["T"], a["Z"], a["R"], a["H"], a["Q"], a["c"], a["d"],
                                           filter_type=self.filter_type, time_varying=tuple(self.seq_names), route=route)
        return [fin(res.filtered_states, True), fin(res.predicted_states, True), fin(res.filtered_covs), fin(res.predicted_covs),
                fin(res.logp), fin(res.ll_obs), fin(sa, True), fin(sP)]


class StandardFilter(BaseFilter):
    """reference kalman_filter.py:255-284"""
    filter_type = "standard"


class CholeskyFilter(BaseFilter):
    """reference kalman_filter.py:287-318 (the L^-T solve is the mathematically correct one;
    the reference's wrong-triangle solve for p > 1 is NOT reproduced -- SURVEY.md A8)"""
    filter_type = "cholesky"


class SingleTimeseriesFilter(BaseFilter):
    """reference kalman_filter.py:321-351"""
    filter_type = "single"


class UnivariateFilter(BaseFilter):
    """reference kalman_filter.py:444-505"""
    filter_type = "univariate"


class SteadyStateFilter(BaseFilter):
    """reference kalman_filter.py:354-441: the covariance recursion is started from the solution
    of the filtering Riccati equation (``solve_discrete_are(T', Z', R Q R', H)``, :388) and the
    gain uses the frozen ``F_inv`` of that fixed point.  Here: ``kf_dare`` (batched fixed-point
    iteration on the GPU) followed by the Standard recursion from ``P_steady`` -- at the fixed
    point every step's ``F^-1`` IS the frozen one, so the outputs agree to the DARE tolerance.
    Kept quirks: slot 0 of ``predicted_covariances`` is the caller's ``P0`` (:397,
    ``_postprocess_scan_results(results, a0, P0)``), and the innovation ignores ``d`` (:400).
    With missing observations the reference keeps the frozen gain although P has left the fixed
    point (its recursion can then diverge); this filter uses the exact gain of the current P there.
    Static system matrices only (:380-383).  ``build_graph`` returns the same six outputs from one PyTensor node
    whose gradient includes the Riccati solve (``kf_dare_backward``: utils/pytensor_scipy.py:33-60);
    ``logp_and_grad`` is the eager form of that value-and-gradient."""
    filter_type = "standard"

    def build_graph(self, data, a0, P0, T, Z, R, H, Q, c=None, d=None, mode=None, outputs="all"):
        """Reference signature (kalman_filter.py:364)."""
        from .op import steady_state_filter_graph  # imports pytensor; fails loudly without it

        self.mode = mode
        return steady_state_filter_graph(data, a0, P0, T, Z, R, H, Q, c=c, d=d, mvn_const=self.mvn_const,
                                         outputs=outputs)

    def logp_and_grad(self, data, a0, P0, T, Z, R, H, Q, c=None, d=None):
        """(logp, dict of gradients w.r.t. a0, P0, T, Z, R, H, Q, c, d) for one draw in the reference layouts --
        what the compiled graph's value-and-gradient evaluates."""
        from .op import GRAD_INPUT_NAMES, perform_steady_state_grad

        m, p = np.asarray(T).shape[-1], np.asarray(Z).shape[-2]
        outs = perform_steady_state_grad(data, a0, P0, T, Z, R, H, Q, np.zeros((m, 1)) if c is None else c,
                                         np.zeros((p, 1)) if d is None else d, mvn_const=self.mvn_const,
                                         engine=self.engine)
        return float(outs[0]), dict(zip(GRAD_INPUT_NAMES, outs[1:]))

    def __call__(self, data, a0, P0, T, Z, R, H, Q, c=None, d=None, *, outputs="all",
                 max_iter=100000, tol=1e-14):
        as_torch = _is_torch(data, a0, P0, T, Z, R, H, Q)
        T_, Z_, R_, H_, Q_ = map(_np_or_t, (T, Z, R, H, Q))
        P0_ = _np_or_t(P0)
        batched = P0_.ndim == 3
        if _check_ndim([T_, Z_, R_, H_, Q_], ["T", "Z", "R", "H", "Q"], batched):
            raise ValueError("All system matrices must be time-invariant to use the SteadyStateFilter")

        def lead(x):
            return x if batched else x[None]

        P_steady, info = self.engine.dare(lead(T_), lead(Z_), lead(R_), lead(H_), lead(Q_),
                                          max_iter=max_iter, tol=tol)
        bad = info.cpu().numpy()
        if self.on_error == "raise" and np.any(bad != 0):
            b = int(np.flatnonzero(bad)[0])
            raise np.linalg.LinAlgError(
                f"draw {b}: discrete Riccati iteration " +
                ("did not converge" if bad[b] == 1 else "met a non positive definite Z P Z' + H"))
        if not batched:
            P_steady = P_steady[0]
        if not as_torch:
            P_steady = P_steady.cpu().numpy()
        res = super().__call__(data, a0, P_steady, T, Z, R, H, Q, c=c, d=None, outputs=outputs)
        if outputs == "all":
            pc = res[3]
            if isinstance(pc, torch.Tensor):
                P0t = P0_ if isinstance(P0_, torch.Tensor) else torch.as_tensor(P0_, device=pc.device)
                if batched:
                    pc[:, 0] = P0t.to(pc.device)
                else:
                    pc[0] = P0t.to(pc.device)
            else:
                P0n = P0_.cpu().numpy() if isinstance(P0_, torch.Tensor) else P0_
                if batched:
                    pc[:, 0] = P0n
                else:
                    pc[0] = P0n
        return res


class KalmanSmoother:
    """reference kalman_smoother.py:11-104"""

    def __init__(self, mode: Optional[str] = None, engine: Optional[KalmanEngine] = None):
        self.mode = mode
        self._engine = engine
        self.seq_names: List[str] = []
        self.non_seq_names: List[str] = ["T", "R", "Q"]

    @property
    def engine(self) -> KalmanEngine:
        if self._engine is None:
            self._engine = get_engine()
        return self._engine

    def build_graph(self, T, R, Q, filtered_states, filtered_covariances, mode=None):
        from .op import kalman_smoother_graph

        self.mode = mode
        return kalman_smoother_graph(T, R, Q, filtered_states, filtered_covariances)

    def __call__(self, T, R, Q, filtered_states, filtered_covariances, *, filter_result=None, data=None, Z=None, H=None,
                 c=None, d=None, filter_type="standard", route="auto"):
        """Reference signature (kalman_smoother.py:56) plus, as keywords, what lets the smoother use the filter's own result
        (extension): ``filter_result`` = the list a filter's eager call returned ([filtered_states, predicted_states, filtered_covs,
        predicted_covs, logp, ll_obs]) together with that model's ``data``, ``Z``, ``H`` (and ``c``, ``d`` when it had intercepts;
        ``filter_type="single"`` for SingleTimeseriesFilter).  With them the smoothed moments come from the PREDICTED tapes wherever
        that recursion is the cheaper one (``engine.smoother_from_filter_applies``; zero c): no pinv(P_hat) -- for ARMA / SARIMA models
        (k_posdef = 1) the difference is 20x - 1000x (include/kf.h) and smoothed covariances that agree with statsmodels to 1e-9
        instead of 1e-2.  Without them: the reference's recursion on the filtered tapes, as before."""
        as_torch = _is_torch(T, R, Q, filtered_states, filtered_covariances)
        T_, R_, Q_ = map(_np_or_t, (T, R, Q))
        fP = _np_or_t(filtered_covariances)
        fa = _np_or_t(filtered_states)
        batched = fP.ndim == 4
        # per draw a 3-D matrix is time varying, time on axis 0 (filters/utilities.py:1-20); the step that smooths t
        # uses slice t, the forward filter's pairing (the reference's scan pairs it with t + 1: see oracle/kalman_numpy.py)
        self.seq_names = _check_ndim([T_, R_, Q_], ["T", "R", "Q"], batched)
        self.non_seq_names = [nm for nm in ("T", "R", "Q") if nm not in self.seq_names]
        m = T_.shape[-1]

        if fa.shape[-1] == 1 and fa.ndim >= 2 and fa.shape[-2] == m:
            fa = fa[..., 0]
        lead = (lambda x: x) if batched else (lambda x: x[None])
        eng = self.engine
        if filter_result is not None and not self.seq_names:
            if data is None or Z is None or H is None:
                raise ValueError("filter_result needs the filter's data, Z and H as well (and c, d when the model had intercepts)")
            from .engine import FilterResult

            ps, pc = _np_or_t(filter_result[1]), _np_or_t(filter_result[3])
            if ps.shape[-1] == 1 and ps.shape[-2] == m:
                ps = ps[..., 0]
            Z_, H_ = _np_or_t(Z), _np_or_t(H)
            p = Z_.shape[-2]
            data_ = _np_or_t(data)
            if data_.ndim == 4 or (data_.ndim == 3 and data_.shape[-1] == 1 and data_.shape[-2] == p):
                data_ = data_[..., 0]  # reference data layout (n,p,1), as in the filters' eager call

            def vec(x, k):   # (k,1) / (B,k,1) column or (k,) / (B,k) vector -> (B,k)
                if x is None:
                    return None
                x = _np_or_t(x)
                if x.shape[-1] == 1 and x.ndim == (3 if batched else 2):
                    x = x[..., 0]
                return lead(x)

            res = FilterResult(logp=None, ll_obs=None, filtered_states=eng.to_device(lead(fa)),
                               predicted_states=eng.to_device(lead(ps)), filtered_covs=eng.to_device(lead(fP)),
                               predicted_covs=eng.to_device(lead(pc)), info=None)
            sa, sP = eng.smooth_result(res, data_, lead(T_), lead(Z_), lead(R_), lead(H_), lead(Q_), vec(c, m), vec(d, p),
                                       filter_type=filter_type, route=route)
        else:
            sa, sP = eng.smooth(lead(T_), lead(R_), lead(Q_), lead(fa), lead(fP), time_varying=tuple(self.seq_names))
        sa = sa[..., None]
        if not batched:
            sa, sP = sa[0], sP[0]
        if as_torch:
            return sa, sP
        return sa.cpu().numpy(), sP.cpu().numpy()


# the reference's FILTER_FACTORY keys (core/statespace.py:25-31)
FILTER_FACTORY = {
    "standard": StandardFilter,
    "steady_state": SteadyStateFilter,
    "univariate": UnivariateFilter,
    "single": SingleTimeseriesFilter,
    "cholesky": CholeskyFilter,
}

__all__ = ["StandardFilter", "UnivariateFilter", "KalmanSmoother", "SingleTimeseriesFilter",
           "CholeskyFilter", "SteadyStateFilter", "FILTER_FACTORY", "BaseFilter"]
