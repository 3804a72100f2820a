"""PyTensor adapter: the Op that lets the HIP kernels stand in for the reference's scan inside a
PyMC model (``PyMCStateSpace.build_statespace_graph()``, reference core/statespace.py:146-174).

Pattern follows the reference's only custom Op, ``SolveDiscreteARE``
(reference utils/pytensor_scipy.py:11-60): ``__props__``, ``make_node`` (``as_tensor_variable``,
one ``Apply``), ``perform(node, inputs, output_storage)``, ``infer_shape``, a gradient.  The
gradient is a second Op whose ``perform`` calls ``kf_filter_backward``.

PyTensor / PyMC are not installed in the build image, so the Op classes are created lazily and
everything numeric lives in the plain functions ``perform_filter`` / ``perform_filter_grad`` /
``perform_logp_and_grad`` / ``perform_filter_batched`` / ``perform_smoother`` (exercised by the GPU
tests); the Op methods only move arrays in and out of them.  ``tests/test_pytensor_op.py`` is the
executable contract for an environment that has PyTensor (it skips here).

What crosses PCIe.  ``perform`` receives host numpy arrays and must return host numpy arrays, so every
output the graph consumes is copied back; the covariance tapes are (2n + 1) m^2 doubles per draw.
  * ``KalmanFilterOp`` carries a ``want`` mask chosen when the graph is built (``build_graph(..., outputs=(names))``,
    default all six as the reference returns them) and computes / copies ONLY those (``store_*`` flags of
    ``kf_filter_forward``); an output that is not wanted is a zero-size placeholder.
  * ``KalmanLogpOp`` -- one output, ``logp``, plus the gradients as extra outputs of the SAME node --
    is what ``build_graph(..., outputs="logp")`` returns: forward (predicted tapes stay in HBM) and
    adjoint run back to back on the device, 8 + (2 m^2 + ...) * 8 bytes come back, and ``L_op`` reuses
    the node's own gradient outputs, so NUTS evaluates value and gradient with one ``perform``.
  * ``KalmanFilterBatchedOp`` takes a leading draw axis B on every system matrix (data shared) and is
    what ``pytensor.graph.replace.vectorize_graph`` maps ``KalmanFilterOp`` to: posterior-predictive
    style evaluation over draws reaches the batched kernels through ``build_statespace_graph()``.
"""
from __future__ import annotations

import numpy as np

from .engine import FilterResult, get_engine, smoother_from_filter_applies  # noqa: F401  (the rule: re-exported)

GRAD_INPUT_NAMES = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")


# --------------------------------------------------------------------------------------------
# numeric bodies (no PyTensor needed)
# --------------------------------------------------------------------------------------------
def _out(x):
    """What an Op stores in output_storage: a float64 C-contiguous array of the output's OWN rank (np.ascontiguousarray
    would turn the 0-d log-likelihood into shape (1,), which a TensorType(float64, ()) output rejects)."""
    a = np.asarray(x, dtype=np.float64)
    return a if a.ndim == 0 else np.ascontiguousarray(a)


def _col(x):
    x = np.asarray(x, dtype=np.float64)
    return x[:, 0] if x.ndim == 2 and x.shape[1] == 1 else x


_TV_ORDER = ("T", "Z", "R", "H", "Q", "c", "d")


def _lead_system(T, Z, R, H, Q, c, d):
    """One draw's system matrices in the reference layouts -> (arrays with a leading draw axis, time_varying names).
    A matrix with one more axis than its static form (T (n,m,m); c (n,m,1)) is time-varying: slice t at step t
    (reference filters/utilities.py:1-20, kalman_filter.py:231-252)."""
    out, tv = [], []
    for name, x in zip(_TV_ORDER, (T, Z, R, H, Q, c, d)):
        x = np.asarray(x, dtype=np.float64)
        if name in ("c", "d"):
            if x.ndim == 3 or (x.ndim == 2 and x.shape[-1] != 1):
                tv.append(name)
                x = x[..., 0] if x.ndim == 3 else x
            else:
                x = _col(x)
        elif x.ndim == 3:
            tv.append(name)
        out.append(x[None])
    return out, tuple(tv)


def perform_filter(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard",
                   mvn_const="reference", h_diagonal=False, engine=None, want=(True,) * 6):
    """One draw, reference layouts in (data (n,p,1), a0 (m,1), c (m,1), d (p,1)) -> the six
    reference outputs [filtered_states (n,m,1), predicted_states (n+1,m,1), filtered_covs,
    predicted_covs, logp (), ll_obs (n,)] as numpy arrays.  ``want[i]`` False: output i is neither
    computed nor copied back (None in its place); a tape pair is produced when either half is wanted."""
    eng = engine or get_engine()
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 3:
        data = data[..., 0]
    sysm, tv = _lead_system(T, Z, R, H, Q, c, d)
    res = eng.forward(data, _col(a0)[None], np.asarray(P0)[None], *sysm,
                      filter_type=filter_type, mvn_const=mvn_const, time_varying=tv,
                      h_diagonal=bool(h_diagonal),  # never auto-detected here: that costs a host sync per call
                      store_filtered=bool(want[0] or want[2]), store_predicted=bool(want[1] or want[3]),
                      store_ll_obs=bool(want[5]))
    _raise_if_not_pd(res.info)
    return [
        res.filtered_states[0].cpu().numpy()[..., None] if want[0] else None,
        res.predicted_states[0].cpu().numpy()[..., None] if want[1] else None,
        res.filtered_covs[0].cpu().numpy() if want[2] else None,
        res.predicted_covs[0].cpu().numpy() if want[3] else None,
        np.asarray(res.logp[0].item(), dtype=np.float64),
        res.ll_obs[0].cpu().numpy() if want[5] else None,
    ]


def _raise_if_not_pd(info):
    bad = info.nonzero()
    if bad.numel():
        b = int(bad[0].item())
        code = int(info[b].item())
        raise np.linalg.LinAlgError(
            f"{code & 0xff}-th leading minor of the innovation covariance not positive definite "
            f"at time step {(code >> 8) - 1}" + (f" (draw {b})" if info.numel() > 1 else ""))


def perform_logp_and_grad(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard",
                          mvn_const="reference", h_diagonal=False, engine=None):
    """logp () and d logp / d (a0, P0, T, Z, R, H, Q, c, d) of one draw in ONE device round trip: the forward
    filter keeps its predicted tapes in HBM, the adjoint kernel consumes them there; only the scalar and
    the nine small gradient arrays are copied back (what a NUTS leapfrog step needs)."""
    eng = engine or get_engine()
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 3:
        data = data[..., 0]
    sysm, tv = _lead_system(T, Z, R, H, Q, c, d)
    args = (data, _col(a0)[None], np.asarray(P0)[None], *sysm)
    dev = [eng.to_device(x) for x in args]
    if not tv:  # one C-ABI call: the predicted tapes never leave the workspace (kf_filter_value_and_grad)
        logp, info, g = eng.value_and_grad(*dev, filter_type=filter_type, mvn_const=mvn_const,
                                           h_diagonal=bool(h_diagonal))
        res = FilterResult(logp=logp, ll_obs=None, filtered_states=None, predicted_states=None, filtered_covs=None,
                           predicted_covs=None, info=info)
    else:
        res = eng.forward(*dev, filter_type=filter_type, mvn_const=mvn_const, h_diagonal=bool(h_diagonal),
                          store_filtered=False, store_ll_obs=False, time_varying=tv)
        g = eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                         g_logp=np.ones(1), filter_type=filter_type, mvn_const=mvn_const, want_cd=True,
                         h_diagonal=bool(h_diagonal), time_varying=tv)
    _raise_if_not_pd(res.info)
    out = {k: g[k][0].cpu().numpy() for k in GRAD_INPUT_NAMES}
    for k in ("a0", "c", "d"):
        out[k] = out[k][..., None]  # columns, as the inputs ((m,1); time-varying (n,m,1))
    return [np.asarray(res.logp[0].item(), dtype=np.float64)] + [out[k] for k in GRAD_INPUT_NAMES]


def _lead_system_batched(T, Z, R, H, Q, c, d):
    """The system of a batch of draws -> (arrays as ``engine.forward`` takes them, names of the time-varying ones).  ONE leading draw
    axis on everything; behind it a matrix is (rows, cols) when static and (n, rows, cols) when time-varying (the reference's 3-D
    convention per draw, filters/utilities.py:1-20), c / d are (m,1) / (p,1) columns or (n,m,1) / (n,p,1) -- round 6: what
    ``vectorize_graph`` makes of a per-draw node with time-varying inputs."""
    out, tv = [], []
    for name, x in zip(_TV_ORDER, (T, Z, R, H, Q, c, d)):
        x = np.asarray(x, dtype=np.float64)
        if name in ("c", "d"):
            if x.ndim >= 3 and x.shape[-1] == 1:
                x = x[..., 0]          # columns: (B,m,1) -> (B,m); (B,n,m,1) -> (B,n,m)
            if x.ndim == 3:
                tv.append(name)
            elif x.ndim != 2:
                raise ValueError(f"{name}: expected (B,k[,1]) or (B,n,k[,1]), got {x.shape}")
        elif x.ndim == 4:
            tv.append(name)
        elif x.ndim != 3:
            raise ValueError(f"Matrix {name} of a batch has {x.ndim} dimensions: 3 (draws, rows, columns) or 4 (draws, time, rows, columns).")
        out.append(x)
    return out, tuple(tv)


def perform_filter_batched(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard",
                           mvn_const="reference", h_diagonal=False, engine=None, want=(True,) * 6):
    """B draws at once: data (n,p,1) shared, every other input with a leading axis B in the reference
    layouts (a0 (B,m,1), ...) -> the six reference outputs with a leading B (logp (B,), ll_obs (B,n))."""
    eng = engine or get_engine()
    data = np.asarray(data, dtype=np.float64)
    if data.ndim in (3, 4):  # (n,p,1) shared by all draws, or (B,n,p,1): one series per draw (KF_FLAG_Y_PER_DRAW)
        data = data[..., 0]

    def vec(x):
        x = np.asarray(x, dtype=np.float64)
        return x[..., 0] if x.ndim == 3 and x.shape[-1] == 1 else x

    sysm, tv = _lead_system_batched(T, Z, R, H, Q, c, d)
    res = eng.forward(data, vec(a0), np.asarray(P0), *sysm, filter_type=filter_type,
                      mvn_const=mvn_const, h_diagonal=bool(h_diagonal), time_varying=tv,
                      store_filtered=bool(want[0] or want[2]), store_predicted=bool(want[1] or want[3]),
                      store_ll_obs=bool(want[5]))
    _raise_if_not_pd(res.info)
    return [
        res.filtered_states.cpu().numpy()[..., None] if want[0] else None,
        res.predicted_states.cpu().numpy()[..., None] if want[1] else None,
        res.filtered_covs.cpu().numpy() if want[2] else None,
        res.predicted_covs.cpu().numpy() if want[3] else None,
        res.logp.cpu().numpy(),
        res.ll_obs.cpu().numpy() if want[5] else None,
    ]


def perform_logp_and_grad_batched(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard",
                                  mvn_const="reference", h_diagonal=False, engine=None):
    """B draws, value AND gradient, in ONE ``kf_filter_value_and_grad`` call -- the sampler's call over a batch of draws
    (vectorised chains, SMC particles, ``vectorize_graph`` over the model's logp; the loop this replaces is the per-draw Python
    loop of reference core/statespace.py:230-236 wrapped around the Potential of :174).
    data (n,p,1) shared by all draws -- or (B,n,p,1), one series per draw -- and every other input with a leading axis B in the
    reference layouts (a0 (B,m,1), P0 (B,m,m), ..., c (B,m,1), d (B,p,1)) -> [logp (B,), g_a0 (B,m,1), g_P0, g_T, g_Z, g_R,
    g_H, g_Q, g_c (B,m,1), g_d (B,p,1)]: row b of every gradient is d logp[b] / d (that input's row b)."""
    eng = engine or get_engine()
    data = np.asarray(data, dtype=np.float64)
    if data.ndim in (3, 4):
        data = data[..., 0]

    def vec(x):
        x = np.asarray(x, dtype=np.float64)
        return x[..., 0] if x.ndim == 3 and x.shape[-1] == 1 else x

    sysm, tv = _lead_system_batched(T, Z, R, H, Q, c, d)
    if not tv:   # one C-ABI call: the predicted tapes never leave the workspace (kf_filter_value_and_grad)
        logp, info, g = eng.value_and_grad(data, vec(a0), np.asarray(P0, dtype=np.float64), *sysm, filter_type=filter_type,
                                           mvn_const=mvn_const, h_diagonal=bool(h_diagonal))
    else:        # time-varying matrices (round 6): forward with the predicted tapes in HBM, then the adjoint sweep -- two calls for all draws
        dev = [eng.to_device(x) for x in (data, vec(a0), np.asarray(P0, dtype=np.float64), *sysm)]
        res = eng.forward(*dev, filter_type=filter_type, mvn_const=mvn_const, h_diagonal=bool(h_diagonal),
                          store_filtered=False, store_ll_obs=False, time_varying=tv)
        g = eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                         g_logp=np.ones(dev[2].shape[0]), filter_type=filter_type, mvn_const=mvn_const, want_cd=True,
                         h_diagonal=bool(h_diagonal), time_varying=tv)
        logp, info = res.logp, res.info
    _raise_if_not_pd(info)
    out = {k: g[k].cpu().numpy() for k in GRAD_INPUT_NAMES}
    for k in ("a0", "c", "d"):
        out[k] = out[k][..., None]
    return [logp.cpu().numpy()] + [out[k] for k in GRAD_INPUT_NAMES]


def _tape_cotangent(g):
    """A tape cotangent from the graph, or None for the zero-size placeholder that stands for "no cotangent on this output"."""
    if g is None:
        return None
    g = np.asarray(g, dtype=np.float64)
    return g if g.size else None


def _state_cotangent(g, batched):
    """(n, m, 1) / (B, n, m, 1) columns of the graph -> the C-ABI's [B][n][m]."""
    g = _tape_cotangent(g)
    if g is None:
        return None
    if g.ndim == (4 if batched else 3):
        g = g[..., 0]
    return g if batched else g[None]


def _cov_cotangent(g, batched):
    g = _tape_cotangent(g)
    if g is None:
        return None
    return g if batched else g[None]


def perform_filter_grad(data, a0, P0, T, Z, R, H, Q, c, d, predicted_states, predicted_covs,
                        g_logp, g_ll_obs, g_filtered_states=None, g_predicted_states=None, g_filtered_covs=None,
                        g_predicted_covs=None, filter_type="standard", mvn_const="reference",
                        h_diagonal=False, engine=None):
    """Gradients of g_logp*logp + <g_ll_obs, ll_obs> (+ <cotangent, tape> for the four tape outputs: round 6,
    kf_filter_backward_tapes) w.r.t. (a0,P0,T,Z,R,H,Q,c,d), shaped like the inputs (a0 (m,1), c (m,1), d (p,1)).
    A zero-size tape cotangent means none; zero-size predicted tapes (the node was built without them) are recomputed here."""
    eng = engine or get_engine()
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 3:
        data = data[..., 0]
    # (host arrays here: PyTensor hands the forward node's outputs back through the graph; the fused
    # KalmanLogpOp exists so that the sampler's value-and-gradient path never does this)
    pa = np.asarray(predicted_states, dtype=np.float64)
    if pa.ndim == 3 and pa.size:
        pa = pa[..., 0]
    sysm, tv = _lead_system(T, Z, R, H, Q, c, d)
    pP = np.asarray(predicted_covs, dtype=np.float64)
    if pa.size == 0 or pP.size == 0:
        res = eng.forward(data, _col(a0)[None], np.asarray(P0)[None], *sysm, filter_type=filter_type, mvn_const=mvn_const,
                          store_ll_obs=False, store_filtered=False, time_varying=tv)
        pa_b, pP_b = res.predicted_states, res.predicted_covs
    else:
        pa_b, pP_b = pa[None], pP[None]
    g = eng.backward(data, _col(a0)[None], np.asarray(P0)[None], *sysm,
                     predicted_states=pa_b, predicted_covs=pP_b,
                     g_logp=np.atleast_1d(np.asarray(g_logp, dtype=np.float64)),
                     g_ll_obs=np.asarray(g_ll_obs, dtype=np.float64)[None],
                     g_filtered_states=_state_cotangent(g_filtered_states, False),
                     g_predicted_states=_state_cotangent(g_predicted_states, False),
                     g_filtered_covs=_cov_cotangent(g_filtered_covs, False),
                     g_predicted_covs=_cov_cotangent(g_predicted_covs, False),
                     filter_type=filter_type, mvn_const=mvn_const, want_cd=True,
                     h_diagonal=h_diagonal, time_varying=tv)
    out = {k: g[k][0].cpu().numpy() for k in GRAD_INPUT_NAMES}
    for k in ("a0", "c", "d"):
        out[k] = out[k][..., None]
    return [out[k] for k in GRAD_INPUT_NAMES]


def perform_filter_grad_batched(data, a0, P0, T, Z, R, H, Q, c, d, predicted_states, predicted_covs,
                                g_logp, g_ll_obs, g_filtered_states=None, g_predicted_states=None, g_filtered_covs=None,
                                g_predicted_covs=None, filter_type="standard", mvn_const="reference",
                                h_diagonal=False, engine=None):
    """``perform_filter_grad`` with ONE leading draw axis on a0 .. d, the predicted tapes, g_logp (B,) and g_ll_obs (B,n): the
    gradients of sum_b (g_logp[b] logp[b] + <g_ll_obs[b], ll_obs[b]>) from ONE kf_filter_backward call, row b of each = draw b's.
    What a cotangent on the batched filter's ``ll_obs`` needs (a model that masks or sums ll_obs -- a burn-in -- after
    ``vectorize_graph``; the per-draw node differentiates it through KalmanFilterGradOp)."""
    eng = engine or get_engine()
    data = np.asarray(data, dtype=np.float64)
    if data.ndim in (3, 4):
        data = data[..., 0]

    def vec(x):
        x = np.asarray(x, dtype=np.float64)
        return x[..., 0] if x.ndim == 3 and x.shape[-1] == 1 else x

    sysm, tv = _lead_system_batched(T, Z, R, H, Q, c, d)
    pa = np.asarray(predicted_states, dtype=np.float64)
    if pa.ndim == 4 and pa.size:
        pa = pa[..., 0]
    pP = np.asarray(predicted_covs, dtype=np.float64)
    if pa.size == 0 or pP.size == 0:   # (the node was built without the predicted tapes: recomputed here)
        res = eng.forward(data, vec(a0), np.asarray(P0, dtype=np.float64), *sysm, filter_type=filter_type, mvn_const=mvn_const,
                          store_ll_obs=False, store_filtered=False, time_varying=tv)
        pa, pP = res.predicted_states, res.predicted_covs
    g = eng.backward(data, vec(a0), np.asarray(P0, dtype=np.float64), *sysm, predicted_states=pa,
                     predicted_covs=pP,
                     g_logp=np.asarray(g_logp, dtype=np.float64), g_ll_obs=np.asarray(g_ll_obs, dtype=np.float64),
                     g_filtered_states=_state_cotangent(g_filtered_states, True),
                     g_predicted_states=_state_cotangent(g_predicted_states, True),
                     g_filtered_covs=_cov_cotangent(g_filtered_covs, True),
                     g_predicted_covs=_cov_cotangent(g_predicted_covs, True),
                     filter_type=filter_type, mvn_const=mvn_const, want_cd=True, h_diagonal=h_diagonal, time_varying=tv)
    out = {k: g[k].cpu().numpy() for k in GRAD_INPUT_NAMES}
    for k in ("a0", "c", "d"):
        out[k] = out[k][..., None]
    return [out[k] for k in GRAD_INPUT_NAMES]


def _steady_inputs(data, a0, P0, T, Z, R, H, Q, c):
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 3:
        data = data[..., 0]
    mats = [np.asarray(x, dtype=np.float64) for x in (T, Z, R, H, Q)]
    if any(x.ndim != 2 for x in mats):
        raise ValueError("All system matrices must be time-invariant to use the SteadyStateFilter")
    return data, _col(a0)[None], np.asarray(P0, dtype=np.float64), [x[None] for x in mats], _col(c)[None]


def _steady_covariance(eng, mats, max_iter=100000, tol=1e-14):
    P, info = eng.dare(*mats, max_iter=max_iter, tol=tol)
    code = int(info[0].item())
    if code:
        raise np.linalg.LinAlgError("discrete Riccati iteration " + (
            "did not converge" if code == 1 else "met a non positive definite Z P Z' + H"))
    return P


def perform_steady_state(data, a0, P0, T, Z, R, H, Q, c, d, mvn_const="reference", engine=None,
                         want=(True,) * 6):
    """SteadyStateFilter (reference kalman_filter.py:354-441) for one draw: kf_dare, then the Standard recursion
    started from P_steady.  Kept quirks: slot 0 of predicted_covs is the caller's P0 (:397), d is ignored (:400)."""
    eng = engine or get_engine()
    data, a0_, P0_, mats, c_ = _steady_inputs(data, a0, P0, T, Z, R, H, Q, c)
    Ps = _steady_covariance(eng, mats)
    res = eng.forward(data, a0_, Ps, *mats, c_, None, filter_type="standard", mvn_const=mvn_const,
                      store_filtered=bool(want[0] or want[2]), store_predicted=bool(want[1] or want[3]),
                      store_ll_obs=bool(want[5]))
    _raise_if_not_pd(res.info)
    pc = None
    if want[3]:
        pc = res.predicted_covs[0].cpu().numpy()
        pc[0] = P0_
    return [
        res.filtered_states[0].cpu().numpy()[..., None] if want[0] else None,
        res.predicted_states[0].cpu().numpy()[..., None] if want[1] else None,
        res.filtered_covs[0].cpu().numpy() if want[2] else None,
        pc,
        np.asarray(res.logp[0].item(), dtype=np.float64),
        res.ll_obs[0].cpu().numpy() if want[5] else None,
    ]


def perform_steady_state_grad(data, a0, P0, T, Z, R, H, Q, c, d, g_logp=None, g_ll_obs=None,
                              mvn_const="reference", engine=None):
    """(logp, gradients of g_logp*logp + <g_ll_obs, ll_obs> w.r.t. (a0,P0,T,Z,R,H,Q,c,d)) of the SteadyStateFilter, all
    on the device: kf_dare -> kf_filter_forward from P_steady -> kf_filter_backward -> kf_dare_backward, which adds the
    path through the Riccati solve (g_P0 of the filter is the cotangent of P_steady; reference
    utils/pytensor_scipy.py:33-60).  P0 and d do not enter the likelihood: their gradients are zero."""
    eng = engine or get_engine()
    data, a0_, P0_, mats, c_ = _steady_inputs(data, a0, P0, T, Z, R, H, Q, c)
    dev = [eng.to_device(x) for x in mats]
    Ps = _steady_covariance(eng, dev)
    y, a0d, cd = eng.to_device(data), eng.to_device(a0_), eng.to_device(c_)
    res = eng.forward(y, a0d, Ps, *dev, cd, None, filter_type="standard", mvn_const=mvn_const,
                      store_filtered=False, store_ll_obs=False)
    _raise_if_not_pd(res.info)
    g = eng.backward(y, a0d, Ps, *dev, cd, None, predicted_states=res.predicted_states,
                     predicted_covs=res.predicted_covs,
                     g_logp=np.ones(1) if g_logp is None else np.atleast_1d(np.asarray(g_logp, dtype=np.float64)),
                     g_ll_obs=None if g_ll_obs is None else np.asarray(g_ll_obs, dtype=np.float64)[None],
                     filter_type="standard", mvn_const=mvn_const, want_cd=True)
    _, info = eng.dare_backward(*dev, Ps, g["P0"], into=g)
    if int(info[0].item()):
        raise np.linalg.LinAlgError("the Lyapunov solve of the Riccati gradient did not converge")
    out = {k: g[k][0].cpu().numpy() for k in GRAD_INPUT_NAMES}
    out["P0"] = np.zeros_like(P0_)
    out["d"] = np.zeros_like(_col(d))
    for k in ("a0", "c", "d"):
        out[k] = out[k][..., None]
    return [np.asarray(res.logp[0].item(), dtype=np.float64)] + [out[k] for k in GRAD_INPUT_NAMES]


def perform_smoother(T, R, Q, filtered_states, filtered_covs, engine=None):
    eng = engine or get_engine()
    fa = np.asarray(filtered_states, dtype=np.float64)
    if fa.ndim == 3:
        fa = fa[..., 0]
    tv = tuple(nm for nm, x in zip(("T", "R", "Q"), (T, R, Q)) if np.asarray(x).ndim == 3)
    sa, sP = eng.smooth(np.asarray(T)[None], np.asarray(R)[None], np.asarray(Q)[None], fa[None],
                        np.asarray(filtered_covs)[None], time_varying=tv)
    return [sa[0].cpu().numpy()[..., None], sP[0].cpu().numpy()]


def perform_smoother_batched(T, R, Q, filtered_states, filtered_covs, engine=None):
    """B draws from ONE ``kf_smoother`` call: T (B,m,m), R (B,m,r), Q (B,r,r), filtered_states (B,n,m,1), filtered_covs
    (B,n,m,m) -> [smoothed_states (B,n,m,1), smoothed_covs (B,n,m,m)].  The loop this replaces: one KalmanSmoother graph call
    per posterior draw (reference core/statespace.py:283-302 over :176-197)."""
    eng = engine or get_engine()
    T, R, Q = (np.asarray(x, dtype=np.float64) for x in (T, R, Q))
    if any(x.ndim not in (3, 4) for x in (T, R, Q)):
        raise ValueError("the batched smoother Op takes T, R, Q with ONE leading draw axis: (B,rows,cols), or (B,n,rows,cols) time-varying")
    tv = tuple(nm for nm, x in zip(("T", "R", "Q"), (T, R, Q)) if x.ndim == 4)   # (round 6: the vectorize_graph image of a time-varying node)
    fa = np.asarray(filtered_states, dtype=np.float64)
    if fa.ndim == 4:
        fa = fa[..., 0]
    sa, sP = eng.smooth(T, R, Q, fa, np.asarray(filtered_covs, dtype=np.float64), time_varying=tv)
    return [sa.cpu().numpy()[..., None], sP.cpu().numpy()]


def perform_smoother_fused(data, T, Z, R, H, Q, c, d, filtered_states, filtered_covs, predicted_states, predicted_covs,
                           filter_type="standard", engine=None, batched=False):
    """The smoother behind THIS library's filter node (``kalman_smoother_graph`` finds it: the reference never smooths anything
    but its own filter's outputs, core/statespace.py:164-197).  All of the filter's tapes are at hand, so the numeric body picks
    the cheaper recursion: where ``engine.smoother_from_filter_applies`` says so (p == 1, static matrices and a static d; m <= 15,
    64 < m <= 1024, or 16 <= m <= 64 with state noise of few columns) ``kf_smoother_from_filter`` on the PREDICTED tapes (no
    pinv(P_hat): include/kf.h), otherwise ``kf_smoother`` on the filtered ones (kalman_smoother.py:56-104 as it stands).
    A non-zero state intercept ``c`` keeps the reference's recursion too: the reference's smoother predicts with
    a_hat = T a_f, WITHOUT c (kalman_smoother.py:99-104), the filter's predicted tapes contain it, and the two recursions then
    differ by O(c) -- the node returns what the reference returns, whatever m is.
    Shapes: the per-draw reference layouts, or with ONE leading draw axis on everything but a shared data when ``batched``."""
    eng = engine or get_engine()
    lead = 1 if batched else 0
    Ta, Za, Ha = (np.asarray(x, dtype=np.float64) for x in (T, Z, H))
    p, m = Za.shape[-2], Ta.shape[-1]
    dv = np.asarray(d, dtype=np.float64)
    if dv.ndim >= 2 and dv.shape[-1] == 1:
        dv = dv[..., 0]  # (p,1) column, (n,p,1) when time-varying
    nds = [np.asarray(x).ndim - lead for x in (T, Z, H, R, Q)] + [dv.ndim - lead]
    cv = np.asarray(c, dtype=np.float64)
    if not smoother_from_filter_applies(p, m, nds, r=np.asarray(R).shape[-1]) or np.any(cv != 0.0):
        return (perform_smoother_batched if batched else perform_smoother)(T, R, Q, filtered_states, filtered_covs, engine=eng)
    y = np.asarray(data, dtype=np.float64)
    if y.ndim == 3 + lead or (batched and y.ndim == 3):  # (n,p,1) [shared] or (B,n,p,1)
        y = y[..., 0]
    pa = np.asarray(predicted_states, dtype=np.float64)
    if pa.ndim == 3 + lead:
        pa = pa[..., 0]
    pP = np.asarray(predicted_covs, dtype=np.float64)
    if not batched:
        Ta, Za, Ha, pa, pP, dv = Ta[None], Za[None], Ha[None], pa[None], pP[None], dv[None]
    sa, sP = eng.smooth_from_filter(y, Ta, Za, Ha, pa, pP, d=dv, filter_type=filter_type)
    if not batched:
        sa, sP = sa[0], sP[0]
    return [sa.cpu().numpy()[..., None], sP.cpu().numpy()]


# --------------------------------------------------------------------------------------------
# PyTensor Ops (built on first use)
# --------------------------------------------------------------------------------------------
_OPS = None


def _placeholder(ndim):
    """What an Op stores for an output it was told not to compute: zero-size for ndim >= 1, NaN for a 0-d output
    (np.zeros(()) would be the VALUE 0.0 -- a silent wrong answer for a log-likelihood; NaN cannot be mistaken for one)."""
    return np.zeros((0,) * ndim, dtype=np.float64) if ndim else np.asarray(np.nan, dtype=np.float64)


def _shapes_following_want(shapes, want):
    """infer_shape of an Op with a ``want`` mask: an output that is not computed has the placeholder's shape."""
    return [s if w else (0,) * len(s) for s, w in zip(shapes, want)]


OUTPUT_NAMES = ("filtered_states", "predicted_states", "filtered_covariances", "predicted_covariances",
                "log_likelihood", "ll_obs")


def want_mask(outputs):
    """``outputs`` of build_graph -> six booleans in the reference's output order (kalman_filter.py:184-191).
    "all": every output; otherwise an iterable of names from OUTPUT_NAMES.  The mask is a property of the Op
    chosen when the graph is BUILT: ``perform`` receives no FunctionGraph in current PyTensor (Apply nodes
    do not carry one), so which outputs have clients cannot be looked up at run time."""
    if outputs == "all" or outputs is None:
        return (True,) * 6
    names = (outputs,) if isinstance(outputs, str) else tuple(outputs)
    unknown = [nm for nm in names if nm not in OUTPUT_NAMES]
    if unknown:
        raise ValueError(f"unknown filter output(s) {unknown}; choose from {OUTPUT_NAMES}")
    return tuple(nm in names for nm in OUTPUT_NAMES)


def _build_ops():
    global _OPS
    if _OPS is not None:
        return _OPS
    try:
        import pytensor
        import pytensor.tensor as pt
        from pytensor.gradient import DisconnectedType, grad_not_implemented
        from pytensor.graph.basic import Apply
        from pytensor.graph.op import Op
    except ImportError as e:  # pragma: no cover - PyTensor is absent from the build image
        raise ImportError(
            "build_graph() needs PyTensor (pip install pytensor pymc). The eager call "
            "`flt(data, a0, P0, T, Z, R, H, Q)` works without it.") from e

    f64 = "float64"

    class KalmanFilterGradOp(Op):
        __props__ = ("filter_type", "mvn_const", "h_diagonal")

        def __init__(self, filter_type="standard", mvn_const="reference", h_diagonal=False):
            self.filter_type, self.mvn_const, self.h_diagonal = filter_type, mvn_const, bool(h_diagonal)

        def make_node(self, *args):
            args = [pt.as_tensor_variable(a) for a in args]
            # inputs: data a0 P0 T Z R H Q c d predicted_states predicted_covs g_logp g_ll_obs
            #         [g_filtered_states g_predicted_states g_filtered_covs g_predicted_covs]   (round 6; zero-size = none)
            outs = [args[i].type() for i in range(1, 10)]
            return Apply(self, args, outs)

        def perform(self, node, inputs, output_storage):
            grads = perform_filter_grad(*inputs, filter_type=self.filter_type,
                                        mvn_const=self.mvn_const, h_diagonal=self.h_diagonal)
            for storage, g in zip(output_storage, grads):
                storage[0] = _out(g)

        def infer_shape(self, fgraph, node, shapes):
            return [shapes[i] for i in range(1, 10)]

    class KalmanFilterOp(Op):
        """[filtered_states, predicted_states, filtered_covs, predicted_covs, logp, ll_obs]
        = Op(data, a0, P0, T, Z, R, H, Q, c, d)   -- reference kalman_filter.py:184-191."""
        __props__ = ("filter_type", "mvn_const", "h_diagonal", "want")

        def __init__(self, filter_type="standard", mvn_const="reference", h_diagonal=False, want=(True,) * 6):
            self.filter_type, self.mvn_const, self.h_diagonal = filter_type, mvn_const, bool(h_diagonal)
            self.want = tuple(bool(w) for w in want)

        def make_node(self, data, a0, P0, T, Z, R, H, Q, c, d):
            ins = [pt.as_tensor_variable(x) for x in (data, a0, P0, T, Z, R, H, Q, c, d)]
            outs = [pt.tensor3(dtype=f64), pt.tensor3(dtype=f64), pt.tensor3(dtype=f64),
                    pt.tensor3(dtype=f64), pt.scalar(dtype=f64), pt.vector(dtype=f64)]
            return Apply(self, ins, outs)

        def perform(self, node, inputs, output_storage):
            want = self.want
            outs = perform_filter(*inputs, filter_type=self.filter_type, mvn_const=self.mvn_const,
                                  h_diagonal=self.h_diagonal, want=want)
            for storage, o, var in zip(output_storage, outs, node.outputs):
                # an output nothing consumes still needs a correctly typed placeholder
                storage[0] = _out(o) if o is not None else _placeholder(var.type.ndim)

        def infer_shape(self, fgraph, node, shapes):
            n = shapes[0][0]
            m = shapes[3][-1]  # T is (m,m), or (n,m,m) when time-varying
            want = tuple(self.want[:4]) + (True, self.want[5])  # (the log-likelihood is always computed)
            return _shapes_following_want([(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,)], want)

        def connection_pattern(self, node):
            # data is never differentiated; every system matrix reaches every output
            return [[False] * 6] + [[True] * 6 for _ in range(9)]

        def L_op(self, inputs, outputs, output_grads):
            g_logp, g_ll = output_grads[4], output_grads[5]
            if any(not isinstance(output_grads[i].type, DisconnectedType) for i in range(4)):
                # round 6: a cotangent on a state / covariance tape -- kf_filter_backward_tapes behind one node that takes all six
                # cotangents (a zero-size placeholder where nothing is connected; the predicted tapes are recomputed inside when
                # this node was built without them)
                if isinstance(g_logp.type, DisconnectedType):
                    g_logp = pt.zeros(())
                if isinstance(g_ll.type, DisconnectedType):
                    g_ll = pt.zeros((inputs[0].shape[0],))
                tape_g = [pt.zeros((0, 0, 0)) if isinstance(output_grads[i].type, DisconnectedType) else output_grads[i]
                          for i in range(4)]
                grads = KalmanFilterGradOp(self.filter_type, self.mvn_const, self.h_diagonal)(
                    *inputs, outputs[1], outputs[3], g_logp, g_ll, *tape_g)
                return [DisconnectedType()()] + list(grads)
            if not (self.want[1] and self.want[3]):
                # the predicted tapes the adjoint kernel reads were not requested from this node: differentiate logp
                # through the fused value-and-gradient node instead (ll_obs cotangents need the tapes)
                if not isinstance(g_ll.type, DisconnectedType):
                    return [DisconnectedType()()] + [
                        grad_not_implemented(self, k, inputs[k], "a cotangent on ll_obs needs the predicted tapes: "
                                             "request predicted_states and predicted_covariances")
                        for k in range(1, 10)]
                fused = KalmanLogpOp(self.filter_type, self.mvn_const, self.h_diagonal)(*inputs)
                return [DisconnectedType()()] + [g_logp * fused[1 + k] for k in range(9)]
            # (zero cotangents shaped from the INPUTS: outputs[5] is a zero-size placeholder when ll_obs was not requested)
            if isinstance(g_logp.type, DisconnectedType):
                g_logp = pt.zeros(())
            if isinstance(g_ll.type, DisconnectedType):
                g_ll = pt.zeros((inputs[0].shape[0],))
            grads = KalmanFilterGradOp(self.filter_type, self.mvn_const, self.h_diagonal)(
                *inputs, outputs[1], outputs[3], g_logp, g_ll)
            return [DisconnectedType()()] + list(grads)

    class KalmanLogpOp(Op):
        """[logp, g_a0, g_P0, g_T, g_Z, g_R, g_H, g_Q, g_c, g_d] = Op(data, a0, P0, T, Z, R, H, Q, c, d):
        value and gradient from one perform (perform_logp_and_grad); L_op hands out the node's own
        gradient outputs scaled by the incoming cotangent."""
        __props__ = ("filter_type", "mvn_const", "h_diagonal")

        def __init__(self, filter_type="standard", mvn_const="reference", h_diagonal=False):
            self.filter_type, self.mvn_const, self.h_diagonal = filter_type, mvn_const, bool(h_diagonal)

        def make_node(self, data, a0, P0, T, Z, R, H, Q, c, d):
            ins = [pt.as_tensor_variable(x) for x in (data, a0, P0, T, Z, R, H, Q, c, d)]
            return Apply(self, ins, [pt.scalar(dtype=f64)] + [ins[i].type() for i in range(1, 10)])

        def perform(self, node, inputs, output_storage):
            outs = perform_logp_and_grad(*inputs, filter_type=self.filter_type, mvn_const=self.mvn_const,
                                         h_diagonal=self.h_diagonal)
            for storage, o in zip(output_storage, outs):
                storage[0] = _out(o)

        def infer_shape(self, fgraph, node, shapes):
            return [()] + [shapes[i] for i in range(1, 10)]

        def connection_pattern(self, node):
            # only logp (output 0) is differentiable, and not with respect to the data
            return [[False] * 10] + [[True] + [False] * 9 for _ in range(9)]

        def L_op(self, inputs, outputs, output_grads):
            g = output_grads[0]
            return [DisconnectedType()()] + [g * outputs[1 + k] for k in range(9)]

    def _bcast_rows(g, like):
        """cotangent g (B,) against a (B, ...) gradient array: g[:, None, ...] * like."""
        return g.dimshuffle(0, *(["x"] * (like.type.ndim - 1))) * like

    class KalmanLogpBatchedOp(Op):
        """[logp (B,), g_a0, g_P0, g_T, g_Z, g_R, g_H, g_Q, g_c, g_d] = Op(data, a0, P0, T, Z, R, H, Q, c, d) with a leading draw
        axis B on a0 .. d (data (n,p,1) shared, or (B,n,p,1)): value and gradient of B draws from ONE perform = ONE
        kf_filter_value_and_grad call (perform_logp_and_grad_batched).  What ``vectorize_graph`` maps ``KalmanLogpOp`` to, and
        what ``KalmanFilterBatchedOp`` differentiates its log-likelihood through: the batch-of-draws axis reaches the batched
        kernels WITH gradients through ``build_statespace_graph()``."""
        __props__ = ("filter_type", "mvn_const", "h_diagonal")

        def __init__(self, filter_type="standard", mvn_const="reference", h_diagonal=False):
            self.filter_type, self.mvn_const, self.h_diagonal = filter_type, mvn_const, bool(h_diagonal)

        def make_node(self, data, a0, P0, T, Z, R, H, Q, c, d):
            ins = [pt.as_tensor_variable(x) for x in (data, a0, P0, T, Z, R, H, Q, c, d)]
            if any(v.type.ndim not in (3, 4) for v in ins[1:]) or ins[1].type.ndim != 3 or ins[2].type.ndim != 3 \
                    or ins[0].type.ndim not in (3, 4):
                raise TypeError("KalmanLogpBatchedOp: a0 .. d carry one leading draw axis (3-D each; a time-varying matrix, c or d: "
                                "4-D, time behind the draw axis); data is (n,p,1) or (B,n,p,1)")
            return Apply(self, ins, [pt.vector(dtype=f64)] + [ins[i].type() for i in range(1, 10)])

        def perform(self, node, inputs, output_storage):
            outs = perform_logp_and_grad_batched(*inputs, filter_type=self.filter_type, mvn_const=self.mvn_const,
                                                 h_diagonal=self.h_diagonal)
            for storage, o in zip(output_storage, outs):
                storage[0] = _out(o)

        def infer_shape(self, fgraph, node, shapes):
            return [(shapes[2][0],)] + [shapes[i] for i in range(1, 10)]

        def connection_pattern(self, node):
            return [[False] * 10] + [[True] + [False] * 9 for _ in range(9)]

        def L_op(self, inputs, outputs, output_grads):
            g = output_grads[0]  # (B,): draw b's cotangent scales row b of every gradient array
            return [DisconnectedType()()] + [_bcast_rows(g, outputs[1 + k]) for k in range(9)]

    class KalmanFilterBatchedGradOp(Op):
        """KalmanFilterGradOp with ONE leading draw axis: gradients = Op(data, a0 .. d, predicted_states, predicted_covs,
        g_logp (B,), g_ll_obs (B,n)) -- perform_filter_grad_batched, one kf_filter_backward call for all draws."""
        __props__ = ("filter_type", "mvn_const", "h_diagonal")

        def __init__(self, filter_type="standard", mvn_const="reference", h_diagonal=False):
            self.filter_type, self.mvn_const, self.h_diagonal = filter_type, mvn_const, bool(h_diagonal)

        def make_node(self, *args):
            args = [pt.as_tensor_variable(a) for a in args]
            return Apply(self, args, [args[i].type() for i in range(1, 10)])

        def perform(self, node, inputs, output_storage):
            grads = perform_filter_grad_batched(*inputs, filter_type=self.filter_type, mvn_const=self.mvn_const,
                                                h_diagonal=self.h_diagonal)
            for storage, g in zip(output_storage, grads):
                storage[0] = _out(g)

        def infer_shape(self, fgraph, node, shapes):
            return [shapes[i] for i in range(1, 10)]

    class KalmanFilterBatchedOp(Op):
        """KalmanFilterOp with a leading draw axis B on a0 .. d (data shared): outputs B-leading."""
        __props__ = ("filter_type", "mvn_const", "h_diagonal", "want")

        def __init__(self, filter_type="standard", mvn_const="reference", h_diagonal=False, want=(True,) * 6):
            self.filter_type, self.mvn_const, self.h_diagonal = filter_type, mvn_const, bool(h_diagonal)
            self.want = tuple(bool(w) for w in want)

        def make_node(self, data, a0, P0, T, Z, R, H, Q, c, d):
            ins = [pt.as_tensor_variable(x) for x in (data, a0, P0, T, Z, R, H, Q, c, d)]
            t4 = pt.tensor4(dtype=f64)
            outs = [t4.type(), t4.type(), t4.type(), t4.type(), pt.vector(dtype=f64), pt.matrix(dtype=f64)]
            return Apply(self, ins, outs)

        def perform(self, node, inputs, output_storage):
            want = self.want
            outs = perform_filter_batched(*inputs, filter_type=self.filter_type, mvn_const=self.mvn_const,
                                          h_diagonal=self.h_diagonal, want=want)
            for storage, o, var in zip(output_storage, outs, node.outputs):
                storage[0] = _out(o) if o is not None else _placeholder(var.type.ndim)

        def infer_shape(self, fgraph, node, shapes):
            n, B, m = shapes[0][-3], shapes[3][0], shapes[3][-1]   # T is (B,m,m), or (B,n,m,m) when time-varying
            want = tuple(self.want[:4]) + (True, self.want[5])
            return _shapes_following_want([(B, n, m, 1), (B, n + 1, m, 1), (B, n, m, m), (B, n + 1, m, m), (B,), (B, n)], want)

        def connection_pattern(self, node):
            return [[False] * 6] + [[True] * 6 for _ in range(9)]

        def L_op(self, inputs, outputs, output_grads):
            # gradients flow through the per-draw log-likelihoods (logp, ll_obs) only.  A cotangent on logp alone comes from the
            # fused batched value-and-gradient node (one kf_filter_value_and_grad call for all B draws); one on ll_obs (ADVICE
            # round 4: a burn-in mask after vectorize_graph) from the batched adjoint over this node's predicted tapes
            others = [output_grads[i] for i in (0, 1, 2, 3)]
            g_logp, g_ll = output_grads[4], output_grads[5]
            if any(not isinstance(g.type, DisconnectedType) for g in others):   # (round 6: as the per-draw node)
                if isinstance(g_logp.type, DisconnectedType):
                    g_logp = pt.zeros((inputs[2].shape[0],))
                if isinstance(g_ll.type, DisconnectedType):
                    g_ll = pt.zeros((inputs[2].shape[0], inputs[0].shape[-3]))
                tape_g = [pt.zeros((0, 0, 0, 0)) if isinstance(g.type, DisconnectedType) else g for g in others]
                grads = KalmanFilterBatchedGradOp(self.filter_type, self.mvn_const, self.h_diagonal)(
                    *inputs, outputs[1], outputs[3], g_logp, g_ll, *tape_g)
                return [DisconnectedType()()] + list(grads)
            if isinstance(g_ll.type, DisconnectedType):
                fused = KalmanLogpBatchedOp(self.filter_type, self.mvn_const, self.h_diagonal)(*inputs)
                return [DisconnectedType()()] + [_bcast_rows(g_logp, fused[1 + k]) for k in range(9)]
            if not (self.want[1] and self.want[3]):
                return [DisconnectedType()()] + [
                    grad_not_implemented(self, k, inputs[k], "a cotangent on ll_obs needs the predicted tapes: request "
                                         "predicted_states and predicted_covariances")
                    for k in range(1, 10)]
            if isinstance(g_logp.type, DisconnectedType):
                g_logp = pt.zeros((inputs[2].shape[0],))
            grads = KalmanFilterBatchedGradOp(self.filter_type, self.mvn_const, self.h_diagonal)(
                *inputs, outputs[1], outputs[3], g_logp, g_ll)
            return [DisconnectedType()()] + list(grads)

    class SteadyStateGradOp(Op):
        """gradients = Op(data, a0, P0, T, Z, R, H, Q, c, d, g_logp, g_ll_obs) -- perform_steady_state_grad."""
        __props__ = ("mvn_const",)

        def __init__(self, mvn_const="reference"):
            self.mvn_const = mvn_const

        def make_node(self, *args):
            args = [pt.as_tensor_variable(a) for a in args]
            return Apply(self, args, [args[i].type() for i in range(1, 10)])

        def perform(self, node, inputs, output_storage):
            outs = perform_steady_state_grad(*inputs[:10], g_logp=inputs[10], g_ll_obs=inputs[11],
                                             mvn_const=self.mvn_const)
            for storage, g in zip(output_storage, outs[1:]):
                storage[0] = _out(g)

        def infer_shape(self, fgraph, node, shapes):
            return [shapes[i] for i in range(1, 10)]

    class SteadyStateFilterOp(Op):
        """The six reference outputs of SteadyStateFilter.build_graph (kalman_filter.py:364-397) from one node:
        kf_dare + the Standard recursion from P_steady; gradients (through logp / ll_obs) include the Riccati solve."""
        __props__ = ("mvn_const", "want")

        def __init__(self, mvn_const="reference", want=(True,) * 6):
            self.mvn_const = mvn_const
            self.want = tuple(bool(w) for w in want)

        def make_node(self, data, a0, P0, T, Z, R, H, Q, c, d):
            ins = [pt.as_tensor_variable(x) for x in (data, a0, P0, T, Z, R, H, Q, c, d)]
            outs = [pt.tensor3(dtype=f64), pt.tensor3(dtype=f64), pt.tensor3(dtype=f64),
                    pt.tensor3(dtype=f64), pt.scalar(dtype=f64), pt.vector(dtype=f64)]
            return Apply(self, ins, outs)

        def perform(self, node, inputs, output_storage):
            want = self.want
            outs = perform_steady_state(*inputs, mvn_const=self.mvn_const, want=want)
            for storage, o, var in zip(output_storage, outs, node.outputs):
                storage[0] = _out(o) if o is not None else _placeholder(var.type.ndim)

        def infer_shape(self, fgraph, node, shapes):
            n, m = shapes[0][0], shapes[3][0]
            return [(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,)]

        def connection_pattern(self, node):
            return [[False] * 6] + [[True] * 6 for _ in range(9)]

        def L_op(self, inputs, outputs, output_grads):
            for i in range(4):
                if not isinstance(output_grads[i].type, DisconnectedType):
                    return [DisconnectedType()()] + [
                        grad_not_implemented(self, k, inputs[k], "gradients flow through log_likelihood / ll_obs only")
                        for k in range(1, 10)]
            g_logp, g_ll = output_grads[4], output_grads[5]
            if isinstance(g_logp.type, DisconnectedType):
                g_logp = pt.zeros_like(outputs[4])
            if isinstance(g_ll.type, DisconnectedType):
                g_ll = pt.zeros_like(outputs[5])
            return [DisconnectedType()()] + list(SteadyStateGradOp(self.mvn_const)(*inputs, g_logp, g_ll))

    class SteadyStateLogpOp(Op):
        """[logp, g_a0 .. g_d] = Op(data, a0, P0, T, Z, R, H, Q, c, d): value and gradient of the SteadyStateFilter's
        log-likelihood from one perform (KalmanLogpOp's pattern)."""
        __props__ = ("mvn_const",)

        def __init__(self, mvn_const="reference"):
            self.mvn_const = mvn_const

        def make_node(self, data, a0, P0, T, Z, R, H, Q, c, d):
            ins = [pt.as_tensor_variable(x) for x in (data, a0, P0, T, Z, R, H, Q, c, d)]
            return Apply(self, ins, [pt.scalar(dtype=f64)] + [ins[i].type() for i in range(1, 10)])

        def perform(self, node, inputs, output_storage):
            outs = perform_steady_state_grad(*inputs, mvn_const=self.mvn_const)
            for storage, o in zip(output_storage, outs):
                storage[0] = _out(o)

        def infer_shape(self, fgraph, node, shapes):
            return [()] + [shapes[i] for i in range(1, 10)]

        def connection_pattern(self, node):
            return [[False] * 10] + [[True] + [False] * 9 for _ in range(9)]

        def L_op(self, inputs, outputs, output_grads):
            g = output_grads[0]
            return [DisconnectedType()()] + [g * outputs[1 + k] for k in range(9)]

    try:  # vectorize_graph (PyMC's draw-vectorised predictive sampling) -> the batched kernels
        from pytensor.graph.replace import _vectorize_node

        def _batched_inputs(node, data, mats):
            """Inputs of a per-draw node after vectorize_graph gave some of them ONE leading draw axis -> (data, B-leading mats),
            or None when nothing is batched.  Unbatched matrices are broadcast along the new axis."""
            # per draw: a0, c, d are (m,1) / (p,1) columns and the matrices 2-D; a time-varying matrix, c or d is 3-D, time on axis 0
            # (round 6: such nodes vectorise too -- the batched kernels take (B, n, ...) inputs)
            core = [v.type.ndim for v in node.inputs[1:]]
            if core[0] != 2 or core[1] != 2 or any(nd not in (2, 3) for nd in core):
                raise NotImplementedError("vectorize_graph: a0 and P0 are static; system matrices are 2-D (static) or 3-D (time-varying)")
            d_extra = data.type.ndim - node.inputs[0].type.ndim
            if d_extra not in (0, 1):
                raise NotImplementedError("one leading batch axis only")
            if d_extra == 0 and not any(x.type.ndim > nd for x, nd in zip(mats, core)):
                return None
            B = data.shape[0] if d_extra else None
            for x, nd in zip(mats, core):
                if x.type.ndim == nd + 1:
                    B = x.shape[0]
                elif x.type.ndim != nd:
                    raise NotImplementedError("one leading batch axis only")
            mats = [x if x.type.ndim == nd + 1 else pt.broadcast_to(x, (B,) + tuple(x.shape))
                    for x, nd in zip(mats, core)]
            return data, mats

        @_vectorize_node.register(KalmanFilterOp)
        def _vectorize_kalman_filter(op, node, data, *mats):
            bi = _batched_inputs(node, data, mats)
            if bi is None:
                return op.make_node(data, *mats)
            return KalmanFilterBatchedOp(op.filter_type, op.mvn_const, op.h_diagonal, op.want).make_node(bi[0], *bi[1])

        @_vectorize_node.register(KalmanLogpOp)
        def _vectorize_kalman_logp(op, node, data, *mats):
            # the sampler's node over a batch of draws: logp (B,) and the nine gradient arrays from ONE perform
            bi = _batched_inputs(node, data, mats)
            if bi is None:
                return op.make_node(data, *mats)
            return KalmanLogpBatchedOp(op.filter_type, op.mvn_const, op.h_diagonal).make_node(bi[0], *bi[1])
    except ImportError:  # pragma: no cover - older PyTensor
        pass

    class KalmanSmootherOp(Op):
        """(smoothed_states, smoothed_covs) = Op(T, R, Q, filtered_states, filtered_covs)
        -- reference kalman_smoother.py:56-85."""
        __props__ = ()

        def make_node(self, T, R, Q, filtered_states, filtered_covs):
            ins = [pt.as_tensor_variable(x) for x in (T, R, Q, filtered_states, filtered_covs)]
            return Apply(self, ins, [pt.tensor3(dtype=f64), pt.tensor3(dtype=f64)])

        def perform(self, node, inputs, output_storage):
            outs = perform_smoother(*inputs)
            for storage, o in zip(output_storage, outs):
                storage[0] = _out(o)

        def infer_shape(self, fgraph, node, shapes):
            return [shapes[3], shapes[4]]

        def L_op(self, inputs, outputs, output_grads):
            return [grad_not_implemented(self, k, inputs[k], "the RTS smoother is not differentiated")
                    for k in range(5)]

    class KalmanSmootherBatchedOp(Op):
        """KalmanSmootherOp with ONE leading draw axis B on every input: T (B,m,m), R, Q, filtered_states (B,n,m,1),
        filtered_covs (B,n,m,m) -> B-leading outputs from ONE kf_smoother call.  What ``vectorize_graph`` maps
        ``KalmanSmootherOp`` to (sample_conditional_posterior's per-draw loop, reference core/statespace.py:283-302)."""
        __props__ = ()

        def make_node(self, T, R, Q, filtered_states, filtered_covs):
            ins = [pt.as_tensor_variable(x) for x in (T, R, Q, filtered_states, filtered_covs)]
            nds = [v.type.ndim for v in ins]
            if any(nd not in (3, 4) for nd in nds[:3]) or nds[3:] != [4, 4]:
                raise TypeError("KalmanSmootherBatchedOp: T, R, Q (3-D; time-varying: 4-D) and the filtered tapes, ONE leading draw axis each")
            return Apply(self, ins, [pt.tensor4(dtype=f64), pt.tensor4(dtype=f64)])

        def perform(self, node, inputs, output_storage):
            outs = perform_smoother_batched(*inputs)
            for storage, o in zip(output_storage, outs):
                storage[0] = _out(o)

        def infer_shape(self, fgraph, node, shapes):
            return [shapes[3], shapes[4]]

        def L_op(self, inputs, outputs, output_grads):
            return [grad_not_implemented(self, k, inputs[k], "the RTS smoother is not differentiated")
                    for k in range(5)]

    class KalmanSmootherFusedOp(Op):
        """(smoothed_states, smoothed_covs) = Op(data, T, Z, R, H, Q, c, d, filtered_states, filtered_covs, predicted_states,
        predicted_covs): the smoother of a model whose tapes come from ``KalmanFilterOp`` on the same graph -- what
        ``kalman_smoother_graph`` builds when it finds that node behind ``filtered_states`` (perform_smoother_fused: the
        predicted-tape recursion where it applies and the state intercept c is zero, the reference's recursion on the filtered
        tapes otherwise).
        ``batched``: every input but a shared ``data`` carries ONE leading draw axis (the ``vectorize_graph`` image)."""
        __props__ = ("filter_type", "batched")

        def __init__(self, filter_type="standard", batched=False):
            self.filter_type, self.batched = filter_type, bool(batched)

        def make_node(self, *args):
            ins = [pt.as_tensor_variable(x) for x in args]
            if len(ins) != 12:
                raise TypeError("KalmanSmootherFusedOp takes 12 inputs")
            out = pt.tensor4 if self.batched else pt.tensor3
            return Apply(self, ins, [out(dtype=f64), out(dtype=f64)])

        def perform(self, node, inputs, output_storage):
            outs = perform_smoother_fused(*inputs, filter_type=self.filter_type, batched=self.batched)
            for storage, o in zip(output_storage, outs):
                storage[0] = _out(o)

        def infer_shape(self, fgraph, node, shapes):
            return [shapes[8], shapes[9]]

        def L_op(self, inputs, outputs, output_grads):
            return [grad_not_implemented(self, k, inputs[k], "the smoother is not differentiated") for k in range(12)]

    try:
        from pytensor.graph.replace import _vectorize_node as _vn

        def _lead_all(node_inputs, new_inputs, shared=()):
            """One leading draw axis on every input (broadcasting the ones vectorize_graph left alone), or None when nothing is
            batched.  ``shared``: positions that may stay unbatched (the data)."""
            extra = [x.type.ndim - v.type.ndim for x, v in zip(new_inputs, node_inputs)]
            if any(e not in (0, 1) for e in extra):
                raise NotImplementedError("one leading batch axis only")
            if not any(extra):
                return None
            B = next(x.shape[0] for x, e in zip(new_inputs, extra) if e)
            return [x if (e or i in shared) else pt.broadcast_to(x, (B,) + tuple(x.shape))
                    for i, (x, e) in enumerate(zip(new_inputs, extra))]

        @_vn.register(KalmanSmootherOp)
        def _vectorize_kalman_smoother(op, node, *ins):
            led = _lead_all(node.inputs, ins)   # (round 6: a time-varying T / R / Q keeps its time axis behind the draw axis)
            if led is None:
                return op.make_node(*ins)
            return KalmanSmootherBatchedOp().make_node(*led)

        @_vn.register(KalmanSmootherFusedOp)
        def _vectorize_kalman_smoother_fused(op, node, *ins):
            if op.batched:
                raise NotImplementedError("one leading batch axis only")
            led = _lead_all(node.inputs, ins, shared=(0,))   # (time-varying inputs: the body falls back to kf_smoother on the filtered tapes)
            if led is None:
                return op.make_node(*ins)
            return KalmanSmootherFusedOp(op.filter_type, batched=True).make_node(*led)
    except ImportError:  # pragma: no cover - older PyTensor
        pass

    _OPS = dict(pt=pt, pytensor=pytensor, KalmanFilterOp=KalmanFilterOp,
                KalmanFilterGradOp=KalmanFilterGradOp, KalmanSmootherOp=KalmanSmootherOp,
                KalmanLogpOp=KalmanLogpOp, KalmanFilterBatchedOp=KalmanFilterBatchedOp,
                KalmanLogpBatchedOp=KalmanLogpBatchedOp, KalmanSmootherBatchedOp=KalmanSmootherBatchedOp,
                KalmanSmootherFusedOp=KalmanSmootherFusedOp, KalmanFilterBatchedGradOp=KalmanFilterBatchedGradOp,
                SteadyStateFilterOp=SteadyStateFilterOp, SteadyStateGradOp=SteadyStateGradOp,
                SteadyStateLogpOp=SteadyStateLogpOp)
    return _OPS


def kalman_filter_batched_graph(data, a0, P0, T, Z, R, H, Q, c=None, d=None, filter_type="standard",
                                mvn_const="reference", h_diagonal=False, outputs="all"):
    """``kalman_filter_graph`` for inputs that already carry ONE leading draw axis B (a0 (B,m,1), P0 (B,m,m), ...; data (n,p,1)
    shared or (B,n,p,1)): the graph ``vectorize_graph`` would produce, built directly.  ``outputs="logp"``: logp (B,) from
    ``KalmanLogpBatchedOp`` (value and gradients of all draws from one C-ABI call); otherwise the six B-leading outputs."""
    ops = _build_ops()
    pt = ops["pt"]
    a0, P0, T, Z, R, H, Q = (pt.as_tensor_variable(x) for x in (a0, P0, T, Z, R, H, Q))
    for prm, name in zip((T, Z, R, H, Q), ("T", "Z", "R", "H", "Q")):
        if prm.ndim != 3:
            raise ValueError(f"Matrix {name} has {prm.ndim} dimensions, a batch of static matrices has 3 (draws, rows, columns).")
    if c is None:
        c = pt.zeros((T.shape[0], T.shape[-1], 1))
    if d is None:
        d = pt.zeros((Z.shape[0], Z.shape[-2], 1))
    if outputs == "logp":
        return ops["KalmanLogpBatchedOp"](filter_type, mvn_const, h_diagonal)(data, a0, P0, T, Z, R, H, Q, c, d)[0]
    return list(ops["KalmanFilterBatchedOp"](filter_type, mvn_const, h_diagonal, want_mask(outputs))(
        data, a0, P0, T, Z, R, H, Q, c, d))


def kalman_filter_graph(data, a0, P0, T, Z, R, H, Q, c=None, d=None, filter_type="standard",
                        mvn_const="reference", h_diagonal=False, outputs="all"):
    """Symbolic equivalent of BaseFilter.build_graph (reference kalman_filter.py:126-163).
    System matrices are 2-D (static) or 3-D (time-varying, time on axis 0: step t uses slice t, and the gradient
    of such a matrix has its 3-D shape); c, d default to zeros (initialize_intercepts :36-51).
    ``h_diagonal=True``: the model's H is diagonal by construction (e.g. VARMAX measurement
    error, reference models/VARMAX.py:80-83) -- the gradient then runs on the scalar-observation
    adjoint kernel and is returned for diag(H) only (KF_FLAG_H_DIAGONAL, include/kf.h).
    ``outputs="logp"`` returns the log-likelihood alone, from ``KalmanLogpOp``; an iterable of names from
    ``OUTPUT_NAMES`` keeps the six-element return list but computes / copies back only the named outputs."""
    ops = _build_ops()
    pt = ops["pt"]
    T, Z, R, H, Q = (pt.as_tensor_variable(x) for x in (T, Z, R, H, Q))
    for prm, name in zip((T, Z, R, H, Q), ("T", "Z", "R", "H", "Q")):
        if prm.ndim not in (2, 3):
            # reference filters/utilities.py:16-18
            raise ValueError(
                f"Matrix {name} has {prm.ndim}, it should either 2 (static) or 3 (time varying).")
    if c is None:
        c = pt.zeros((T.shape[-1], 1))
    if d is None:
        d = pt.zeros((Z.shape[-2], 1))
    if outputs == "logp":  # the sampler's path: value and gradient from one node, no tapes over PCIe
        return ops["KalmanLogpOp"](filter_type, mvn_const, h_diagonal)(data, a0, P0, T, Z, R, H, Q, c, d)[0]
    return list(ops["KalmanFilterOp"](filter_type, mvn_const, h_diagonal, want_mask(outputs))(
        data, a0, P0, T, Z, R, H, Q, c, d))


def steady_state_filter_graph(data, a0, P0, T, Z, R, H, Q, c=None, d=None, mvn_const="reference", outputs="all"):
    """Symbolic SteadyStateFilter.build_graph (reference kalman_filter.py:364-397): static matrices only (:380-383)."""
    ops = _build_ops()
    pt = ops["pt"]
    T, Z, R, H, Q = (pt.as_tensor_variable(x) for x in (T, Z, R, H, Q))
    for prm in (T, Z, R, H, Q):
        if prm.ndim != 2:
            raise ValueError("All system matrices must be time-invariant to use the SteadyStateFilter")
    if c is None:
        c = pt.zeros((T.shape[0], 1))
    if d is None:
        d = pt.zeros((Z.shape[0], 1))
    if outputs == "logp":
        return ops["SteadyStateLogpOp"](mvn_const)(data, a0, P0, T, Z, R, H, Q, c, d)[0]
    return list(ops["SteadyStateFilterOp"](mvn_const, want_mask(outputs))(data, a0, P0, T, Z, R, H, Q, c, d))


def _same_variable(x, fx):
    """Is the smoother's matrix ``x`` the filter node's input ``fx``?  The same variable, or two constants holding the same values
    (``as_tensor_variable`` of one numpy array twice gives two equal constants)."""
    if x is fx:
        return True
    dx, dfx = getattr(x, "data", None), getattr(fx, "data", None)  # TensorConstant.data
    if dx is None or dfx is None or getattr(x, "owner", None) is not None or getattr(fx, "owner", None) is not None:
        return False
    return np.shape(dx) == np.shape(dfx) and bool(np.all(np.asarray(dx) == np.asarray(dfx)))


def kalman_smoother_graph(T, R, Q, filtered_states, filtered_covariances, fuse=True):
    """Symbolic KalmanSmoother.build_graph (reference kalman_smoother.py:56-85).  The reference only ever smooths the outputs of
    its own filter on the same model (core/statespace.py:164-197); when ``filtered_states`` and ``filtered_covariances`` ARE
    outputs 0 and 2 of one ``KalmanFilterOp`` / ``KalmanFilterBatchedOp`` node that also computes its predicted tapes, the
    smoother node takes that node's inputs and predicted tapes along (``KalmanSmootherFusedOp``) -- provided T, R, Q ARE that node's
    T, R, Q: for p == 1, static matrices, a zero state intercept c (checked at run time: the reference's smoother ignores c,
    kalman_smoother.py:99-104, the predicted tapes do not) and the shapes of ``engine.smoother_from_filter_applies`` its body is
    ``kf_smoother_from_filter`` -- same moments, no pinv(P_hat).  ``fuse=False`` (or tapes of any other origin, or another T / R / Q):
    the reference's recursion on the filtered tapes."""
    ops = _build_ops()
    pt = ops["pt"]
    fs, fc = pt.as_tensor_variable(filtered_states), pt.as_tensor_variable(filtered_covariances)
    T, R, Q = (pt.as_tensor_variable(x) for x in (T, R, Q))
    own = getattr(fs, "owner", None)
    if fuse and own is not None and getattr(fc, "owner", None) is own and isinstance(
            own.op, (ops["KalmanFilterOp"], ops["KalmanFilterBatchedOp"])):
        batched = isinstance(own.op, ops["KalmanFilterBatchedOp"])
        want = own.op.want
        if own.outputs[0] is fs and own.outputs[2] is fc and want[1] and want[3]:
            data, _a0, _P0, fT, fZ, fR, fH, fQ, fc_, fd = own.inputs
            # the predicted tapes are the FILTER's model: fuse only when the smoother is given that very model (the reference hands
            # both the same variables, core/statespace.py:164-197); anything else smooths the filtered tapes with the caller's T, R, Q
            if all(_same_variable(x, fx) for x, fx in ((T, fT), (R, fR), (Q, fQ))):
                return tuple(ops["KalmanSmootherFusedOp"](own.op.filter_type, batched)(
                    data, fT, fZ, fR, fH, fQ, fc_, fd, fs, fc, own.outputs[1], own.outputs[3]))
    if fs.type.ndim == 4:
        return tuple(ops["KalmanSmootherBatchedOp"]()(T, R, Q, fs, fc))
    return tuple(ops["KalmanSmootherOp"]()(T, R, Q, fs, fc))
