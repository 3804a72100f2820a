"""SteadyStateFilter (batched Riccati solve + filter) and batched
unconditional simulation.  CPU part: the restatements against SciPy / the definition; GPU part:
the HIP entry points against the restatements."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W
from oracle import steady_state_numpy as so


def _one(w, b=0):
    return {k: w[k][b] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")}


def test_oracle_dare_is_a_fixed_point_of_the_filter_recursion():
    w = W.structural(B=2, n=30)
    d = _one(w)
    out = so.steady_state_filter(w["y"], **d)
    P = out["P_steady"]
    Zm, H, T = d["Z"], d["H"], d["T"]
    F = Zm @ P @ Zm.T + H
    Pn = T @ (P - P @ Zm.T @ np.linalg.solve(F, Zm @ P)) @ T.T + d["R"] @ d["Q"] @ d["R"].T
    assert_allclose(Pn, P, rtol=1e-9, atol=1e-10)
    # started at the fixed point the covariance never moves, whatever the data
    assert_allclose(out["predicted_covs"][1:], np.broadcast_to(P, out["predicted_covs"][1:].shape),
                    rtol=1e-8, atol=1e-9)
    assert np.array_equal(out["predicted_covs"][0], d["P0"])


def test_oracle_simulation_matches_definition():
    rng = np.random.default_rng(0)
    w = W.random_dense(1, 5, 2, 4, 2, seed=3)
    d = _one(w)
    zs, zo = rng.standard_normal((50, 2)), rng.standard_normal((50, 2))
    st, ob = so.simulate_statespace(d["T"], d["Z"], d["R"], d["H"], d["Q"], zs, zo, x0=np.ones(4))
    assert_allclose(st[0], 1.0)
    LQ, LH = np.linalg.cholesky(d["Q"]), np.linalg.cholesky(d["H"])
    assert_allclose(st[7], d["T"] @ st[6] + d["R"] @ (zs[7] @ LQ))
    assert_allclose(ob[7], d["Z"] @ st[6] + zo[7] @ LH)


def _fd(f, x, eps=1e-6, symmetric=False):
    num = np.zeros_like(x)
    for idx in np.ndindex(x.shape):
        e = np.zeros_like(x)
        e[idx] = eps
        if symmetric:
            e = e + e.T
        num[idx] = (f(x + e) - f(x - e)) / (2 * eps)
    return num


def _dense_model(seed, m=4, p=2, r=3):
    rng = np.random.default_rng(seed)
    T = rng.standard_normal((m, m))
    T *= 0.8 / np.abs(np.linalg.eigvals(T)).max()
    Q = rng.standard_normal((r, r))
    H = rng.standard_normal((p, p))
    return dict(T=T, Z=rng.standard_normal((p, m)), R=rng.standard_normal((m, r)), H=H @ H.T + 0.5 * np.eye(p),
                Q=Q @ Q.T + np.eye(r))


def test_oracle_dare_gradient_matches_finite_differences():
    """dare_grad restates SolveDiscreteARE.grad (utils/pytensor_scipy.py:33-60); the reference has no test of it, so it
    is pinned here by central differences of SciPy's solve_discrete_are."""
    from scipy.linalg import solve_discrete_are

    mdl = _dense_model(0)
    gP = np.random.default_rng(1).standard_normal((4, 4))

    def obj(**kw):
        a = dict(mdl, **kw)
        return float(np.sum(gP * solve_discrete_are(a["T"].T, a["Z"].T, a["R"] @ a["Q"] @ a["R"].T, a["H"])))

    g = so.steady_state_grad(mdl["T"], mdl["Z"], mdl["R"], mdl["H"], mdl["Q"], gP)
    for k in ("T", "Z", "R"):
        num = _fd(lambda x: obj(**{k: x}), mdl[k])
        assert_allclose(g[k], num, rtol=1e-6, atol=1e-6, err_msg=k)
    for k in ("H", "Q"):  # symmetric arguments: only the symmetric part of the entrywise gradient is observable
        num = _fd(lambda x: obj(**{k: x}), mdl[k], symmetric=True)
        assert_allclose(g[k] + g[k].T, num, rtol=1e-6, atol=1e-6, err_msg=k)


def test_oracle_steady_state_logp_gradient_matches_finite_differences_of_the_frozen_gain_filter():
    """The filter started from P_steady and the reference's frozen-F_inv recursion are the same function of the
    parameters (complete data), so the gradient `Kalman adjoint + Riccati adjoint` must match central differences of
    the line-by-line restatement of SteadyStateFilter (kalman_filter.py:354-441)."""
    mdl = _dense_model(3, m=3, p=1, r=2)
    rng = np.random.default_rng(5)
    y = rng.standard_normal((12, 1))
    a0, P0 = rng.standard_normal(3), np.eye(3)
    lp, g = so.steady_state_logp_and_grad(y, a0, P0, **mdl)
    assert abs(lp - so.steady_state_filter(y, a0, P0, **mdl)["logp"]) < 1e-9
    for k in ("T", "Z", "R"):
        num = _fd(lambda x: so.steady_state_filter(y, a0, P0, **dict(mdl, **{k: x}))["logp"], mdl[k])
        assert_allclose(g[k], num, rtol=2e-6, atol=2e-6, err_msg=k)
    for k in ("H", "Q"):
        num = _fd(lambda x: so.steady_state_filter(y, a0, P0, **dict(mdl, **{k: x}))["logp"], mdl[k], symmetric=True)
        assert_allclose(g[k] + g[k].T, num, rtol=2e-6, atol=2e-6, err_msg=k)
    num = _fd(lambda x: so.steady_state_filter(y, x, P0, **mdl)["logp"], a0)
    assert_allclose(g["a0"], num, rtol=1e-6, atol=1e-6)


@pytest.fixture(scope="module")
def eng():
    import torch

    from pymc_statespace_amd import KalmanEngine

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return KalmanEngine()


STEADY_CASES = {
    "structural_m10": lambda: W.structural(B=6, n=60),
    "arma11": lambda: W.arma11(B=8, n=50),
    "varmax_k3": lambda: W.varmax(B=4, n=40, k=3, order=(1, 1)),
    "local_level": lambda: W.nile_local_level(B=3),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(STEADY_CASES))
def test_hip_dare_matches_scipy(eng, case):
    from scipy.linalg import solve_discrete_are

    w = STEADY_CASES[case]()
    P, info = eng.dare(w["T"], w["Z"], w["R"], w["H"], w["Q"])
    assert not info.cpu().numpy().any()
    P = P.cpu().numpy()
    for b in range(w["B"]):
        ref = solve_discrete_are(w["T"][b].T, w["Z"][b].T, w["R"][b] @ w["Q"][b] @ w["R"][b].T, w["H"][b])
        assert_allclose(P[b], ref, rtol=1e-8, atol=1e-9 * (1 + np.abs(ref).max()), err_msg=f"{case} draw {b}")


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["structural_m10", "varmax_k3", "local_level"])
def test_hip_steady_state_filter_matches_oracle(eng, case):
    from pymc_statespace_amd import FILTER_FACTORY

    w = STEADY_CASES[case]()
    # complete data: with missing rows the reference's FROZEN gain stops being the Kalman gain
    # (P leaves the fixed point while F_inv stays) and its recursion can even diverge; the HIP
    # filter keeps using the exact gain there -- a documented difference (filters.py)
    y = w["y"].copy()
    flt = FILTER_FACTORY["steady_state"](engine=eng)
    outs = flt(y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    names = ["filtered_states", "predicted_states", "filtered_covs", "predicted_covs", "logp", "ll_obs"]
    for b in range(w["B"]):
        ref = so.steady_state_filter(y, **_one(w, b))
        for nm, o in zip(names, outs):
            got = o[b][..., 0] if nm.endswith("states") else o[b]
            scale = 1.0 + np.max(np.abs(ref[nm]))
            assert np.max(np.abs(got - ref[nm])) / scale < 1e-8, (case, b, nm)
    # one draw, reference shapes
    one = flt(y, w["a0"][0], w["P0"][0], w["T"][0], w["Z"][0], w["R"][0], w["H"][0], w["Q"][0])
    n, m = y.shape[0], w["m"]
    assert one[0].shape == (n, m, 1) and one[3].shape == (n + 1, m, m) and one[4].shape == ()


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(STEADY_CASES))
def test_hip_dare_backward_matches_oracle(eng, case):
    w = STEADY_CASES[case]()
    P, info = eng.dare(w["T"], w["Z"], w["R"], w["H"], w["Q"])
    gP = np.random.default_rng(2).standard_normal(P.shape)
    g, info = eng.dare_backward(w["T"], w["Z"], w["R"], w["H"], w["Q"], P, gP)
    assert not info.cpu().numpy().any()
    for b in range(w["B"]):
        ref = so.steady_state_grad(w["T"][b], w["Z"][b], w["R"][b], w["H"][b], w["Q"][b], gP[b])
        for k in ("T", "Z", "R", "H", "Q"):
            got = g[k][b].cpu().numpy()
            assert np.max(np.abs(got - ref[k])) / (1.0 + np.max(np.abs(ref[k]))) < 1e-8, (case, b, k)
    # accumulate: the result is added to what the buffers hold
    g2 = {k: v.clone() for k, v in g.items()}
    eng.dare_backward(w["T"], w["Z"], w["R"], w["H"], w["Q"], P, gP, into=g2)
    for k in g:
        assert_allclose(g2[k].cpu().numpy(), 2.0 * g[k].cpu().numpy(), rtol=1e-12, atol=1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["structural_m10", "varmax_k3", "local_level"])
def test_hip_steady_state_logp_and_grad_matches_oracle(eng, case):
    """SteadyStateFilter.logp_and_grad (what build_graph's value-and-gradient node evaluates): kf_dare ->
    kf_filter_forward -> kf_filter_backward -> kf_dare_backward, against the oracle's Kalman + Riccati adjoints."""
    from pymc_statespace_amd import SteadyStateFilter
    from pymc_statespace_amd import op

    w = STEADY_CASES[case]()
    d = _one(w, 1)
    flt = SteadyStateFilter(engine=eng)
    lp, g = flt.logp_and_grad(w["y"][..., None], d["a0"][:, None], d["P0"], d["T"], d["Z"], d["R"], d["H"], d["Q"])
    rlp, rg = so.steady_state_logp_and_grad(w["y"], **d)
    assert abs(lp - rlp) < 1e-8 * max(1.0, abs(rlp))
    for k in ("a0", "T", "Z", "R", "H", "Q", "c"):
        a, b = np.asarray(g[k]), rg[k]
        assert a.reshape(b.shape).shape == b.shape
        assert np.max(np.abs(a.reshape(b.shape) - b)) / (1.0 + np.max(np.abs(b))) < 1e-7, (case, k)
    assert not np.any(g["P0"]) and not np.any(g["d"])
    # the six-output body: slot 0 of predicted_covs is the caller's P0 (:397)
    outs = op.perform_steady_state(w["y"][..., None], d["a0"][:, None], d["P0"], d["T"], d["Z"], d["R"], d["H"], d["Q"],
                                   np.zeros((w["m"], 1)), np.zeros((w["p"], 1)))
    ref = so.steady_state_filter(w["y"], **d)
    assert np.array_equal(outs[3][0], d["P0"]) and abs(float(outs[4]) - ref["logp"]) < 1e-8 * max(1.0, abs(ref["logp"]))
    assert_allclose(outs[0][..., 0], ref["filtered_states"], rtol=1e-7, atol=1e-8)


@pytest.mark.gpu
def test_hip_steady_state_filter_errors(eng):
    from pymc_statespace_amd import SteadyStateFilter

    w = W.structural(B=2, n=10)
    with pytest.raises(ValueError, match="time-invariant"):
        SteadyStateFilter(engine=eng)(w["y"], w["a0"], w["P0"], np.repeat(w["T"][:, None], 10, 1), w["Z"], w["R"],
                                       w["H"], w["Q"])
    T = w["T"].copy()
    T[1] *= 50.0  # explosive transition: the Riccati iteration cannot settle in 200 steps
    with pytest.raises(np.linalg.LinAlgError, match="Riccati"):
        SteadyStateFilter(engine=eng)(w["y"], w["a0"], w["P0"], T, w["Z"], w["R"], w["H"], w["Q"], max_iter=200)


@pytest.mark.gpu
@pytest.mark.parametrize("pmr,hzero", [((2, 4, 2), False), ((1, 10, 4), False), ((3, 7, 3), True)])
def test_hip_unconditional_simulation_matches_oracle(eng, pmr, hzero):
    from pymc_statespace_amd.simulation import unconditional_simulations

    p, m, r = pmr
    B, n_sims, n = 5, 70, 40  # 70 simulations: more than one wavefront of lanes per draw
    w = W.random_dense(B, 5, p, m, r, seed=21, h_zero=hzero)
    rng = np.random.default_rng(1)
    zs, zo = rng.standard_normal((B, n_sims, n, r)), rng.standard_normal((B, n_sims, n, p))
    x0 = rng.standard_normal((B, m))
    for use_x0 in (True, False):
        st, ob, info = eng.simulate_statespace(w["T"], w["Z"], w["R"], w["H"], w["Q"], n, n_sims,
                                               x0 if use_x0 else None, z_state=zs, z_obs=zo)
        assert not info.cpu().numpy().any()
        st, ob = st.cpu().numpy(), ob.cpu().numpy()
        assert st.shape == (B * n_sims, n, m) and ob.shape == (B * n_sims, n, p)
        for b in (0, B - 1):
            for j in (0, 63, 64, n_sims - 1):
                rs, ro = so.simulate_statespace(w["T"][b], w["Z"][b], w["R"][b], w["H"][b], w["Q"][b],
                                                zs[b, j], zo[b, j], x0[b] if use_x0 else None)
                assert_allclose(st[b * n_sims + j], rs, rtol=1e-12, atol=1e-12)
                assert_allclose(ob[b * n_sims + j], ro, rtol=1e-12, atol=1e-12)
    # host mirror of the reference function: shapes and the sample-major ordering
    s2, o2 = unconditional_simulations(w["T"], w["Z"], w["R"], w["H"], w["Q"], n_steps=12, n_simulations=3)
    assert s2.shape == (B * 3, 12, m) and o2.shape == (B * 3, 12, p)
    assert np.all(s2[:, 0] == 0.0) and np.all(o2[:, 0] == 0.0)
