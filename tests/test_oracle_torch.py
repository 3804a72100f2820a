"""The gradient oracle: (i) torch forward == pinned numpy forward, (ii) the hand-derived
adjoint recursion the HIP backward kernel implements == torch.autograd of the reference
operation sequence, (iii) a central finite difference sanity check."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import workloads as W
from oracle import kalman_adjoint_numpy as ka
from oracle import kalman_numpy as kn
from oracle import kalman_torch as kt

CASES = {
    "structural_m10": lambda: W.structural(B=3, n=25),
    "arma11": lambda: W.arma11(B=4, n=30),
    "varmax_m10_p5": lambda: W.varmax(B=2, n=12, order=(2, 0)),
    "dense_p3m4_fullH": lambda: W.random_dense(3, 15, 3, 4, 2, seed=5, h_diag=False),
    "dense_p2m6": lambda: W.random_dense(2, 12, 2, 6, 3, seed=6),
}


def _args(w, y=None):
    return (w["y"] if y is None else y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("ft", ["standard", "univariate"])
def test_torch_forward_matches_numpy(case, ft):
    w = CASES[case]()
    if ft == "univariate" and case == "dense_p3m4_fullH":
        pytest.skip("diag(H) only")
    lp, ll, _ = kt.logp_and_grads(*_args(w), filter_type=ft, mvn_const="exact")
    ref = kn.kalman_filter(*_args(w), filter_type=ft, mvn_const="exact")
    assert_allclose(ll, ref["ll_obs"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("case", sorted(CASES))
def test_torch_tapes_match_numpy(case):
    """The four tape outputs of the torch restatement (what the tape-cotangent gradients differentiate) are the pinned
    numpy oracle's, slot for slot."""
    import torch

    w = CASES[case]()
    tapes = {}
    ts = [torch.tensor(np.asarray(w[k], dtype=np.float64)) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    kt.kalman_logp(w["y"], *ts, mvn_const="exact", tapes=tapes)
    ref = kn.kalman_filter(*_args(w), mvn_const="exact")
    for k in ("filtered_states", "predicted_states", "filtered_covs", "predicted_covs"):
        assert_allclose(tapes[k].numpy(), ref[k], rtol=1e-9, atol=1e-10, err_msg=k)


def test_tape_cotangent_gradient_matches_finite_difference():
    """Central differences of a linear functional of the tapes against autograd through them."""
    w = W.random_dense(1, 8, 2, 3, 2, seed=11)
    rng = np.random.default_rng(3)
    gt = dict(filtered_states=rng.standard_normal((1, 8, 3)), predicted_covs=rng.standard_normal((1, 9, 3, 3)))
    gt["predicted_covs"] = 0.5 * (gt["predicted_covs"] + np.swapaxes(gt["predicted_covs"], -1, -2))
    _, _, g = kt.logp_and_grads(*_args(w), g_tapes=gt)

    def f(T):
        r = kn.kalman_filter(w["y"], w["a0"], w["P0"], T, w["Z"], w["R"], w["H"], w["Q"], mvn_const="exact")
        return float((gt["filtered_states"] * r["filtered_states"]).sum() + (gt["predicted_covs"] * r["predicted_covs"]).sum())

    for (i, j) in ((0, 0), (1, 2), (2, 1)):
        h = 1e-6
        Tp, Tm = w["T"].copy(), w["T"].copy()
        Tp[0, i, j] += h
        Tm[0, i, j] -= h
        fd = (f(Tp) - f(Tm)) / (2 * h)
        assert abs(fd - g["T"][0, i, j]) < 1e-6 * (1.0 + abs(fd)), (i, j, fd, g["T"][0, i, j])


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("missing", [False, True])
def test_hand_adjoint_matches_autograd(case, missing):
    w = CASES[case]()
    y = w["y"].copy()
    if missing:
        y[2] = np.nan
        y[5, 0] = np.nan
        y[-1] = np.nan
    rng = np.random.default_rng(0)
    B, n = w["B"], w["n"]
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, w["m"]))
    d = 0.1 * rng.standard_normal((B, w["p"]))
    _, _, gt = kt.logp_and_grads(*_args(w, y), c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll)
    gh = ka.logp_grads(*_args(w, y), c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll)
    for k in kt.NAMES:
        scale = 1.0 + np.max(np.abs(gt[k]))
        assert np.max(np.abs(gh[k] - gt[k])) / scale < 1e-10, k


def test_univariate_and_standard_gradients_agree_for_diagonal_H():
    """Same function when H is diagonal -> one backward kernel serves every filter type."""
    w = CASES["varmax_m10_p5"]()
    _, _, gs = kt.logp_and_grads(*_args(w), filter_type="standard")
    _, _, gu = kt.logp_and_grads(*_args(w), filter_type="univariate")
    for k in kt.NAMES:
        a, b = gs[k], gu[k]
        if k == "H":  # univariate only ever sees diag(H)
            a = a * np.eye(a.shape[-1])
        scale = 1.0 + np.max(np.abs(a))
        if k in ("P0", "Q"):  # entrywise partials of symmetric inputs may differ; symmetric parts agree
            a, b = a + np.swapaxes(a, -1, -2), b + np.swapaxes(b, -1, -2)
        assert np.max(np.abs(a - b)) / scale < 1e-9, k


def test_autograd_matches_finite_difference():
    w = W.structural(B=1, n=20)
    _, _, g = kt.logp_and_grads(*_args(w))

    def f(T):
        return kn.kalman_filter(w["y"], w["a0"], w["P0"], T, w["Z"], w["R"], w["H"], w["Q"])["logp"][0]

    for (i, j) in [(0, 1), (8, 9), (2, 5), (9, 8)]:
        h = 1e-6
        Tp, Tm = w["T"].copy(), w["T"].copy()
        Tp[0, i, j] += h
        Tm[0, i, j] -= h
        fd = (f(Tp) - f(Tm)) / (2 * h)
        assert_allclose(g["T"][0, i, j], fd, rtol=1e-5, atol=1e-6)


def test_time_varying_oracle_reduces_to_the_static_one():
    """kalman_torch with time_varying: matrices repeated along time give the static log-likelihood, and their gradient
    tapes sum to the static gradients (the oracle of tests/test_gpu_parity.py::test_time_varying_gradient_*)."""
    import numpy as np

    import workloads as W
    from oracle import kalman_torch as kt

    B, n, p, m, r = 2, 9, 2, 3, 2
    w = W.random_dense(B, n, p, m, r, seed=5)
    a = (w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    l0, _, g0 = kt.logp_and_grads(*a)
    rep = lambda x: np.repeat(x[:, None], n, axis=1)  # noqa: E731
    l1, _, g1 = kt.logp_and_grads(w["y"], w["a0"], w["P0"], rep(w["T"]), rep(w["Z"]), w["R"], rep(w["H"]), w["Q"],
                                  time_varying=("T", "Z", "H"))
    assert np.allclose(l0, l1, rtol=1e-13)
    for k in ("T", "Z", "H"):
        assert g1[k].shape == (B, n) + g0[k].shape[1:]
        assert np.allclose(g1[k].sum(1), g0[k], rtol=1e-10, atol=1e-12)
    for k in ("a0", "P0", "R", "Q"):
        assert np.allclose(g1[k], g0[k], rtol=1e-10, atol=1e-12)
