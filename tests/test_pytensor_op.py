"""Executable contract of the PyTensor Ops (pymc_statespace_amd/op.py) for an environment that HAS PyTensor.

PyTensor / PyMC are installed neither in the build image nor on the GPU box, so this module skips there
(``pytest.importorskip``); ``make_node`` / ``perform`` / ``L_op`` / ``infer_shape`` / the vectorize rule are
otherwise unexecuted -- a maintainer wiring ``FILTER_FACTORY`` runs this file first.
Reference being replaced: pymc_statespace/filters/kalman_filter.py:126-193, utils/pytensor_scipy.py:11-60
(the Op pattern), tests/test_kalman_filter.py:226-241 (the Nile parity case).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

pytensor = pytest.importorskip("pytensor")
pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402


@pytest.fixture(scope="module")
def nile():
    g = load_golden("nile_llt.npz")
    m, p = 2, 1
    vals = [g["y"][..., None], g["a0"][0][:, None], g["P0"][0], g["T"][0], g["Z"][0], g["R"][0], g["H"][0],
            g["Q"][0]]
    return g, vals, m, p


def _symbols():
    import pytensor.tensor as pt

    data = pt.tensor3("data")
    a0 = pt.matrix("a0")
    mats = [pt.matrix(nm) for nm in ("P0", "T", "Z", "R", "H", "Q")]
    return [data, a0] + mats


def test_filter_op_matches_statsmodels_golden(nile):
    from pymc_statespace_amd.filters import KalmanSmoother, StandardFilter

    g, vals, m, p = nile
    sym = _symbols()
    outs = StandardFilter().build_graph(*sym)
    assert len(outs) == 6
    sm = KalmanSmoother().build_graph(sym[3], sym[5], sym[7], outs[0], outs[2])
    f = pytensor.function(sym, list(outs) + list(sm), on_unused_input="ignore")
    res = f(*vals)
    n = vals[0].shape[0]
    assert [r.shape for r in res[:6]] == [(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,)]
    assert_allclose(res[4], g["llf_obs"].sum(), rtol=2e-9)
    assert_allclose(res[0][..., 0], g["filtered_state"], rtol=1e-7, atol=1e-7)
    assert_allclose(res[6][..., 0], g["smoothed_state"], rtol=1e-7, atol=1e-7)


def test_only_consumed_outputs_are_computed(nile, monkeypatch):
    from pymc_statespace_amd import op
    from pymc_statespace_amd.filters import StandardFilter

    g, vals, m, p = nile
    sym = _symbols()
    outs = StandardFilter().build_graph(*sym)
    seen = {}
    real = op.perform_filter

    def spy(*a, **k):
        seen["want"] = k.get("want")
        return real(*a, **k)

    monkeypatch.setattr(op, "perform_filter", spy)
    f = pytensor.function(sym, outs[4], on_unused_input="ignore")
    f(*vals)
    assert seen["want"] == (False, False, False, False, True, False)


def test_logp_op_value_and_gradient_from_one_node(nile):
    import pytensor.tensor as pt

    from oracle import kalman_torch as kt
    from pymc_statespace_amd.filters import StandardFilter

    g, vals, m, p = nile
    sym = _symbols()
    logp = StandardFilter().build_graph(*sym, outputs="logp")
    grads = pytensor.grad(logp, sym[1:])
    f = pytensor.function(sym, [logp] + grads)
    kalman_nodes = [nd for nd in f.maker.fgraph.toposort() if "Kalman" in type(nd.op).__name__]
    assert len(kalman_nodes) == 1, "value and gradient must come from ONE perform"
    res = f(*vals)
    lp, _, gt = kt.logp_and_grads(g["y"], g["a0"], g["P0"], g["T"], g["Z"], g["R"], g["H"], g["Q"],
                                  g_logp=np.ones(1))
    assert_allclose(res[0], lp[0], rtol=1e-11)
    for name, got in zip(("a0", "P0", "T", "Z", "R", "H", "Q"), res[1:]):
        assert_allclose(got.reshape(gt[name][0].shape), gt[name][0], rtol=1e-8, atol=1e-8, err_msg=name)


def test_vectorize_graph_reaches_the_batched_kernels(nile):
    import pytensor.tensor as pt
    from pytensor.graph.replace import vectorize_graph

    from pymc_statespace_amd.filters import StandardFilter

    g, vals, m, p = nile
    sym = _symbols()
    outs = StandardFilter().build_graph(*sym)
    batched_H = pt.tensor3("H_draws")
    new = vectorize_graph(outs[4], replace={sym[6]: batched_H})
    f = pytensor.function([s for s in sym if s is not sym[6]] + [batched_H], new)
    assert any(type(nd.op).__name__ == "KalmanFilterBatchedOp" for nd in f.maker.fgraph.toposort())
    Hs = np.stack([vals[6], 2.0 * vals[6], 0.5 * vals[6]])
    got = f(*(vals[:6] + vals[7:]), Hs)
    assert got.shape == (3,)
    assert_allclose(got[0], g["llf_obs"].sum(), rtol=2e-9)
