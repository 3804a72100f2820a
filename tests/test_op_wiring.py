"""Graph-level wiring of the PyTensor Ops (pymc_statespace_amd/op.py) that needs no GPU: arity and types of every
``make_node``, ``connection_pattern`` shapes, ``__props__`` equality, which nodes ``grad`` / ``vectorize_graph``
create.  Runs against the real PyTensor when installed, else against the test-only stand-in
(tests/_pytensor_shim).  The numeric half (perform / L_op values) is tests/test_pytensor_op.py, on the GPU.
Reference pattern: pymc_statespace/utils/pytensor_scipy.py:11-60; caller core/statespace.py:146-197."""
import pytest

from conftest import use_pytensor_or_standin

pytensor = use_pytensor_or_standin()


def _symbols():
    import pytensor.tensor as pt

    return [pt.tensor3("data"), pt.matrix("a0")] + [pt.matrix(nm) for nm in ("P0", "T", "Z", "R", "H", "Q")]


def _op_names(f):
    return [type(nd.op).__name__ for nd in f.maker.fgraph.toposort()]


def test_every_op_class_builds_a_node_with_the_reference_arity():
    import pytensor.tensor as pt

    from pymc_statespace_amd import op

    ops = op._build_ops()
    sym = _symbols()
    c, d = pt.matrix("c"), pt.matrix("d")
    ins = sym + [c, d]
    six = [3, 3, 3, 3, 0, 1]       # kalman_filter.py:184-191: ndim of the six outputs
    for name in ("KalmanFilterOp", "SteadyStateFilterOp"):
        node = ops[name]().make_node(*ins)
        assert len(node.inputs) == 10 and [o.type.ndim for o in node.outputs] == six, name
        pat = node.op.connection_pattern(node)
        assert len(pat) == 10 and all(len(row) == 6 for row in pat) and not any(pat[0])
    for name in ("KalmanLogpOp", "SteadyStateLogpOp"):
        node = ops[name]().make_node(*ins)
        assert [o.type.ndim for o in node.outputs] == [0] + [v.type.ndim for v in ins[1:]], name
        pat = node.op.connection_pattern(node)
        assert len(pat) == 10 and all(len(row) == 10 for row in pat) and not any(pat[0])
        assert all(row == [True] + [False] * 9 for row in pat[1:])
    node = ops["KalmanFilterBatchedOp"]().make_node(sym[0], *[pt.tensor3(f"b{k}") for k in range(9)])
    assert [o.type.ndim for o in node.outputs] == [4, 4, 4, 4, 1, 2]
    pat = node.op.connection_pattern(node)
    assert len(pat) == 10 and all(len(row) == 6 for row in pat) and not any(pat[0])
    node = ops["KalmanLogpBatchedOp"]().make_node(sym[0], *[pt.tensor3(f"b{k}") for k in range(9)])
    assert [o.type.ndim for o in node.outputs] == [1] + [3] * 9
    pat = node.op.connection_pattern(node)
    assert not any(pat[0]) and all(row == [True] + [False] * 9 for row in pat[1:])
    with pytest.raises(TypeError):
        ops["KalmanLogpBatchedOp"]().make_node(*ins)       # per-draw (2-D) inputs are not a batch
    node = ops["KalmanSmootherOp"]().make_node(sym[3], sym[5], sym[7], pt.tensor3("fs"), pt.tensor3("fP"))
    assert len(node.inputs) == 5 and [o.type.ndim for o in node.outputs] == [3, 3]
    node = ops["KalmanFilterGradOp"]().make_node(*ins, pt.tensor3("pa"), pt.tensor3("pP"), pt.scalar("g"), pt.vector("gl"))
    assert len(node.inputs) == 14 and [o.type.ndim for o in node.outputs] == [v.type.ndim for v in ins[1:]]
    node = ops["SteadyStateGradOp"]().make_node(*ins, pt.scalar("g"), pt.vector("gl"))
    assert len(node.inputs) == 12 and len(node.outputs) == 9


def test_props_make_equal_ops_equal():
    from pymc_statespace_amd import op

    ops = op._build_ops()
    K = ops["KalmanFilterOp"]
    assert K("cholesky", "exact", True) == K("cholesky", "exact", True)
    assert hash(K("cholesky", "exact", True)) == hash(K("cholesky", "exact", True))
    assert K("cholesky") != K("standard")
    assert K(want=op.want_mask(("log_likelihood",))) != K()
    assert op.want_mask("all") == (True,) * 6
    assert op.want_mask(("filtered_states", "ll_obs")) == (True, False, False, False, False, True)
    with pytest.raises(ValueError):
        op.want_mask(("smoothed_states",))


def test_grad_and_vectorize_create_the_intended_nodes():
    import pytensor.tensor as pt
    from pytensor.graph.replace import vectorize_graph

    from pymc_statespace_amd.filters import StandardFilter, SteadyStateFilter

    sym = _symbols()
    logp = StandardFilter().build_graph(*sym, outputs="logp")
    f = pytensor.function(sym, [logp] + pytensor.grad(logp, sym[1:]))
    assert sum("Kalman" in nm for nm in _op_names(f)) == 1          # value and gradient from ONE node
    outs = StandardFilter().build_graph(*sym)
    f = pytensor.function(sym, pytensor.grad(outs[4], sym[3]), on_unused_input="ignore")
    assert "KalmanFilterGradOp" in _op_names(f) and "KalmanFilterOp" in _op_names(f)
    # round 6: a loss on a tape output is differentiated too (kf_filter_backward_tapes behind the same gradient node, which then
    # takes the four tape cotangents as well -- a zero-size placeholder for the unconnected ones)
    f = pytensor.function(sym, pytensor.grad(outs[0].sum() + outs[3].sum(), sym[3]), on_unused_input="ignore")
    gnodes = [nd for nd in f.maker.fgraph.toposort() if type(nd.op).__name__ == "KalmanFilterGradOp"]
    assert len(gnodes) == 1 and len(gnodes[0].inputs) == 18
    lean = StandardFilter().build_graph(*sym, outputs=("filtered_states",))   # (no predicted tapes on the node: recomputed inside)
    f = pytensor.function(sym, pytensor.grad(lean[0].sum(), sym[3]), on_unused_input="ignore")
    assert "KalmanFilterGradOp" in _op_names(f)
    sm_op_outs = StandardFilter().build_graph(*sym)
    from pymc_statespace_amd.filters import KalmanSmoother
    sm = KalmanSmoother().build_graph(sym[3], sym[5], sym[7], sm_op_outs[0], sm_op_outs[2])
    with pytest.raises(Exception, match="(?i)not implemented|undefined|null|not differentiated"):   # the smoother still is not
        pytensor.grad(sm[0].sum(), sym[3])
    few = StandardFilter().build_graph(*sym, outputs=("log_likelihood", "filtered_states"))
    f = pytensor.function(sym, pytensor.grad(few[4], sym[3]), on_unused_input="ignore")
    assert "KalmanLogpOp" in _op_names(f) and "KalmanFilterGradOp" not in _op_names(f)
    bH = pt.tensor3("H_draws")
    f = pytensor.function([s for s in sym if s is not sym[6]] + [bH], vectorize_graph(outs[4], replace={sym[6]: bH}))
    assert "KalmanFilterBatchedOp" in _op_names(f) and "KalmanFilterOp" not in _op_names(f)
    ss = SteadyStateFilter().build_graph(*sym)
    f = pytensor.function(sym, pytensor.grad(ss[4], [sym[3], sym[7]]), on_unused_input="ignore")
    assert "SteadyStateGradOp" in _op_names(f)
    with pytest.raises(ValueError):
        StandardFilter().build_graph(sym[0], sym[1], sym[2], pt.vector("T1"), *sym[4:])   # filters/utilities.py:16-18


def test_graphs_evaluate_end_to_end_with_oracle_backed_bodies(monkeypatch):
    """function() / grad() through make_node -> perform -> infer_shape -> L_op on the CPU: the numeric bodies (GPU) are
    replaced by the numpy / torch oracle, so that arity, output order, ranks (the 0-d log-likelihood!), shapes and the
    gradient plumbing of the Op classes are checked here too, not only on the GPU box."""
    import numpy as np
    import pytensor.tensor as pt
    from numpy.testing import assert_allclose

    from conftest import load_golden
    from oracle import kalman_numpy as kn
    from oracle import kalman_torch as kt
    from pymc_statespace_amd import op
    from pymc_statespace_amd.filters import KalmanSmoother, StandardFilter

    def col(x):
        x = np.asarray(x, dtype=np.float64)
        return x[:, 0] if x.ndim == 2 and x.shape[1] == 1 else x

    def fake_filter(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard", mvn_const="reference", h_diagonal=False,
                    engine=None, want=(True,) * 6):
        r = kn.kalman_filter(np.asarray(data)[..., 0], col(a0)[None], P0[None], T[None], Z[None], R[None], H[None], Q[None],
                             c=col(c)[None], d=col(d)[None], filter_type=filter_type, mvn_const=mvn_const)
        outs = [r["filtered_states"][0][..., None], r["predicted_states"][0][..., None], r["filtered_covs"][0],
                r["predicted_covs"][0], np.asarray(r["logp"][0]), r["ll_obs"][0]]
        return [o if w else None for o, w in zip(outs, want)]

    def fake_logp_and_grad(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard", mvn_const="reference",
                           h_diagonal=False, engine=None):
        lp, _, g = kt.logp_and_grads(np.asarray(data)[..., 0], col(a0)[None], P0[None], T[None], Z[None], R[None], H[None],
                                     Q[None], c=col(c)[None], d=col(d)[None], filter_type=filter_type, mvn_const=mvn_const)
        out = {k: g[k][0] for k in op.GRAD_INPUT_NAMES}
        for k in ("a0", "c", "d"):
            out[k] = out[k][..., None]
        return [np.asarray(lp[0])] + [out[k] for k in op.GRAD_INPUT_NAMES]

    def fake_smoother(T, R, Q, fs, fP, engine=None):
        sa, sP = kn.kalman_smoother(T[None], R[None], Q[None], np.asarray(fs)[None, ..., 0], np.asarray(fP)[None])
        return [sa[0][..., None], sP[0]]

    monkeypatch.setattr(op, "perform_filter", fake_filter)
    monkeypatch.setattr(op, "perform_logp_and_grad", fake_logp_and_grad)
    def fake_fused(data, T, Z, R, H, Q, c, d, fs, fP, ps, pP, filter_type="standard", engine=None, batched=False):
        assert not batched and np.asarray(ps).shape[0] == np.asarray(fs).shape[0] + 1   # the filter node's predicted tapes came along
        return fake_smoother(T, R, Q, fs, fP)

    monkeypatch.setattr(op, "perform_smoother", fake_smoother)
    monkeypatch.setattr(op, "perform_smoother_fused", fake_fused)
    g = load_golden("nile_llt.npz")
    vals = [g["y"][..., None], g["a0"][0][:, None], g["P0"][0], g["T"][0], g["Z"][0], g["R"][0], g["H"][0], g["Q"][0]]
    sym = _symbols()
    outs = StandardFilter().build_graph(*sym)
    sm = KalmanSmoother().build_graph(sym[3], sym[5], sym[7], outs[0], outs[2])
    f = pytensor.function(sym, list(outs) + list(sm), on_unused_input="ignore")
    assert "KalmanSmootherFusedOp" in _op_names(f) and "KalmanSmootherOp" not in _op_names(f)   # behind this library's filter node
    res = f(*vals)
    n, m = vals[0].shape[0], 2
    assert [r.shape for r in res] == [(n, m, 1), (n + 1, m, 1), (n, m, m), (n + 1, m, m), (), (n,), (n, m, 1), (n, m, m)]
    assert_allclose(res[6][..., 0], g["smoothed_state"][0], rtol=1e-7, atol=1e-7)
    # tapes of any other origin (free inputs), or fuse=False: the reference's recursion on the filtered tapes
    fs_in, fP_in = pt.tensor3("fs"), pt.tensor3("fP")
    sm2 = KalmanSmoother().build_graph(sym[3], sym[5], sym[7], fs_in, fP_in)
    f2 = pytensor.function([sym[3], sym[5], sym[7], fs_in, fP_in], list(sm2))
    assert "KalmanSmootherOp" in _op_names(f2) and "KalmanSmootherFusedOp" not in _op_names(f2)
    r2 = f2(vals[3], vals[5], vals[7], res[0], res[2])
    assert_allclose(r2[0], res[6], rtol=1e-12)
    assert_allclose(r2[1], res[7], rtol=1e-12)
    sm3 = op.kalman_smoother_graph(sym[3], sym[5], sym[7], outs[0], outs[2], fuse=False)
    assert "KalmanSmootherFusedOp" not in _op_names(pytensor.function(sym, list(sm3), on_unused_input="ignore"))
    # another T than the filter node's: the predicted tapes belong to the FILTER's model -- no fusion
    T_other = pt.matrix("T_other")
    sm5 = op.kalman_smoother_graph(T_other, sym[5], sym[7], outs[0], outs[2])
    names5 = _op_names(pytensor.function(sym + [T_other], list(sm5), on_unused_input="ignore"))
    assert "KalmanSmootherOp" in names5 and "KalmanSmootherFusedOp" not in names5
    # a filter node that does not compute its predicted tapes cannot feed the fused smoother
    few2 = StandardFilter().build_graph(*sym, outputs=("filtered_states", "filtered_covariances"))
    sm4 = KalmanSmoother().build_graph(sym[3], sym[5], sym[7], few2[0], few2[2])
    assert "KalmanSmootherFusedOp" not in _op_names(pytensor.function(sym, list(sm4), on_unused_input="ignore"))
    assert_allclose(res[4], g["llf_obs"].sum(), rtol=2e-9)
    logp = StandardFilter().build_graph(*sym, outputs="logp")
    f = pytensor.function(sym, [logp] + pytensor.grad(logp, sym[1:]))
    res = f(*vals)
    assert res[0].shape == () and [r.shape for r in res[1:]] == [v.shape for v in vals[1:]]
    few = StandardFilter().build_graph(*sym, outputs=("log_likelihood",))
    f = pytensor.function(sym, [few[4], few[2], pytensor.grad(few[4], sym[3])], on_unused_input="ignore")
    lp, fP, gT = f(*vals)
    assert fP.shape == (0, 0, 0) and gT.shape == (m, m)
    assert_allclose(gT, res[3], rtol=1e-12)
    # time-varying T keeps its 3-D shape in the gradient
    Tt = pt.tensor3("Tt")
    lp_tv = StandardFilter().build_graph(sym[0], sym[1], sym[2], Tt, *sym[4:], outputs="logp")
    assert pytensor.grad(lp_tv, Tt).type.ndim == 3


def test_the_batch_axis_reaches_one_value_and_gradient_node(monkeypatch):
    """Row (b2) of the scope table: ``vectorize_graph`` over the sampler's node (KalmanLogpOp) and the gradient of a batched
    filter's log-likelihoods both end in ONE KalmanLogpBatchedOp node = one kf_filter_value_and_grad call for all draws
    (reference: the per-draw loop of core/statespace.py:230-236 around the Potential of :174).  Numeric bodies replaced by the
    torch oracle; the GPU half is tests/test_pytensor_op.py."""
    import numpy as np
    import pytensor.tensor as pt
    from numpy.testing import assert_allclose
    from pytensor.graph.replace import vectorize_graph

    from conftest import load_golden
    from oracle import kalman_torch as kt
    from pymc_statespace_amd import op
    from pymc_statespace_amd.filters import StandardFilter

    calls = []

    def fake_batched(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard", mvn_const="reference", h_diagonal=False,
                     engine=None):
        calls.append(np.asarray(P0).shape[0])
        lp, _, g = kt.logp_and_grads(np.asarray(data)[..., 0], a0[..., 0], P0, T, Z, R, H, Q, c=c[..., 0], d=d[..., 0],
                                     filter_type=filter_type, mvn_const=mvn_const)
        out = {k: g[k] for k in op.GRAD_INPUT_NAMES}
        for k in ("a0", "c", "d"):
            out[k] = out[k][..., None]
        return [lp] + [out[k] for k in op.GRAD_INPUT_NAMES]

    def fake_filter_batched(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard", mvn_const="reference", h_diagonal=False,
                            engine=None, want=(True,) * 6):
        # (the stand-in evaluates the forward node for the SHAPE of its log-likelihoods; real PyTensor infers it)
        from oracle import kalman_numpy as kn
        r = kn.kalman_filter(np.asarray(data)[..., 0], a0[..., 0], P0, T, Z, R, H, Q, c=c[..., 0], d=d[..., 0],
                             filter_type=filter_type, mvn_const=mvn_const)
        outs = [r["filtered_states"][..., None], r["predicted_states"][..., None], r["filtered_covs"], r["predicted_covs"],
                r["logp"], r["ll_obs"]]
        return [o if w or k == 4 else None for k, (o, w) in enumerate(zip(outs, want))]

    monkeypatch.setattr(op, "perform_logp_and_grad_batched", fake_batched)
    monkeypatch.setattr(op, "perform_filter_batched", fake_filter_batched)
    g = load_golden("nile_llt.npz")
    B = 5
    scale = np.linspace(0.5, 2.0, B)
    vals = [g["y"][..., None], g["a0"][0][:, None], g["P0"][0], g["T"][0], g["Z"][0], g["R"][0], g["H"][0], g["Q"][0]]
    Hs, Qs = scale[:, None, None] * vals[6], scale[::-1, None, None] * vals[7]
    sym = _symbols()
    # (a) vectorize_graph over the per-draw logp node: H and Q gain a draw axis, everything else is broadcast
    logp = StandardFilter().build_graph(*sym, outputs="logp")
    bH, bQ = pt.tensor3("H_draws"), pt.tensor3("Q_draws")
    blogp = vectorize_graph(logp, replace={sym[6]: bH, sym[7]: bQ})
    assert blogp.type.ndim == 1
    w = pt.vector("w")
    cost = (w * blogp).sum()
    ins = sym[:6] + [bH, bQ, w]
    f = pytensor.function(ins, [blogp] + pytensor.grad(cost, [bH, bQ, sym[3]]))
    names = _op_names(f)
    assert names.count("KalmanLogpBatchedOp") == 1 and "KalmanLogpOp" not in names
    wv = np.arange(1.0, B + 1.0)
    lp, gH, gQ, gT = f(*vals[:6], Hs, Qs, wv)
    assert calls == [B], "value and gradients of all draws from ONE batched call"
    ref_lp, _, ref_g = kt.logp_and_grads(g["y"], np.repeat(g["a0"], B, 0), np.repeat(g["P0"], B, 0), np.repeat(g["T"], B, 0),
                                         np.repeat(g["Z"], B, 0), np.repeat(g["R"], B, 0), Hs, Qs, g_logp=wv)
    assert lp.shape == (B,) and gH.shape == (B, 1, 1) and gQ.shape == (B, 2, 2)
    assert_allclose(lp, ref_lp, rtol=1e-12)
    assert_allclose(gH, ref_g["H"], rtol=1e-10)
    assert_allclose(gQ, ref_g["Q"], rtol=1e-10)
    assert_allclose(gT, ref_g["T"].sum(0), rtol=1e-9, atol=1e-9)   # T was broadcast to all draws: its gradient sums over them
    # (a2) T batched as well: the default intercepts zeros((T.shape[-1], 1)) hang off a Shape_i node whose input gains the draw axis
    calls.clear()
    bT = pt.tensor3("T_draws")
    blogp2 = vectorize_graph(logp, replace={sym[3]: bT, sym[6]: bH, sym[7]: bQ})
    f = pytensor.function([sym[0], sym[1], sym[2], bT, sym[4], sym[5], bH, bQ], blogp2)
    assert _op_names(f).count("KalmanLogpBatchedOp") == 1
    assert_allclose(f(vals[0], vals[1], vals[2], np.repeat(vals[3][None], B, 0), vals[4], vals[5], Hs, Qs), ref_lp, rtol=1e-12)
    assert calls == [B]
    # (b) the batched filter's log-likelihood differentiates through the same node
    calls.clear()
    sym3 = [sym[0]] + [pt.tensor3(nm) for nm in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    outs = StandardFilter().build_graph(*sym3, batched=True)
    assert [o.type.ndim for o in outs] == [4, 4, 4, 4, 1, 2]
    gT = pytensor.grad(outs[4].sum(), sym3[3])
    f = pytensor.function(sym3, gT, on_unused_input="ignore")
    assert "KalmanLogpBatchedOp" in _op_names(f)
    rep = lambda x: np.repeat(x[None], B, 0)  # noqa: E731
    got = f(vals[0], rep(vals[1]), rep(vals[2]), rep(vals[3]), rep(vals[4]), rep(vals[5]), Hs, Qs)
    _, _, ref_g1 = kt.logp_and_grads(g["y"], np.repeat(g["a0"], B, 0), np.repeat(g["P0"], B, 0), np.repeat(g["T"], B, 0),
                                      np.repeat(g["Z"], B, 0), np.repeat(g["R"], B, 0), Hs, Qs)
    assert got.shape == (B, 2, 2) and calls == [B]
    assert_allclose(got, ref_g1["T"], rtol=1e-9, atol=1e-9)
    # (round 6) a loss on a tape of the batched node: the batched gradient node with the four tape cotangents
    f2 = pytensor.function(sym3, pytensor.grad(outs[0].sum(), sym3[3]), on_unused_input="ignore")
    gn = [nd for nd in f2.maker.fgraph.toposort() if type(nd.op).__name__ == "KalmanFilterBatchedGradOp"]
    assert len(gn) == 1 and len(gn[0].inputs) == 18
    # (c) built directly: logp (B,) from the batched node
    blp = StandardFilter().build_graph(*sym3, outputs="logp", batched=True)
    f = pytensor.function(sym3, blp)
    assert _op_names(f).count("KalmanLogpBatchedOp") == 1
    assert_allclose(f(vals[0], rep(vals[1]), rep(vals[2]), rep(vals[3]), rep(vals[4]), rep(vals[5]), Hs, Qs), ref_lp, rtol=1e-12)


def test_vectorize_graph_over_the_smoother_reaches_one_batched_call(monkeypatch):
    """VERDICT round 4, missing #2: ``vectorize_graph`` over a smoother node (sample_conditional_posterior's per-draw loop,
    reference core/statespace.py:283-302 over :176-197) ends in ONE batched node -- KalmanSmootherBatchedOp for tapes of any
    origin, KalmanSmootherFusedOp(batched=True) behind this library's filter node -- and its body is called once for all draws.
    Numeric bodies replaced by the numpy oracle; the GPU half is tests/test_pytensor_op.py."""
    import numpy as np
    import pytensor.tensor as pt
    from numpy.testing import assert_allclose
    from pytensor.graph.replace import vectorize_graph

    from conftest import load_golden
    from oracle import kalman_numpy as kn
    from pymc_statespace_amd import op
    from pymc_statespace_amd.filters import KalmanSmoother, StandardFilter

    calls = []

    def fake_smoother_batched(T, R, Q, fs, fP, engine=None):
        calls.append(("rts", np.asarray(T).shape[0]))
        sa, sP = kn.kalman_smoother(T, R, Q, np.asarray(fs)[..., 0], np.asarray(fP))
        return [sa.reshape(np.asarray(fs).shape), sP]

    def fake_fused(data, T, Z, R, H, Q, c, d, fs, fP, ps, pP, filter_type="standard", engine=None, batched=False):
        assert batched and np.asarray(data).ndim == 3 and all(np.asarray(x).ndim == 3 for x in (T, Z, R, H, Q, d))
        assert np.asarray(ps).shape[:2] == (np.asarray(T).shape[0], np.asarray(fs).shape[1] + 1)
        calls.append(("fused", np.asarray(T).shape[0]))
        sa, sP = kn.kalman_smoother(T, R, Q, np.asarray(fs)[..., 0], np.asarray(fP))
        return [sa.reshape(np.asarray(fs).shape), sP]

    def fake_filter_batched(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard", mvn_const="reference", h_diagonal=False,
                            engine=None, want=(True,) * 6):
        r = kn.kalman_filter(np.asarray(data)[..., 0], a0[..., 0], P0, T, Z, R, H, Q, c=c[..., 0], d=d[..., 0],
                             filter_type=filter_type, mvn_const=mvn_const)
        return [r["filtered_states"][..., None], r["predicted_states"][..., None], r["filtered_covs"], r["predicted_covs"],
                r["logp"], r["ll_obs"]]

    monkeypatch.setattr(op, "perform_smoother_batched", fake_smoother_batched)
    monkeypatch.setattr(op, "perform_smoother_fused", fake_fused)
    monkeypatch.setattr(op, "perform_filter_batched", fake_filter_batched)
    g = load_golden("nile_llt.npz")
    B = 4
    scale = np.linspace(0.5, 2.0, B)
    vals = [g["y"][..., None], g["a0"][0][:, None], g["P0"][0], g["T"][0], g["Z"][0], g["R"][0], g["H"][0], g["Q"][0]]
    Hs, Qs = scale[:, None, None] * vals[6], scale[::-1, None, None] * vals[7]
    rep = lambda x: np.repeat(np.asarray(x)[None], B, 0)  # noqa: E731
    flt = kn.kalman_filter(g["y"], rep(g["a0"][0]), rep(vals[2]), rep(vals[3]), rep(vals[4]), rep(vals[5]), Hs, Qs)
    ref_a, ref_P = kn.kalman_smoother(rep(vals[3]), rep(vals[5]), Qs, flt["filtered_states"], flt["filtered_covs"])
    sym = _symbols()
    bH, bQ = pt.tensor3("H_draws"), pt.tensor3("Q_draws")
    # (a) behind the filter node: filter AND smoother gain the draw axis, one batched call each
    outs = StandardFilter().build_graph(*sym)
    sm = KalmanSmoother().build_graph(sym[3], sym[5], sym[7], outs[0], outs[2])
    bsm = vectorize_graph(list(sm), replace={sym[6]: bH, sym[7]: bQ})
    assert [v.type.ndim for v in bsm] == [4, 4]
    f = pytensor.function(sym[:6] + [bH, bQ], bsm)
    names = _op_names(f)
    assert names.count("KalmanSmootherFusedOp") == 1 and names.count("KalmanFilterBatchedOp") == 1
    assert "KalmanSmootherOp" not in names and "KalmanFilterOp" not in names
    sa, sP = f(*vals[:6], Hs, Qs)
    assert calls == [("fused", B)], calls
    assert_allclose(sa[..., 0], ref_a.reshape(B, -1, 2), rtol=1e-12)
    assert_allclose(sP, ref_P, rtol=1e-12)
    # (b) tapes of free origin: the plain smoother node maps to KalmanSmootherBatchedOp
    calls.clear()
    fs_in, fP_in = pt.tensor3("fs"), pt.tensor3("fP")
    sm2 = KalmanSmoother().build_graph(sym[3], sym[5], sym[7], fs_in, fP_in)
    bfs, bfP = pt.tensor4("fs_draws"), pt.tensor4("fP_draws")
    bsm2 = vectorize_graph(list(sm2), replace={fs_in: bfs, fP_in: bfP, sym[7]: bQ})
    f = pytensor.function([sym[3], sym[5], bQ, bfs, bfP], bsm2)
    assert _op_names(f).count("KalmanSmootherBatchedOp") == 1 and "KalmanSmootherOp" not in _op_names(f)
    sa2, sP2 = f(vals[3], vals[5], Qs, flt["filtered_states"].reshape(B, -1, 2, 1), flt["filtered_covs"])
    assert calls == [("rts", B)], calls
    assert_allclose(sa2, sa, rtol=1e-12)
    assert_allclose(sP2, sP, rtol=1e-12)
    # (c) built directly on batched inputs
    sym3 = [sym[0]] + [pt.tensor3(nm) for nm in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    outs3 = StandardFilter().build_graph(*sym3, batched=True)
    sm3 = KalmanSmoother().build_graph(sym3[3], sym3[5], sym3[7], outs3[0], outs3[2])
    assert [v.type.ndim for v in sm3] == [4, 4]
    f = pytensor.function(sym3, list(sm3), on_unused_input="ignore")
    assert _op_names(f).count("KalmanSmootherFusedOp") == 1


def test_batched_filter_differentiates_ll_obs(monkeypatch):
    """a cotangent on the BATCHED filter's ll_obs (a burn-in mask after vectorize_graph) used to end in
    grad_not_implemented while the per-draw node differentiates it.  It now goes through KalmanFilterBatchedGradOp -- one
    kf_filter_backward call over the node's predicted tapes; body replaced by the torch oracle here."""
    import numpy as np
    import pytensor.tensor as pt
    import torch
    from numpy.testing import assert_allclose

    from conftest import load_golden
    from oracle import kalman_numpy as kn
    from oracle import kalman_torch as kt
    from pymc_statespace_amd import op
    from pymc_statespace_amd.filters import StandardFilter

    calls = []

    def fake_filter_batched(data, a0, P0, T, Z, R, H, Q, c, d, filter_type="standard", mvn_const="reference", h_diagonal=False,
                            engine=None, want=(True,) * 6):
        r = kn.kalman_filter(np.asarray(data)[..., 0], a0[..., 0], P0, T, Z, R, H, Q, c=c[..., 0], d=d[..., 0],
                             filter_type=filter_type, mvn_const=mvn_const)
        return [r["filtered_states"][..., None], r["predicted_states"][..., None], r["filtered_covs"], r["predicted_covs"],
                r["logp"], r["ll_obs"]]

    def fake_grad_batched(data, a0, P0, T, Z, R, H, Q, c, d, ps, pP, g_logp, g_ll, filter_type="standard",
                          mvn_const="reference", h_diagonal=False, engine=None):
        calls.append(np.asarray(P0).shape[0])
        assert np.asarray(ps).shape[1] == np.asarray(data).shape[0] + 1 and np.asarray(g_ll).shape == (np.asarray(P0).shape[0], np.asarray(data).shape[0])
        t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float64, requires_grad=True)  # noqa: E731
        ins = [t(a0[..., 0]), t(P0), t(T), t(Z), t(R), t(H), t(Q), t(c[..., 0]), t(d[..., 0])]
        lp, ll = kt.kalman_logp(torch.tensor(np.asarray(data)[..., 0]), *ins[:7], c=ins[7], d=ins[8], filter_type=filter_type,
                                mvn_const=mvn_const)
        cost = (torch.tensor(np.asarray(g_logp)) * lp).sum() + (torch.tensor(np.asarray(g_ll)) * ll).sum()
        gs = torch.autograd.grad(cost, ins, allow_unused=True)
        out = [np.zeros(tuple(i.shape)) if g_ is None else g_.numpy() for g_, i in zip(gs, ins)]
        for k in (0, 7, 8):
            out[k] = out[k][..., None]
        return out

    monkeypatch.setattr(op, "perform_filter_batched", fake_filter_batched)
    monkeypatch.setattr(op, "perform_filter_grad_batched", fake_grad_batched)
    g = load_golden("nile_llt.npz")
    B, n = 3, g["y"].shape[0]
    rep = lambda x: np.repeat(np.asarray(x)[None], B, 0)  # noqa: E731
    scale = np.linspace(0.5, 2.0, B)
    Hs, Qs = scale[:, None, None] * g["H"][0], scale[::-1, None, None] * g["Q"][0]
    sym3 = [pt.tensor3("data")] + [pt.tensor3(nm) for nm in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    outs = StandardFilter().build_graph(*sym3, batched=True)
    burn = np.zeros((B, n)); burn[:, 10:] = 1.0
    cost = (pt.as_tensor_variable(burn) * outs[5]).sum()
    gH, gQ = pytensor.grad(cost, [sym3[6], sym3[7]])
    f = pytensor.function(sym3, [gH, gQ], on_unused_input="ignore")
    assert "KalmanFilterBatchedGradOp" in _op_names(f)
    vals = [g["y"][..., None], rep(g["a0"][0][:, None]), rep(g["P0"][0]), rep(g["T"][0]), rep(g["Z"][0]), rep(g["R"][0]), Hs, Qs]
    got_H, got_Q = f(*vals)
    assert calls == [B] and got_H.shape == (B, 1, 1) and got_Q.shape == (B, 2, 2)
    # reference: autograd of the masked sum, draw by draw
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float64, requires_grad=True)  # noqa: E731
    Ht, Qt = t(Hs), t(Qs)
    _, ll = kt.kalman_logp(torch.tensor(g["y"]), torch.tensor(rep(g["a0"][0])), torch.tensor(rep(g["P0"][0])),
                           torch.tensor(rep(g["T"][0])), torch.tensor(rep(g["Z"][0])), torch.tensor(rep(g["R"][0])), Ht, Qt,
                           )
    rH, rQ = torch.autograd.grad((torch.tensor(burn) * ll).sum(), [Ht, Qt])
    assert_allclose(got_H, rH.numpy(), rtol=1e-10)
    assert_allclose(got_Q, rQ.numpy(), rtol=1e-10, atol=1e-12)
    # without the predicted tapes on the node the cotangent cannot be served
    few = StandardFilter().build_graph(*sym3, batched=True, outputs=("ll_obs",))
    with pytest.raises(Exception, match="(?i)not implemented|undefined|null|predicted"):
        pytensor.grad((pt.as_tensor_variable(burn) * few[5]).sum(), sym3[6])


def test_which_recursion_the_fused_smoother_node_takes():
    """engine.smoother_from_filter_applies (re-exported by op): the smoother from the filter's predicted tapes for p == 1, static matrices
    (and a static d) and every m <= 1024 -- lane kernel, block tiles, 16 x 16 MFMA tiles (round 6), workspace --; m == 64 alone keeps
    round 5's rank rule (12 r <= m + 11: its on-chip form needs five wavefronts on four SIMDs); else kf_smoother."""
    from pymc_statespace_amd.op import smoother_from_filter_applies as f

    static = [2, 2, 2, 2, 2]
    assert not f(2, 10, static) and not f(1, 10, [3, 2, 2, 2, 2]) and not f(1, 10, [2, 2, 2, 3, 2])   # p > 1; a time-varying T; R
    assert all(f(1, m, static) for m in range(1, 64)) and all(f(1, m, static, r=m) for m in (16, 20, 32, 48, 63, 65, 79, 80, 367, 1024))
    assert f(1, 10, static + [1]) and not f(1, 10, static + [2])   # a sixth entry: d -- (p,) static, (n,p) time-varying
    assert not f(1, 1025, static) and not f(1, 0, static)
    assert not f(1, 64, static) and not f(1, 64, static, r=64) and not f(1, 64, static, r=7) and f(1, 64, static, r=6) and f(1, 64, static, r=1)


def test_fused_smoother_body_keeps_the_reference_recursion_for_a_nonzero_c_or_a_time_varying_d():
    """ADVICE round 5 (high / medium): the reference's smoother predicts with a_hat = T a_f -- no state intercept c
    (kalman_smoother.py:99-104) -- while the filter's predicted tapes contain c, so kf_smoother_from_filter on them differs by O(c);
    and a time-varying d (n,p,1) is not something kf_smoother_from_filter takes.  perform_smoother_fused must hand both to kf_smoother
    on the filtered tapes, whatever m is; c == 0 with a static d goes to the predicted-tape recursion."""
    import numpy as np
    import torch

    from pymc_statespace_amd import op

    class FakeEngine:
        def __init__(self):
            self.calls = []

        def smooth(self, T, R, Q, fa, fP, time_varying=()):
            self.calls.append("rts")
            return torch.zeros(fa.shape), torch.zeros(fP.shape)

        def smooth_from_filter(self, y, T, Z, H, pa, pP, d=None, filter_type="standard"):
            self.calls.append("from_filter")
            B, n1, m = pa.shape
            return torch.zeros(B, n1 - 1, m), torch.zeros(B, n1 - 1, m, m)

    n, m, r = 7, 10, 4
    rng = np.random.default_rng(0)
    data = rng.normal(size=(n, 1, 1))
    T, Z, R, H, Q = rng.normal(size=(m, m)), rng.normal(size=(1, m)), rng.normal(size=(m, r)), np.eye(1), np.eye(r)
    fs, fP, ps, pP = np.zeros((n, m, 1)), np.zeros((n, m, m)), np.zeros((n + 1, m, 1)), np.zeros((n + 1, m, m))
    c0, d0 = np.zeros((m, 1)), np.zeros((1, 1))
    for c, d, want in ((c0, d0, "from_filter"), (c0 + 0.5, d0, "rts"), (c0, np.zeros((n, 1, 1)), "rts"), (c0, d0 + 2.0, "from_filter")):
        eng = FakeEngine()
        out = op.perform_smoother_fused(data, T, Z, R, H, Q, c, d, fs, fP, ps, pP, engine=eng)
        assert eng.calls == [want], (eng.calls, want)
        assert out[0].shape == (n, m, 1) and out[1].shape == (n, m, m)
    # the batched image: one leading draw axis on everything but the shared data
    B = 3
    lead = lambda x: np.broadcast_to(x, (B,) + x.shape).copy()  # noqa: E731
    for c, want in ((lead(c0), "from_filter"), (lead(c0) + np.array([0.0, 0.0, 1e-3])[:, None, None], "rts")):
        eng = FakeEngine()
        out = op.perform_smoother_fused(data, lead(T), lead(Z), lead(R), lead(H), lead(Q), c, lead(d0), lead(fs), lead(fP), lead(ps),
                                        lead(pP), engine=eng, batched=True)
        assert eng.calls == [want] and out[0].shape == (B, n, m, 1)


def test_vectorize_graph_over_time_varying_nodes_builds_batched_nodes():
    """Round 6 (missing #5): a per-draw node with a time-varying (3-D) matrix vectorises -- the batched nodes take the
    draw axis in FRONT of the time axis ((B, n, rows, cols)); only a0 / P0 must be static.  Node classes and output ranks (CPU; the numeric half:
    tests/test_pytensor_op.py::test_vectorize_graph_over_time_varying_nodes)."""
    import pytensor.tensor as pt
    from pytensor.graph.replace import vectorize_graph

    from pymc_statespace_amd.filters import KalmanSmoother, StandardFilter

    sym = _symbols()
    Tt = pt.tensor3("Tt")
    ins = [sym[0], sym[1], sym[2], Tt] + sym[4:]
    outs = StandardFilter().build_graph(*ins)
    sm = KalmanSmoother().build_graph(Tt, sym[5], sym[7], outs[0], outs[2])
    logp = StandardFilter().build_graph(*ins, outputs="logp")
    bT = pt.tensor4("T_draws")
    vec = vectorize_graph([outs[4], outs[2], sm[1], logp], replace={Tt: bT})
    assert [v.type.ndim for v in vec] == [1, 4, 4, 1]
    f = pytensor.function([sym[0], sym[1], sym[2], bT] + sym[4:], vec, on_unused_input="ignore")
    names = _op_names(f)
    assert "KalmanFilterBatchedOp" in names and "KalmanLogpBatchedOp" in names and "KalmanSmootherFusedOp" in names
    assert not any(nm in names for nm in ("KalmanFilterOp", "KalmanLogpOp", "KalmanSmootherOp"))
    # a batched a0 WITH a time-varying T is fine; a time-varying a0 is not a thing
    ba0 = pt.tensor3("a0_draws")
    v2 = vectorize_graph(logp, replace={sym[1]: ba0})
    assert v2.type.ndim == 1
