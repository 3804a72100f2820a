"""N > 1 path on CPU: world_size-2 gloo process group, draw sharding + logp all-gather.

The per-rank compute is injected (here: the numpy oracle) -- these tests pin the sharding and
the collective, not the kernels (no GPU in this container)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import workloads as W
from pymc_statespace_amd.distributed import all_gather_draws, shard_bounds, sharded_logp


def test_shard_bounds_cover_and_balance():
    for B in (1, 2, 7, 8, 4096, 4097, 16384):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import kalman_numpy as kn

        w = W.structural(B=B, n=20)
        calls = []

        def local(y, a0, P0, T, Z, R, H, Q):
            calls.append(a0.shape[0])
            out = kn.kalman_filter(y.numpy(), *[x.numpy() for x in (a0, P0, T, Z, R, H, Q)])
            return torch.from_numpy(out["logp"])

        mats = [torch.from_numpy(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
        full = sharded_logp(local, mats, B, shared_inputs=(torch.from_numpy(w["y"]),))
        lo, hi = shard_bounds(B, rank, world)
        assert calls == [hi - lo]
        # also gather a 2-D per-draw payload (e.g. gradients w.r.t. a0)
        g = all_gather_draws(mats[0][lo:hi] * 2.0, B)
        assert torch.equal(g, mats[0] * 2.0)
        ret[rank] = full.numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 11])
def test_two_rank_gloo_allgather_matches_single_process(B):
    from oracle import kalman_numpy as kn

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, ret), nprocs=world, join=True)
    w = W.structural(B=B, n=20)
    ref = kn.kalman_filter(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])["logp"]
    for r in range(world):
        assert np.array_equal(ret[r], ref), f"rank {r}"


class _OracleEngine:
    """Stands in for KalmanEngine on CPU: rejects B == 0 exactly as kf_filter_forward does."""

    device = torch.device("cpu")

    def forward(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, out=None, **kw):
        from oracle import kalman_numpy as kn
        from pymc_statespace_amd.engine import FilterResult

        if T.shape[0] < 1:
            raise RuntimeError("kf_filter_forward failed (-1): bad dims: B=0")
        o = kn.kalman_filter(y.numpy(), *[x.numpy() for x in (a0, P0, T, Z, R, H, Q)],
                             mvn_const=kw.get("mvn_const", "reference"))
        return FilterResult(logp=torch.from_numpy(o["logp"]), ll_obs=None, filtered_states=None,
                            predicted_states=None, filtered_covs=None, predicted_covs=None,
                            info=torch.zeros(T.shape[0], dtype=torch.int32))


def _worker_sharded_kalman(rank, world, port, B, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pymc_statespace_amd.distributed import ShardedKalman

        w = W.structural(B=B, n=15)
        sk = ShardedKalman(_OracleEngine())
        y = torch.from_numpy(w["y"])
        mats = [torch.from_numpy(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
        full, local = sk.logp(y, *mats, return_local=True)       # global arrays in, sharded inside
        lo, hi = shard_bounds(B, rank, world)
        assert local.logp.shape[0] == hi - lo
        g2, res = sk.forward(y, *[x[lo:hi] for x in mats], B_global=B)   # this rank's shard in
        assert torch.equal(g2, full)
        ret[rank] = full.numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [1, 5])
def test_sharded_kalman_gloo_with_an_empty_shard(B):
    """B < world leaves a rank without draws: it must skip the launch (the C-ABI rejects B = 0)
    and still join the all-gather -- before, the other ranks hung in the collective."""
    from oracle import kalman_numpy as kn

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_sharded_kalman, args=(world, _free_port(), B, ret), nprocs=world, join=True)
    w = W.structural(B=B, n=15)
    ref = kn.kalman_filter(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])["logp"]
    for r in range(world):
        assert np.array_equal(ret[r], ref), f"rank {r}"


class _OracleGradEngine(_OracleEngine):
    """value_and_grad / smooth from the numpy restatements (oracle/kalman_adjoint_numpy.py, kalman_numpy.kalman_smoother)."""

    def forward(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, out=None, **kw):
        from oracle import kalman_numpy as kn
        from pymc_statespace_amd.engine import FilterResult

        if T.shape[0] < 1:
            raise RuntimeError("kf_filter_forward failed (-1): bad dims: B=0")
        o = kn.kalman_filter(y.numpy(), *[x.numpy() for x in (a0, P0, T, Z, R, H, Q)])
        t = torch.from_numpy
        B, m = a0.shape
        return FilterResult(logp=t(o["logp"]), ll_obs=None, filtered_states=t(o["filtered_states"]).reshape(B, -1, m),
                            predicted_states=t(o["predicted_states"]).reshape(B, -1, m), filtered_covs=t(o["filtered_covs"]),
                            predicted_covs=t(o["predicted_covs"]), info=torch.zeros(B, dtype=torch.int32))

    def value_and_grad(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, g_logp=None, **kw):
        from oracle import kalman_adjoint_numpy as ka
        from oracle import kalman_numpy as kn

        if T.shape[0] < 1:
            raise RuntimeError("kf_filter_value_and_grad failed (-1): bad dims: B=0")
        mats = [x.numpy() for x in (a0, P0, T, Z, R, H, Q)]
        lp = kn.kalman_filter(y.numpy(), *mats)["logp"]
        g = ka.logp_grads(y.numpy(), *mats, g_logp=None if g_logp is None else g_logp.numpy())
        return (torch.from_numpy(lp), torch.zeros(T.shape[0], dtype=torch.int32),
                {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items()})

    def backward(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, predicted_states, predicted_covs, g_logp=None, g_ll_obs=None,
                 g_filtered_states=None, g_predicted_states=None, g_filtered_covs=None, g_predicted_covs=None, want_cd=False, **kw):
        """torch.autograd through the torch restatement (oracle/kalman_torch.py): the checker of kf_filter_backward_tapes."""
        from oracle import kalman_torch as kt

        if T.shape[0] < 1:
            raise RuntimeError("kf_filter_backward_tapes failed (-1): bad dims: B=0")
        gt = {k: v.numpy() for k, v in (("filtered_states", g_filtered_states), ("predicted_states", g_predicted_states),
                                        ("filtered_covs", g_filtered_covs), ("predicted_covs", g_predicted_covs)) if v is not None}
        B = T.shape[0]
        _, _, g = kt.logp_and_grads(y.numpy(), *[x.numpy() for x in (a0, P0, T, Z, R, H, Q)],
                                    g_logp=np.zeros(B) if g_logp is None else g_logp.numpy(),
                                    g_ll_obs=None if g_ll_obs is None else g_ll_obs.numpy(), g_tapes=gt or None)
        return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items()}

    def smooth(self, T, R, Q, fa, fP, **kw):
        from oracle import kalman_numpy as kn

        if T.shape[0] < 1:
            raise RuntimeError("kf_smoother failed (-1): bad dims: B=0")
        sa, sP = kn.kalman_smoother(T.numpy(), R.numpy(), Q.numpy(), fa.numpy(), fP.numpy())
        return torch.from_numpy(sa).reshape(fa.shape), torch.from_numpy(sP)

    # the REAL dispatch (KalmanEngine.smooth_result: which recursion a caller that holds the filter's result gets) over oracle bodies
    from pymc_statespace_amd.engine import KalmanEngine as _KE

    smooth_result = _KE.smooth_result
    smooth_from_filter_ok = _KE.smooth_from_filter_ok

    def to_device(self, x, shape=None):
        return None if x is None else torch.as_tensor(x)

    def smooth_from_filter(self, data, T, Z, H, pa, pP, d=None, filter_type="standard"):
        from oracle import kalman_numpy as kn

        if T.shape[0] < 1:
            raise RuntimeError("kf_smoother_from_filter failed (-1): bad dims: B=0")
        sa, sP = kn.smoother_from_filter(data.numpy(), T.numpy(), Z.numpy(), H.numpy(), pa.numpy(), pP.numpy(),
                                         d=None if d is None else d.numpy(), filter_type=filter_type)
        return torch.from_numpy(sa), torch.from_numpy(sP)


def _worker_sharded_grad_and_smooth(rank, world, port, B, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pymc_statespace_amd.distributed import ShardedKalman

        w = W.nile_local_linear_trend(B=B)
        sk = ShardedKalman(_OracleGradEngine())
        y = torch.from_numpy(w["y"])
        mats = [torch.from_numpy(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
        lo, hi = shard_bounds(B, rank, world)
        shard = [x[lo:hi] for x in mats]
        gl = torch.linspace(0.5, 1.5, B, dtype=torch.float64)
        lp, info, g = sk.value_and_grad(y, *shard, B_global=B, g_logp=gl[lo:hi])
        assert lp.shape == (B,) and info.shape == (B,) and info.dtype == torch.int32
        assert all(g[k].shape[0] == B for k in sk.GRAD_KEYS)
        lp2, info2, g_loc = sk.value_and_grad(y, *shard, B_global=B, g_logp=gl[lo:hi], gather_grads=False)
        assert torch.equal(lp2, lp) and all(torch.equal(g_loc[k], g[k][lo:hi]) for k in sk.GRAD_KEYS)
        # the reference's recursion on the filtered tapes ...
        full, res, sa, sP = sk.filter_and_smooth(y, *shard, B_global=B, route="rts")
        assert sa.shape[0] == hi - lo and sP.shape[0] == hi - lo and torch.equal(full, lp)
        if hi > lo:
            sa2, sP2 = sk.smooth_local(shard[2], shard[4], shard[6], res.filtered_states, res.filtered_covs)
            assert torch.equal(sa2, sa) and torch.equal(sP2, sP)
        # ... and what a caller gets by default (round 6): the filter's result is at hand, p == 1, static, no c -> the predicted tapes
        full_b, res_b, sa_b, sP_b = sk.filter_and_smooth(y, *shard, B_global=B)
        assert torch.equal(full_b, lp) and sa_b.shape == sa.shape and sP_b.shape == sP.shape
        if hi > lo:
            sa3, sP3 = sk.smooth_from_filter_local(y, shard[2], shard[3], shard[5], res_b.predicted_states, res_b.predicted_covs)
            assert torch.equal(sa3, sa_b) and torch.equal(sP3, sP_b)
            # a non-zero state intercept keeps the reference's recursion (it ignores c: kalman_smoother.py:99-104)
            c1 = torch.full((hi - lo, shard[0].shape[1]), 0.25, dtype=torch.float64)
            _, res_c, sa_c, sP_c = sk.filter_and_smooth(y, *shard, c1, B_global=B)
            sa_r, sP_r = sk.smooth_local(shard[2], shard[4], shard[6], res_c.filtered_states, res_c.filtered_covs)
            assert torch.equal(sa_c, sa_r) and torch.equal(sP_c, sP_r)
        else:
            sk.filter_and_smooth(y, *shard, torch.zeros(0, shard[0].shape[1], dtype=torch.float64), B_global=B)  # (joins the gather)
        with pytest.raises(NotImplementedError):   # every rank, empty or not
            sk.value_and_grad(y, *shard, B_global=B, time_varying=("T",))
        # (round 6) gradients through the tapes: cotangents sharded like the tapes, nothing exchanged; an empty shard returns
        # empty arrays that pack into the same all-gather as the log-likelihood gradients
        gen = torch.Generator().manual_seed(5)
        n, m = y.shape[0], shard[0].shape[1]
        g_fa = torch.randn(B, n, m, dtype=torch.float64, generator=gen)
        g_pP = torch.randn(B, n + 1, m, m, dtype=torch.float64, generator=gen)
        g_pP = 0.5 * (g_pP + g_pP.transpose(-1, -2))
        gtl = sk.backward_local(y, *shard, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                                g_filtered_states=g_fa[lo:hi], g_predicted_covs=g_pP[lo:hi])
        assert all(gtl[k].shape[0] == hi - lo for k in sk.GRAD_KEYS)
        _, _, gt_full = sk.gather_value_and_grad(lp[lo:hi], info[lo:hi], gtl, B)
        with pytest.raises(NotImplementedError):
            sk.backward_local(y, *shard, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, time_varying=("T",))
        ret[rank] = dict(logp=lp.numpy(), grads={k: v.numpy() for k, v in g.items()}, lo=lo, hi=hi, sa=sa.numpy(), sP=sP.numpy(),
                         sa_b=sa_b.numpy(), sP_b=sP_b.numpy(), tape_grads={k: v.numpy() for k, v in gt_full.items()},
                         g_fa=g_fa.numpy(), g_pP=g_pP.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [1, 5, 6])
def test_sharded_value_and_grad_and_smoother_gloo(B):
    """gradients are gathered like the log-likelihoods (ONE all-gather of the packed rows), smoothed tapes stay
    sharded; even, uneven and empty shards."""
    from oracle import kalman_adjoint_numpy as ka
    from oracle import kalman_numpy as kn

    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_sharded_grad_and_smooth, args=(world, _free_port(), B, ret), nprocs=world, join=True)
    w = W.nile_local_linear_trend(B=B)
    mats = [w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    flt = kn.kalman_filter(w["y"], *mats)
    gl = np.linspace(0.5, 1.5, B)
    g = ka.logp_grads(w["y"], *mats, g_logp=gl)
    sa, sP = kn.kalman_smoother(w["T"], w["R"], w["Q"], flt["filtered_states"], flt["filtered_covs"])
    sa = sa.reshape(B, -1, w["m"])
    sb, sPb = kn.smoother_from_filter(w["y"], w["T"], w["Z"], w["H"], flt["predicted_states"], flt["predicted_covs"])
    for r in range(world):
        assert np.array_equal(ret[r]["logp"], flt["logp"]), f"rank {r}"
        for k in g:
            assert np.array_equal(ret[r]["grads"][k], g[k]), (r, k)
        lo, hi = ret[r]["lo"], ret[r]["hi"]
        assert np.array_equal(ret[r]["sa"], sa[lo:hi]) and np.array_equal(ret[r]["sP"], sP[lo:hi])
        assert np.array_equal(ret[r]["sa_b"], sb[lo:hi]) and np.array_equal(ret[r]["sP_b"], sPb[lo:hi])
    from oracle import kalman_torch as kt

    _, _, gref = kt.logp_and_grads(w["y"], *mats, g_tapes=dict(filtered_states=ret[0]["g_fa"], predicted_covs=ret[0]["g_pP"]))
    for r in range(world):   # every draw's tape gradient, whichever rank computed it
        for k in gref:
            assert np.allclose(ret[r]["tape_grads"][k], gref[k], rtol=1e-12, atol=1e-12), (r, k)


def test_bench_launcher_builds_a_torchrun_command(monkeypatch):
    """`bench.py --gpus N` without a launcher must start N ranks itself, from a parent that has not
    touched the GPU (the driver invokes it exactly like this)."""
    import importlib
    import sys

    import bench

    importlib.reload(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch" not in bench.__dict__          # the parent never imported torch
