"""N > 1 path with the REAL engine: ShardedKalman behind an RCCL ("nccl") process group, the
bench launcher as the driver invokes it, and the device guard of the C-ABI.

A 1-GPU box can only form a 1-rank RCCL group; the 2-rank control flow (weak + strong probe) is run
there with both ranks sharing cuda:0 over gloo (bench.py's KF_BENCH_SHARE_GPU test mode), the
2-rank RCCL variant skips unless two GPUs are visible."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import workloads as W
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(cmd, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-3000:]
    return json.loads(lines[0])


_SHARDED_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import workloads as W
from oracle import c_oracle as co
from pymc_statespace_amd import KalmanEngine
from pymc_statespace_amd.distributed import ShardedKalman, shard_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
assert dist.get_world_size() == world and dist.get_backend() == "nccl"
B = {B}
w = W.structural(B=B, n=120)
eng = KalmanEngine()
sk = ShardedKalman(eng)
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
full, local = sk.logp(*dev, mvn_const="exact", return_local=True)
lo, hi = shard_bounds(B, rank, world)
assert local.logp.shape[0] == hi - lo and full.shape == (B,)
ref = co.kalman_filter(w["y"], *[w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q")],
                       mvn_const="exact", full=False)["logp"]
err = float(np.abs(full.cpu().numpy() - ref).max())
assert err < 1e-8, err
shard = [x[lo:hi] for x in dev[1:]]
g, res = sk.forward(dev[0], *shard, B_global=B, mvn_const="exact")
assert torch.equal(g, full) and res.filtered_covs.shape[0] == hi - lo
# round 5: logp + gradient, gathered with ONE all-gather of the packed rows; smoothed tapes stay on the rank
lp, info, gr = sk.value_and_grad(dev[0], *shard, B_global=B, mvn_const="exact")
lp1, info1, gr1 = eng.value_and_grad(*dev, mvn_const="exact")
assert lp.shape == (B,) and not info.any() and float((lp - lp1).abs().max()) < 1e-9
for k in sk.GRAD_KEYS:
    assert gr[k].shape == gr1[k].shape, k
    scale = gr1[k].abs().reshape(B, -1).max(dim=1).values.clamp_min(1e-300)
    assert float(((gr[k] - gr1[k]).abs().reshape(B, -1).max(dim=1).values / scale).max()) < 1e-9, k
sa, sP = sk.smooth_local(shard[2], shard[4], shard[6], res.filtered_states, res.filtered_covs)
fa, fP = sk.smooth_from_filter_local(dev[0], shard[2], shard[3], shard[5], res.predicted_states, res.predicted_covs)
assert sa.shape[0] == hi - lo and float((sa - fa).abs().max()) < 1e-9 and float((sP - fP).abs().max()) < 1e-9
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("SHARDED_OK", err)
"""


def _torchrun(nproc, script_text, tmp_path, B):
    script = tmp_path / "worker.py"
    script.write_text(script_text.format(root=ROOT, B=B))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0 and "SHARDED_OK" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]


def test_sharded_kalman_one_rank_rccl_group_real_engine(tmp_path):
    """ShardedKalman over backend "nccl" (= RCCL) with libkf_hip.so doing the filtering."""
    _torchrun(1, _SHARDED_WORKER, tmp_path, B=37)


def test_sharded_kalman_two_rank_rccl_group_real_engine(tmp_path):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _torchrun(2, _SHARDED_WORKER, tmp_path, B=37)


def test_bench_under_a_launcher_one_rank():
    """The driver's N > 1 invocation shape (python -m torch.distributed.run ... bench.py --gpus N) at N = 1."""
    j = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
              "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
              os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
              "--batch", "512", "--cpu-sample", "32"])
    assert j["n_gpus"] == 1 and j["scaling"] == "weak" and j["config"]["global_draws"] == 512
    assert j["parity"]["max_abs_dlogp_vs_oracle"] < 1e-8
    assert j["roofline"]["bound"] == "hbm" and j["cpu_baseline"]["kind"] == "port"


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_sharing_the_gpu(scaling):
    """`python bench.py --gpus 2` with no launcher: the script starts its own two ranks.  On a 1-GPU
    box both ranks share cuda:0 and exchange over gloo (test mode) -- this pins the control flow of
    the N > 1 path (sharding, gathered-logp parity, strong-scaling probe), not its speed."""
    j = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
              "--batch", "600", "--cpu-sample", "32", "--scaling", scaling],
             env_extra={"KF_BENCH_SHARE_GPU": "1"})
    assert j["n_gpus"] == 2 and j["scaling"] == scaling
    if scaling == "weak":
        assert j["config"]["global_draws"] == 1200 and j["config"]["draws_per_gpu"] == 600
        s = j["strong_scaling"]
        assert s["global_draws"] == 600 and s["draws_per_gpu"] == 300
        assert s["max_abs_dlogp_sharded_vs_single_gpu"] == 0.0   # same kernel, same draws: same bits
    else:
        assert j["config"]["global_draws"] == 600 and j["config"]["draws_per_gpu"] == 300
        assert "strong_scaling" not in j
    assert j["parity"]["max_abs_dlogp_vs_oracle"] < 1e-8
    # round 5: the sharded logp + gradient (one all-gather of the packed rows) and the per-rank smoothers are timed beside the forward
    ops = j["other_entry_points_sharded"]
    assert "error" not in ops, ops
    for k in ("logp_and_gradient_gathered_ms", "logp_and_gradient_local_grads_ms", "rts_smoother_local_ms",
              "smoother_from_filter_local_ms"):
        assert ops[k] > 0.0, k


def test_engine_on_another_device_than_the_current_one():
    """a handle created for cuda:1 must launch on cuda:1 while cuda:0 is current."""
    import torch

    from pymc_statespace_amd import KalmanEngine

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    torch.cuda.set_device(0)
    eng1 = KalmanEngine(torch.device("cuda", 1))
    w = W.structural(B=33, n=80)
    res = eng1.forward(w["y"], *[w[k] for k in KEYS], mvn_const="exact")
    assert res.logp.device.index == 1 and torch.cuda.current_device() == 0
    ref = co.kalman_filter(w["y"], *[w[k] for k in KEYS], mvn_const="exact", full=False)["logp"]
    assert np.abs(res.logp.cpu().numpy() - ref).max() < 1e-8
    w2 = W.daily_seasonal(B=2, n=6, s=70)        # workspace path + hipFuncSetAttribute on device 1
    r2 = eng1.forward(w2["y"], *[w2[k] for k in KEYS], mvn_const="exact", store_filtered=False,
                      store_predicted=False)
    ref2 = co.kalman_filter(w2["y"], *[w2[k] for k in KEYS], mvn_const="exact", full=False)["logp"]
    assert np.abs(r2.logp.cpu().numpy() - ref2).max() < 1e-8


def test_engine_same_device_guard_is_transparent():
    """On one GPU the guard must be a no-op: current device unchanged, results as before."""
    import torch

    from pymc_statespace_amd import KalmanEngine

    eng = KalmanEngine()
    w = W.arma11(B=70, n=64)
    res = eng.forward(w["y"], *[w[k] for k in KEYS], mvn_const="exact")
    ref = co.kalman_filter(w["y"], *[w[k] for k in KEYS], mvn_const="exact", full=False)["logp"]
    assert np.abs(res.logp.cpu().numpy() - ref).max() < 1e-8 * max(1.0, np.abs(ref).max() / 1e4)
    assert torch.cuda.current_device() == eng.device.index
