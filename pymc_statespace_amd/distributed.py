"""Multi-GPU: shard the batch-of-draws axis, all-gather the per-draw log-likelihoods.

Draws are independent (time is sequential only WITHIN a draw), so the path shards with no
halo and no reduction: rank k filters draws [lo_k, hi_k) on its own GPU and the only exchange
is one all-gather of B/world doubles per rank -- over RCCL on xGMI when
the process group's backend is "nccl" (= RCCL on ROCm), over gloo in the CPU tests.  Output
tapes stay sharded (at the large configs they do not fit one GPU).

One process per GPU; `torch.distributed` must already be initialised by the launcher
(torchrun / `python -m torch.distributed.run`).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of range(B): the first B % world ranks get one extra draw."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_gather_draws(local: torch.Tensor, B: int, group=None) -> torch.Tensor:
    """Gather per-draw values (leading axis = this rank's shard) from every rank into the full
    (B, ...) tensor, in draw order, on every rank.  Uneven shards are padded to the largest.
    A gloo group given device tensors stages them through the host (gloo moves host memory)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(B, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} draws, expected {hi - lo}")
    if local.is_cuda and dist.get_backend(group) == "gloo":
        return all_gather_draws(local.cpu(), B, group).to(local.device)
    width = -(-B // world)
    tail = local.shape[1:]
    if B % world == 0:
        out = torch.empty((B,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((width,) + tail, dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local
    buf = torch.empty((world * width,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    parts = []
    for k in range(world):
        a, b = shard_bounds(B, k, world)
        parts.append(buf[k * width: k * width + (b - a)])
    return torch.cat(parts, dim=0)


def sharded_logp(compute_local: Callable[..., torch.Tensor], batched_inputs: Sequence[torch.Tensor],
                 B: int, group=None, shared_inputs: Sequence = ()) -> torch.Tensor:
    """Evaluate ``compute_local(*shared_inputs, *[x[lo:hi] for x in batched_inputs])`` on this
    rank's shard and all-gather the resulting per-draw vector.  Returns (B,) on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(B, rank, world)
    local = compute_local(*shared_inputs, *[x[lo:hi] for x in batched_inputs])
    if world == 1:
        return local
    return all_gather_draws(local, B, group)


class ShardedKalman:
    """The engine behind a process group: same calls, draws split over the ranks.

    ``logp`` takes the GLOBAL (B, ...) arrays and shards them with ``shard_bounds``;
    ``forward_local`` / ``gather_logp`` / ``forward`` take this rank's shard (what a launcher that
    generates or receives only its own draws holds) -- tapes stay on the rank that filtered them.
    Without an initialised process group everything degenerates to the single-GPU call."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def forward_local(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, out=None, **fkw):
        """kf_filter_forward on this rank's draws.  An EMPTY shard (more ranks than draws) skips
        the launch -- the C-ABI rejects B = 0 -- and returns empty tensors, so that the rank still
        joins the collective that follows."""
        if T.shape[0] == 0:
            from .engine import FilterResult

            dev = self.engine.device
            return FilterResult(logp=torch.empty(0, device=dev, dtype=torch.float64), ll_obs=None,
                                filtered_states=None, predicted_states=None, filtered_covs=None,
                                predicted_covs=None, info=torch.empty(0, device=dev, dtype=torch.int32))
        return self.engine.forward(y, a0, P0, T, Z, R, H, Q, c, d, out=out, **fkw)

    def gather_logp(self, local_logp: torch.Tensor, B_global: int) -> torch.Tensor:
        """All-gather of the per-draw log-likelihoods (the path's only exchange step)."""
        if self._world() == 1:
            return local_logp
        return all_gather_draws(local_logp, B_global, self.group)

    def forward(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, B_global: int, out=None, **fkw):
        """(gathered (B_global,) logp on every rank, this rank's FilterResult)."""
        res = self.forward_local(y, a0, P0, T, Z, R, H, Q, c, d, out=out, **fkw)
        return self.gather_logp(res.logp, B_global), res

    def logp(self, y, a0, P0, T, Z, R, H, Q, *, filter_type="standard", mvn_const="reference",
             return_local: bool = False):
        """Full (B,) log-likelihood vector on every rank (+ this rank's FilterResult)."""
        B = T.shape[0]
        holder = {}

        def local(y_, *mats):
            res = self.forward_local(y_, *mats, filter_type=filter_type, mvn_const=mvn_const,
                                     store_ll_obs=False, store_filtered=False, store_predicted=False)
            holder["res"] = res
            return res.logp

        full = sharded_logp(local, [a0, P0, T, Z, R, H, Q], B, self.group, shared_inputs=(y,))
        return (full, holder["res"]) if return_local else full
