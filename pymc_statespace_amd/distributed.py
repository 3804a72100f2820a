"""Multi-GPU: shard the batch-of-draws axis, all-gather the per-draw log-likelihoods (and, when the sampler asks for
them, the per-draw gradients -- SURVEY.md section 8e: "gradients, when requested, are gathered the same way").

Draws are independent (time is sequential only WITHIN a draw), so the path shards with no
halo and no reduction: rank k filters draws [lo_k, hi_k) on its own GPU and the only exchange
is one all-gather of B/world doubles per rank -- over RCCL on xGMI when
the process group's backend is "nccl" (= RCCL on ROCm), over gloo in the CPU tests.  A value-and-gradient
call packs logp, info and the nine gradient arrays of a draw into one row and gathers THAT: still one
collective per evaluation (the message is latency-bound on xGMI: B/world x (2 + n_theta) doubles).  Output
tapes -- filtered, predicted, smoothed -- stay sharded (at the large configs they do not fit one GPU).

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

    # ------------------------------------------------------------------ logp + gradient (what NUTS evaluates)
    GRAD_KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q", "c", "d")

    def value_and_grad_local(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, **kw):
        """kf_filter_value_and_grad on this rank's draws -> (logp, info, dict of gradients).  An empty shard skips the
        launch and returns empty tensors of the right trailing shapes."""
        if kw.get("time_varying"):
            #  kf_filter_value_and_grad takes static models only: a time-varying gradient has (b, n, rows, cols)
            # arrays, whose packed row width would differ between an empty and a non-empty shard -- reject it on EVERY rank
            raise NotImplementedError("the sharded value-and-gradient takes static system matrices (use engine.forward + "
                                      "engine.backward with time_varying=... on each rank's shard)")
        if T.shape[0] == 0:
            dev, f64 = self.engine.device, torch.float64
            # trailing shapes from the inputs themselves (an empty shard still carries them): what the non-empty ranks pack
            shp = dict(a0=tuple(a0.shape[1:]), P0=tuple(P0.shape[1:]), T=tuple(T.shape[1:]), Z=tuple(Z.shape[1:]),
                       R=tuple(R.shape[1:]), H=tuple(H.shape[1:]), Q=tuple(Q.shape[1:]), c=(T.shape[-1],), d=(Z.shape[-2],))
            return (torch.empty(0, device=dev, dtype=f64), torch.empty(0, device=dev, dtype=torch.int32),
                    {k: torch.empty((0,) + shp[k], device=dev, dtype=f64) for k in self.GRAD_KEYS})
        return self.engine.value_and_grad(y, a0, P0, T, Z, R, H, Q, c, d, **kw)

    def gather_value_and_grad(self, logp, info, grads, B_global: int):
        """ONE all-gather for logp, info and the nine gradient arrays: each draw's values packed into a row
        [logp | info | g_a0 | g_P0 | ... | g_d] (info travels as a double: exact for int32).  Returns the full
        (B_global, ...) arrays on every rank."""
        if self._world() == 1:
            return logp, info, grads
        b = logp.shape[0]
        import math

        cols = [logp.reshape(b, 1), info.to(logp.dtype).reshape(b, 1)]
        cols += [grads[k].reshape(b, math.prod(grads[k].shape[1:])) for k in self.GRAD_KEYS]  # (an empty shard: 0 x width)
        widths = [c.shape[1] for c in cols]
        full = all_gather_draws(torch.cat(cols, dim=1), B_global, self.group)
        parts = torch.split(full, widths, dim=1)
        out = {k: parts[2 + i].reshape((B_global,) + tuple(grads[k].shape[1:])) for i, k in enumerate(self.GRAD_KEYS)}
        return parts[0].reshape(B_global).contiguous(), parts[1].reshape(B_global).to(torch.int32), out

    def value_and_grad(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, B_global: int, gather_grads: bool = True, **kw):
        """This rank's shard in; (logp (B_global,), info (B_global,), gradients) out on every rank -- gathered like the
        log-likelihoods (``gather_grads=False``: only logp and info are gathered, the gradients stay this rank's shard,
        for a caller that reduces them over theta locally first).  g_logp, when given, is this rank's shard of it."""
        logp, info, g = self.value_and_grad_local(y, a0, P0, T, Z, R, H, Q, c, d, **kw)
        if self._world() == 1:
            return logp, info, g
        if gather_grads:
            return self.gather_value_and_grad(logp, info, g, B_global)
        b = logp.shape[0]
        full = all_gather_draws(torch.stack([logp, info.to(logp.dtype)], dim=1).reshape(b, 2), B_global, self.group)
        return full[:, 0].contiguous(), full[:, 1].to(torch.int32), g

    def backward_local(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, predicted_states, predicted_covs, **kw):
        """kf_filter_backward / kf_filter_backward_tapes on this rank's draws and ITS tapes: the cotangents of a tape live where the
        tape does (``g_logp``, ``g_ll_obs``, ``g_filtered_states`` ... are this rank's shards), nothing is exchanged.  Returns the
        dict of gradients (a0, P0, T, Z, R, H, Q, c, d) -- ``gather_value_and_grad``'s packing applies to it as it stands.  An empty
        shard skips the launch and returns empty tensors of the right trailing shapes (static models; a time-varying gradient has
        its matrix's (b, n, ..) shape: call the engine on each rank's shard)."""
        if kw.get("time_varying"):
            raise NotImplementedError("the sharded backward takes static system matrices (use engine.backward with "
                                      "time_varying=... on each rank's shard)")
        if T.shape[0] == 0:
            dev, f64 = self.engine.device, torch.float64
            shp = dict(a0=tuple(a0.shape[1:]), P0=tuple(P0.shape[1:]), T=tuple(T.shape[1:]), Z=tuple(Z.shape[1:]),
                       R=tuple(R.shape[1:]), H=tuple(H.shape[1:]), Q=tuple(Q.shape[1:]), c=(T.shape[-1],), d=(Z.shape[-2],))
            return {k: torch.empty((0,) + shp[k], device=dev, dtype=f64) for k in self.GRAD_KEYS}
        kw.setdefault("want_cd", True)
        return self.engine.backward(y, a0, P0, T, Z, R, H, Q, c, d, predicted_states=predicted_states,
                                    predicted_covs=predicted_covs, **kw)

    # ------------------------------------------------------------------ smoother: tapes stay where they were filtered
    def smooth_local(self, T, R, Q, filtered_states, filtered_covs, **kw):
        """kf_smoother on this rank's tapes; nothing is exchanged (a smoothed tape is as large as the filtered one it came
        from -- C5: 0.8 TB over the job).  An empty shard returns empty tapes."""
        if T.shape[0] == 0:
            return torch.empty_like(filtered_states), torch.empty_like(filtered_covs)
        return self.engine.smooth(T, R, Q, filtered_states, filtered_covs, **kw)

    def smooth_from_filter_local(self, data, T, Z, H, predicted_states, predicted_covs, **kw):
        """kf_smoother_from_filter on this rank's predicted tapes (p == 1, static matrices, m <= 1024: include/kf.h); nothing is
        exchanged."""
        if T.shape[0] == 0:
            dev, f64 = self.engine.device, torch.float64
            n, m = predicted_states.shape[1] - 1, T.shape[-1]
            return torch.empty(0, n, m, device=dev, dtype=f64), torch.empty(0, n, m, m, device=dev, dtype=f64)
        return self.engine.smooth_from_filter(data, T, Z, H, predicted_states, predicted_covs, **kw)

    def filter_and_smooth(self, y, a0, P0, T, Z, R, H, Q, c=None, d=None, *, B_global: int, route: str = "auto", **fkw):
        """Forward filter + smoother of this rank's draws: (gathered logp (B_global,), this rank's FilterResult, this rank's
        smoothed states and covariances) -- build_statespace_graph + build_smoother_graph (core/statespace.py:164-197) for a
        shard.  The only exchange is the logp all-gather.  The smoother runs on the filter's PREDICTED tapes wherever that is the
        cheaper recursion (``KalmanEngine.smooth_result``: p == 1, static matrices, zero c -- no pinv(P_hat)), else on the filtered
        ones; ``route="rts"`` forces the reference's recursion."""
        res = self.forward_local(y, a0, P0, T, Z, R, H, Q, c, d, **fkw)
        full = self.gather_logp(res.logp, B_global)
        if T.shape[0] == 0:
            dev, f64 = self.engine.device, torch.float64
            n, m = y.shape[-2], T.shape[-1]
            return full, res, torch.empty(0, n, m, device=dev, dtype=f64), torch.empty(0, n, m, m, device=dev, dtype=f64)
        sa, sP = self.engine.smooth_result(res, y, T, Z, R, H, Q, c, d, filter_type=fkw.get("filter_type", "standard"),
                                           time_varying=tuple(fkw.get("time_varying", ())), route=route)
        return full, res, sa, sP
