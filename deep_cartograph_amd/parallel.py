"""Frame-sharded data parallelism: one process per GPU, every rank owns a
contiguous block of frames, the small result buffers of the HIP kernels are all-reduced with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).
No data-path tensor ever crosses ranks except the `lag`-row halo of the covariance pass."""
from __future__ import annotations

from typing import Optional, Tuple

import torch


class Comm:
    """Thin view of the default process group; degenerates to a no-op in a single process."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist if dist.is_available() and dist.is_initialized() else None
        self.group = group
        self.rank = self._dist.get_rank(group) if self._dist else 0
        self.world = self._dist.get_world_size(group) if self._dist else 1

    @property
    def active(self) -> bool:
        return self.world > 1

    def _ar(self, t: torch.Tensor, op) -> torch.Tensor:
        if self.active:
            self._dist.all_reduce(t, op=op, group=self.group)
        return t

    def sum_(self, t: torch.Tensor) -> torch.Tensor:
        return self._ar(t, self._dist.ReduceOp.SUM) if self.active else t

    def min_(self, t: torch.Tensor) -> torch.Tensor:
        return self._ar(t, self._dist.ReduceOp.MIN) if self.active else t

    def max_(self, t: torch.Tensor) -> torch.Tensor:
        return self._ar(t, self._dist.ReduceOp.MAX) if self.active else t

    def barrier(self):
        if self.active:
            self._dist.barrier(group=self.group)

    def all_gather(self, t: torch.Tensor):
        if not self.active:
            return [t]
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t.contiguous(), group=self.group)
        return out

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """Concatenation in rank order of per-rank tensors whose first dimension differs (the
        frame shards): sizes are exchanged first, shards padded to the largest for the collective."""
        if not self.active:
            return t
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        sizes = [int(x.item()) for x in self.all_gather(n)]
        pad = torch.zeros((max(sizes),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        parts = self.all_gather(pad)
        return torch.cat([p[:m] for p, m in zip(parts, sizes)], dim=0)

    def sum_scalar(self, v: float, device="cpu") -> float:
        t = torch.tensor([float(v)], dtype=torch.float64, device=device)
        return float(self.sum_(t).item())


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of `n` frames owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_col_stats(raw: torch.Tensor, comm: Comm) -> torch.Tensor:
    """raw = [sum | sumsq | min | max] (4 x F float64) of a shard -> global (in place)."""
    if comm.active:
        comm.sum_(raw[0])
        comm.sum_(raw[1])
        comm.min_(raw[2])
        comm.max_(raw[3])
    return raw


def reduce_minmax(mm: torch.Tensor, comm: Comm) -> torch.Tensor:
    """mm = [min | max] (2 x d) of a shard -> global (in place)."""
    if comm.active:
        comm.min_(mm[0])
        comm.max_(mm[1])
    return mm


def append_halo(X_local: torch.Tensor, lag: int, comm: Comm) -> Tuple[torch.Tensor, int]:
    """Time-lagged pairs (i, i+lag) of a trajectory cut into contiguous shards: every rank but
    the last borrows the first `lag` rows of its successor, so the union of the per-shard pair
    sets is exactly the single-process pair set.  Returns (rows incl. halo, n_pairs_local)."""
    n_local = X_local.shape[0]
    if not comm.active or lag == 0:
        return X_local, n_local - lag
    if n_local < lag:
        raise ValueError(f"shard of {n_local} frames is shorter than the lag {lag}")
    heads = comm.all_gather(X_local[:lag].contiguous())
    if comm.rank == comm.world - 1:
        return X_local, n_local - lag
    return torch.cat([X_local, heads[comm.rank + 1]], dim=0), n_local


def reduce_nearest(dist: torch.Tensor, rows: torch.Tensor, comm: Comm):
    """Per-centroid (distance, global row) candidates of each shard -> the global winner:
    smallest distance, lowest row on ties (np.argmin semantics)."""
    if not comm.active:
        return dist, rows
    d_all = torch.stack(comm.all_gather(dist))   # [world, k]
    r_all = torch.stack(comm.all_gather(rows))
    best_d = d_all.min(dim=0).values
    cand = torch.where(d_all == best_d[None, :], r_all, torch.full_like(r_all, torch.iinfo(torch.int64).max))
    return best_d, cand.min(dim=0).values


def global_topk(values: torch.Tensor, payload: torch.Tensor, m: int, comm: Comm) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The m largest `values` over every rank's shard, largest first (ties: lower rank, then lower
    local index -- the order a single process sees in the concatenated array with a stable sort).
    Returns (values [m], owner rank [m], payload rows [m, ...]); every rank gets the same result."""
    m_loc = min(m, values.shape[0])
    order = torch.argsort(values, descending=True, stable=True)[:m_loc]
    v_loc, p_loc = values[order], payload[order]
    if not comm.active:
        return v_loc, torch.zeros(m_loc, dtype=torch.int64, device=values.device), p_loc
    r_loc = torch.full((m_loc,), comm.rank, dtype=torch.int64, device=values.device)
    v_all, r_all, p_all = comm.all_gather_rows(v_loc), comm.all_gather_rows(r_loc), comm.all_gather_rows(p_loc)
    pick = torch.argsort(v_all, descending=True, stable=True)[:m]   # rank-major concatenation: stable = rank, then index
    return v_all[pick], r_all[pick], p_all[pick]
