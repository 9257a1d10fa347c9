"""Chains sharded over the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in CPU tests).  The reference has no distributed code; chains are
independent between flow refits (mcmc/base.py:74-77), so the only traffic is

  C1  all-gather of each rank's share of the refit buffer (tuning.train_val_split), and
  C2  one all-reduce(SUM) of [sum_x (d), sum_x2 (d), counters] at the end of sample().

Native noise is keyed by the GLOBAL chain id (Shard.bounds -> chain_offset), so the chains a rank
simulates are bit-for-bit the chains a single GPU would have simulated.
"""
import os
from typing import Tuple

import torch
import torch.distributed as tdist


class Shard:
    def __init__(self, rank: int = None, world: int = None, group=None):
        self.group = group
        if rank is None:
            rank = tdist.get_rank(group) if tdist.is_initialized() else 0
        if world is None:
            world = tdist.get_world_size(group) if tdist.is_initialized() else 1
        self.rank, self.world = int(rank), int(world)

    def _single(self) -> bool:
        """One rank: the collectives are identities and skipped (NFMC_SHARD_NO_SHORTCUT=1 runs them anyway, to
        rehearse the RCCL code path on a single GPU)."""
        return self.world == 1 and not (os.environ.get('NFMC_SHARD_NO_SHORTCUT') == '1' and tdist.is_initialized())

    def bounds(self, n_global: int) -> Tuple[int, int]:
        """Contiguous block of chains owned by this rank (block sizes differ by at most one)."""
        base, rem = divmod(n_global, self.world)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)

    def _backend_device(self, t: torch.Tensor):
        if self.world > 1 and tdist.get_backend(self.group) == 'gloo':
            return t.cpu()
        return t

    def broadcast_int(self, value: int) -> int:
        if self._single():
            return int(value)
        dev = 'cpu' if tdist.get_backend(self.group) == 'gloo' else torch.device('cuda', torch.cuda.current_device())
        t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
        tdist.broadcast(t, src=0, group=self.group)
        return int(t.item())

    def _scalar(self, value, dtype):
        dev = 'cpu' if tdist.get_backend(self.group) == 'gloo' else torch.device('cuda', torch.cuda.current_device())
        return torch.tensor([value], dtype=dtype, device=dev)

    def all_reduce_min_int(self, value: int) -> int:
        """The smallest `value` over the ranks (row counts agreed before an all-gather of equal shares)."""
        if self._single():
            return int(value)
        t = self._scalar(int(value), torch.int64)
        tdist.all_reduce(t, op=tdist.ReduceOp.MIN, group=self.group)
        return int(t.item())

    def all_reduce_mean_(self, t: torch.Tensor, weight: float) -> torch.Tensor:
        """In place: the `weight`-weighted mean of `t` over the ranks (per-rank statistics of unequal chain blocks)."""
        if self._single():
            return t
        pack = torch.cat([t.detach().reshape(-1).double() * weight, t.new_tensor([weight]).double()])
        self.all_reduce_sum_(pack)
        t.copy_((pack[:-1] / pack[-1]).reshape(t.shape).to(t.dtype))
        return t

    def all_gather_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """C1: concatenate every rank's (k, ...) block in rank order (equal k on every rank)."""
        if self._single():
            return rows
        src = self._backend_device(rows.contiguous())
        out = torch.empty((self.world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        tdist.all_gather_into_tensor(out, src, group=self.group)
        return out.to(rows.device)

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        if self._single():
            return t
        buf = self._backend_device(t)
        tdist.all_reduce(buf, op=tdist.ReduceOp.SUM, group=self.group)
        if buf is not t:
            t.copy_(buf)
        return t

    def merge_statistics(self, st):
        """C2: make `st` (MCMCStatistics / JumpNFMCStatistics) the statistics of ALL chains on every rank."""
        if self._single():
            return st
        e1, e2 = st.expectations['first_moment'], st.expectations['second_moment']
        d = e1.total.numel()
        names = ['n_accepted_trajectories', 'n_attempted_trajectories', 'n_divergences', 'n_target_gradient_calls',
                 'n_target_calls', 'n_nonfinite_log_ratios']
        if hasattr(st, 'n_accepted_jumps'):
            names += ['n_accepted_jumps', 'n_attempted_jumps']
        pack = torch.cat([e1.total.reshape(-1).double(), e2.total.reshape(-1).double(),
                          torch.tensor([float(e1.n_seen)] + [float(getattr(st, k)) for k in names],
                                       dtype=torch.float64)])
        if tdist.get_backend(self.group) != 'gloo':
            pack = pack.to(torch.device('cuda', torch.cuda.current_device()))
        self.all_reduce_sum_(pack)
        pack = pack.cpu()
        e1.total = pack[:d].reshape(e1.total.shape).clone()
        e2.total = pack[d:2 * d].reshape(e2.total.shape).clone()
        e1.n_seen = e2.n_seen = int(round(float(pack[2 * d])))
        for i, k in enumerate(names):
            setattr(st, k, int(round(float(pack[2 * d + 1 + i]))))
        return st
