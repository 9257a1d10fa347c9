"""Per-coordinate variance deviations of the C5 shard run per rank of an 8-way split (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import sample
from nfmc_amd.dist import Shard
from nfmc_amd.potentials import SumOfSquares
d, n = 256, 32768


class Rows:
    shape = (8 * n, d)

    def __init__(self, seed):
        self.seed = seed

    def __getitem__(self, sl):
        return torch.randn(n, d, generator=torch.Generator().manual_seed(self.seed + sl.start // n)) * 0.7071


for rank in range(8):
    for seed in (0, 5):
        sh = Shard(rank=rank, world=8)
        sh.merge_statistics = lambda s_: s_
        torch.manual_seed(1)
        out = sample(SumOfSquares((d,)), strategy='jump_hmc', flow='realnvp', x0=Rows(100), n_iterations=8, show_progress=False,
                     seed=seed, shard=sh, param_kwargs={'store_samples': False},
                     inner_kernel_kwargs={'n_leapfrog_steps': 20, 'step_size': 0.05})
        dev = (out.variance - 0.5) / 0.5
        i = int(dev.abs().argmax())
        print('rank', rank, 'seed', seed, 'acc %.3f' % out.statistics.acceptance_rate, 'rel std %.2e' % float(dev.std()),
              'max %.2e at %d' % (float(dev[i]), i), 'mean dev %.2e' % float(dev.mean()),
              'mean|x| %.2e' % float(out.mean.abs().max()), flush=True)
