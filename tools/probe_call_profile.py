"""Where the fixed host-side cost of one sample() call goes at the C3 shape: cProfile over 200 calls of T = 1."""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd.sample import create_sampler
from nfmc_amd.potentials import SumOfSquares

dev = torch.device('cuda', 0)
x0 = (torch.randn(65536, 64) * 0.7071).to(dev)
torch.manual_seed(1)
s = create_sampler(SumOfSquares((64,)), strategy='jump_mala', flow='realnvp',
                   param_kwargs={'n_iterations': 1, 'store_samples': False}, inner_param_kwargs={'n_iterations': 100})
s.seed = 0
for _ in range(5):
    s.sample(x0, show_progress=False)
gc.collect(); gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200):
    s.sample(x0, show_progress=False)
torch.cuda.synchronize(); print('%.4f ms per call of T = 1' % ((time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    s.sample(x0, show_progress=False)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
