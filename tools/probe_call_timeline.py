"""GPU-side timeline of sample() calls at the C3 shape (run under `rocprofv3 --kernel-trace --memory-copy-trace`):
five calls of T = 2 with a fresh sampler each, like bench.py's repetitions."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd.sample import create_sampler
from nfmc_amd.potentials import SumOfSquares

dev = torch.device('cuda', 0)
x0 = (torch.randn(65536, 64) * 0.7071).to(dev)


def call(T):
    torch.manual_seed(1)
    s = create_sampler(SumOfSquares((64,)), strategy='jump_mala', flow='realnvp',
                       param_kwargs={'n_iterations': T, 'store_samples': False}, inner_param_kwargs={'n_iterations': 100})
    s.seed = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.sample(x0, show_progress=False)
    torch.cuda.synchronize(); return time.perf_counter() - t0


call(2); gc.collect(); gc.disable()
for _ in range(5):
    print('%.3f ms' % (call(2) * 1e3))
