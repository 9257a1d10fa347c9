"""IMH wall time for few chains (the reference's default is n_chains = 100): data-parallel vs sequential kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import sample
from nfmc_amd.potentials import SumOfSquares

def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    for n in (100, 1000, 8192, 16384):
        x0 = (torch.randn(n, 64, generator=g) * 0.7071).to(dev)
        for par in ('1', '0'):
            os.environ['NFMC_IMH_PARALLEL'] = par
            best = None
            for rep in range(3):
                torch.manual_seed(1)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = sample(SumOfSquares((64,)), strategy='imh', flow='realnvp', x0=x0, n_iterations=1000, show_progress=False,
                             seed=0, param_kwargs={'store_samples': False})
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print('n=%6d parallel=%s  %.3f ms  %.3g chain-steps/s  var %.4f' % (n, par, best * 1e3, n * 1000 / best, float(out.variance.mean())), flush=True)

if __name__ == '__main__':
    main()
