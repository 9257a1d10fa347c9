"""C5 shard (jump_hmc d=256, 32768 chains, K=5, L=20): wall time per outer step; env knobs apply."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import sample
from nfmc_amd.potentials import SumOfSquares

def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    d = int(os.environ.get('C5_D', '256'))
    n = int(os.environ.get('C5_N', '32768'))
    x0 = (torch.randn(n, d, generator=g) * 0.7071).to(dev)
    best = None
    for rep in range(4):
        torch.manual_seed(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = sample(SumOfSquares((d,)), strategy='jump_hmc', flow='realnvp', x0=x0, n_iterations=20, show_progress=False,
                     seed=0, inner_kernel_kwargs={'n_leapfrog_steps': 20, 'step_size': 0.05},
                     param_kwargs={'store_samples': False})
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print('d=%d n=%d  %.1f us per outer step  %.3g chain-steps/s  var %.4f jump_acc %.4f' % (
        d, n, best / 20 * 1e6, n * 120 / best, float(out.variance.mean()), out.statistics.jump_acceptance_rate), flush=True)

if __name__ == '__main__':
    main()
