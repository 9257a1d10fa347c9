"""C5 shard with the flow refit switched on (jump.py:193-201): wall time of the refit path at full size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import sample
from nfmc_amd.potentials import SumOfSquares

def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    d, n = 256, 32768
    x0 = (torch.randn(n, d, generator=g) * 0.7071).to(dev)
    for fit in (False, True):
        best = None
        for rep in range(2):
            torch.manual_seed(1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = sample(SumOfSquares((d,)), strategy='jump_hmc', flow='realnvp', x0=x0, n_iterations=14, show_progress=False,
                         seed=0, inner_kernel_kwargs={'n_leapfrog_steps': 20, 'step_size': 0.05},
                         param_kwargs={'store_samples': False, 'fit_nf': fit, 'n_jumps_before_training': 10,
                                       'flow_fit_kwargs': {'n_epochs': 20, 'early_stopping': False}})
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        st = out.statistics
        print('fit_nf=%s: %.1f ms for 14 outer iterations (4 refits of 20 epochs when on); jump acceptance %.4f var %.4f' % (
            fit, best * 1e3, st.jump_acceptance_rate, float(out.variance.mean())), flush=True)

if __name__ == '__main__':
    main()
