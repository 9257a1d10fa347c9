"""Fixed host-side cost of one sample() call at the C3 shape (what is left when the kernels take no time): wall time
for T = 1 vs T = 21 outer steps and a cProfile of the T = 1 call."""
import cProfile, gc, os, pstats, sys, time
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


def main():
    call(2); gc.collect(); gc.disable()
    t1 = min(call(1) for _ in range(5)); t21 = min(call(21) for _ in range(5))
    per = (t21 - t1) / 20
    print('T=1 %.3f ms, T=21 %.3f ms: %.4f ms per outer step, fixed %.3f ms per call' % (t1 * 1e3, t21 * 1e3, per * 1e3, (t1 - per) * 1e3), flush=True)
    pr = cProfile.Profile(); pr.enable(); call(1); pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(30)


if __name__ == '__main__':
    main()
