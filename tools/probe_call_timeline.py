"""GPU-side timeline of sample() calls at the C3 shape (run under `rocprofv3 --kernel-trace --memory-copy-trace`):
five calls of T = 2 with a fresh sampler each, like bench.py's repetitions.  PROBE_CFG=C2: the C2 call instead (imh,
8192 chains, 1000 transitions, bench.py's sampler)."""
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


if os.environ.get('PROBE_CFG') == 'C2':
    import bench
    cfg = bench.CONFIGS['C2']
    x0 = torch.randn(cfg['n_per_gpu'], cfg['d']).to(dev)

    def call(T):
        s = bench.build_sampler(cfg, T)
        s.seed = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.sample(x0, show_progress=False)
        torch.cuda.synchronize(); return time.perf_counter() - t0

    call(20); gc.collect(); gc.disable()
    for _ in range(5):
        print('%.3f ms' % (call(20) * 1e3))
else:
    call(2); gc.collect(); gc.disable()
    for _ in range(5):
        print('%.3f ms' % (call(2) * 1e3))
