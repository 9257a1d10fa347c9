"""mala_kernel<8,8> at d = 64: time of a 100-transition launch against the number of chains.  90 VGPRs leave 5 waves per
SIMD = 5120 resident waves = 40960 chains; C3's 65536 chains are 8192 waves = one full round + a 3-waves-per-SIMD rest."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd.sample import create_sampler
from nfmc_amd.potentials import SumOfSquares

dev = torch.device('cuda', 0)
for n in [int(v) for v in os.environ.get("PROBE_NS", "20480,40960,49152,57344,65536,81920,122880,131072").split(",")]:
    x0 = (torch.randn(n, 64) * 0.7071).to(dev)
    s = create_sampler(SumOfSquares((64,)), strategy='mala', param_kwargs={'n_iterations': 2000, 'store_samples': False})
    s.seed = 0
    s.sample(x0, show_progress=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.sample(x0, show_progress=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('n %6d  waves %5d  %.4f ms per 100 transitions  %.3e chain-steps/s' % (n, n * 8 // 64, dt / 20 * 1e3, n * 2000 / dt), flush=True)
