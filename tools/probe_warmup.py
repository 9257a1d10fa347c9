"""Cost of a warmup at the C3 shape (65536 x 64, MALA): 100 tuning transitions vs 100 sampling transitions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd.samplers import mcmc
from nfmc_amd.potentials import SumOfSquares

d, n, K = 64, 65536, 100
x0 = (torch.randn(n, d, generator=torch.Generator().manual_seed(0)) * 0.7071).cuda()


def timed(fn, reps=4):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best * 1e3


def sampling():
    s = mcmc.MALA((d,), SumOfSquares((d,)), None, mcmc.LangevinParameters(n_iterations=K, store_samples=False))
    s.seed = 1
    s.sample(x0, show_progress=False)


def warmup(every, device=True):
    os.environ['NFMC_TUNE_DEVICE'] = '1' if device else '0'
    s = mcmc.MALA((d,), SumOfSquares((d,)), None, mcmc.LangevinParameters(n_warmup_iterations=K, store_samples=False, tune_every=every))
    s.seed = 1
    s.warmup(x0, show_progress=False)
    return s.kernel.step_size


t_s = timed(sampling)
print('sampling, %d transitions: %.2f ms' % (K, t_s))
for every, device, one_level in [(1, False, False), (1, True, True), (1, True, False), (5, True, False), (10, True, False), (25, True, False)]:
    if one_level:   # round 2-3: tuning launches of 256 workgroups, one workgroup folds their slabs
        os.environ['NFMC_TUNE_ONE_LEVEL'] = '1'
    else:
        os.environ.pop('NFMC_TUNE_ONE_LEVEL', None)
    t = timed(lambda: warmup(every, device))
    print('warmup tune_every=%-3d %-6s %-22s: %.2f ms  (%.1fx sampling)  tuned step %.4f' % (
        every, 'device' if device else 'host', '(256 workgroups, 1 fold)' if one_level else '', t, t / t_s, warmup(every, device)))
