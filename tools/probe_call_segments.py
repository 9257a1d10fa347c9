"""Host time of the pieces of one sample() call at the C3 shape (T = 2, fresh sampler per call like bench.py): wall-clock
accumulators around the main host-side functions over 200 calls."""
import collections, functools, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import hip
from nfmc_amd.sample import create_sampler
from nfmc_amd.potentials import SumOfSquares
from nfmc_amd.samplers import common, jump, mcmc
from nfmc_amd import containers, flows

acc = collections.defaultdict(float); cnt = collections.Counter()


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or '%s.%s' % (getattr(obj, '__name__', obj), name)

    @functools.wraps(f)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t0; cnt[label] += 1
    setattr(obj, name, g)


wrap(common.Run, '__init__', 'Run.__init__')
wrap(hip.DeviceStats, '__init__', 'DeviceStats.__init__')
wrap(hip.DeviceStats, 'host_totals', 'DeviceStats.host_totals')
wrap(jump, 'resolve_target'); wrap(jump, 'flow_mh_supported'); wrap(jump, 'launch_flow_mh')
wrap(mcmc.Langevin, '_launch', 'Langevin._launch')
wrap(flows.RealNVP, 'packed', 'RealNVP.packed')
wrap(containers.MCMCStatistics, 'absorb_device_sums', 'absorb_device_sums')
wrap(jump, 'JumpNFMCOutput')

dev = torch.device('cuda', 0)
x0 = (torch.randn(65536, 64) * 0.7071).to(dev)


def call():
    torch.manual_seed(1)
    s = create_sampler(SumOfSquares((64,)), strategy='jump_mala', flow='realnvp',
                       param_kwargs={'n_iterations': 2, 'store_samples': False}, inner_param_kwargs={'n_iterations': 100})
    s.seed = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.sample(x0, show_progress=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); return t1 - t0, time.perf_counter() - t0


for _ in range(5):
    call()
gc.collect(); gc.disable(); acc.clear(); cnt.clear()
tot_host = tot = 0.0
N = 200
for _ in range(N):
    a, b = call(); tot_host += a; tot += b
print('per call: %.1f us until sample() returns, %.1f us until the stream is idle' % (tot_host / N * 1e6, tot / N * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-28s %7.1f us per call (%d calls)' % (k, v / N * 1e6, cnt[k] // N))
