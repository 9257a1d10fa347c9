import sys, time, torch
sys.path.insert(0, '.')
from nfmc_amd.potentials import Funnel
from nfmc_amd.sample import create_sampler
d, n = 128, 65536
dev = torch.device('cuda', 0)
pot = Funnel((d,), 3.0)
torch.manual_seed(1)
def make(h, T=10):
    return create_sampler(pot, strategy='neutra_hmc', flow='realnvp', flow_kwargs={'conditioner_kwargs': {'n_hidden': 128, 'n_layers': 2}},
                          inner_kernel_kwargs={'n_leapfrog_steps': 10, 'step_size': h}, param_kwargs={'n_iterations': T, 'store_samples': False})
s = make(0.02)
flow = s.kernel.flow
t0 = time.time()
flow.variational_fit(lambda v: -pot(v), n_epochs=int(sys.argv[1]) if len(sys.argv) > 1 else 200, lr=0.01, n_samples=1024, early_stopping=False, keep_best_weights=True, show_progress=False)
torch.cuda.synchronize(); print('fit s', time.time() - t0, flush=True)
state = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}
g = torch.Generator().manual_seed(0)
z0 = torch.randn(n, d, generator=g).to(dev)
for h in (0.02, 0.05, 0.1, 0.2, 0.3):
    s = make(h, T=20)
    s.kernel.flow.load_state_dict(state)
    s.seed = 0
    out = s.sample(z0, show_progress=False)
    zl = out.running_samples.last_sample
    x, _ = s.kernel.flow.bijection.inverse(zl)
    print('h', h, 'acc %.3f' % out.statistics.acceptance_rate, 'z var %.3f' % float(out.variance.mean()), 'x0 std %.3f' % float(x[:, 0].std()), 'finite', bool(torch.isfinite(zl).all()), flush=True)
