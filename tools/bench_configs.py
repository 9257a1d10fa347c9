"""Throughput + sanity of every BASELINE.json config on one MI355X (C5: one GPU's shard of 32768 chains).
Prints one JSON line per config; `python tools/bench_configs.py > gpurun_out/configs.jsonl`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import sample
from nfmc_amd.potentials import Funnel, SumOfSquares


def timed(fn, reps=3):
    best, out = None, None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, out


def report(name, n, transitions, dt, out, extra=None):
    st = out.statistics
    line = {'config': name, 'chain_steps_per_s': n * transitions / dt, 'seconds': dt, 'n_chains': n,
            'transitions_per_chain': transitions, 'acceptance': st.acceptance_rate,
            'mean_abs_max': float(out.mean.abs().max()), 'variance_mean': float(out.variance.mean())}
    if hasattr(st, 'jump_acceptance_rate'):
        line['jump_acceptance'] = st.jump_acceptance_rate
    line.update(extra or {})
    print(json.dumps(line), flush=True)


def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    # C1: README example (plumbing): callable target, d=25, 100 chains, 200 outer x (100 MALA + 1 jump), samples stored
    torch.manual_seed(0)
    f = lambda: sample(lambda x: torch.sum(x ** 2, dim=1), event_shape=(25,), strategy='jump_mala', flow='realnvp',
                       n_chains=100, n_iterations=200, show_progress=False, seed=0)
    dt, out = timed(f, 2)
    report('C1 jump_mala d=25 n=100 T=200 (README)', 100, 200 * 101, dt, out, {'samples_shape': list(out.samples.shape)})
    # C2: imh d=64 n=8192 T=1000
    x0 = (torch.randn(8192, 64, generator=g) * 0.7071).to(dev)
    torch.manual_seed(1)
    f = lambda: sample(SumOfSquares((64,)), strategy='imh', flow='realnvp', x0=x0, n_iterations=1000, show_progress=False,
                       seed=0, param_kwargs={'store_samples': False})
    dt, out = timed(f)
    report('C2 imh d=64 n=8192 T=1000', 8192, 1000, dt, out)
    # C4: neutra_hmc funnel d=128 n=65536 L=10, conditioner 128x2 (matrix cores)
    x0 = (0.5 * torch.randn(65536, 128, generator=g)).to(dev)
    torch.manual_seed(1)
    f = lambda: sample(Funnel((128,), 3.0), strategy='neutra_hmc', flow='realnvp', x0=x0, n_iterations=10,
                       flow_kwargs={'conditioner_kwargs': {'n_hidden': 128, 'n_layers': 2}}, show_progress=False, seed=0,
                       inner_kernel_kwargs={'n_leapfrog_steps': 10, 'step_size': 0.02}, param_kwargs={'store_samples': False})
    dt, out = timed(f, 2)
    flops = 65536 * 10 * (10 + 0.1) * 2 * 245760   # gradient evaluations x MACs x 2
    report('C4 neutra_hmc funnel d=128 n=65536 L=10 H=128', 65536, 10, dt, out, {'mfma_tflops': flops / dt / 1e12})
    # C5 (one GPU's shard): jump_hmc d=256 n=32768 K=5 L=20 T=20
    x0 = (torch.randn(32768, 256, generator=g) * 0.7071).to(dev)
    torch.manual_seed(1)
    f = lambda: sample(SumOfSquares((256,)), strategy='jump_hmc', flow='realnvp', x0=x0, n_iterations=20, show_progress=False,
                       seed=0, inner_kernel_kwargs={'n_leapfrog_steps': 20, 'step_size': 0.05},
                       param_kwargs={'store_samples': False})
    dt, out = timed(f)
    report('C5 shard jump_hmc d=256 n=32768 K=5 L=20 T=20', 32768, 20 * 6, dt, out)


if __name__ == '__main__':
    main()
