"""C3 shape (jump_mala d=64, 65536 chains, 100 inner steps) for several conditioner widths / flow kinds:
which flow kernel serves the jump and what it costs per outer step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import sample
from nfmc_amd.potentials import SumOfSquares

def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    d, n = 64, 65536
    x0 = (torch.randn(n, d, generator=g) * 0.7071).to(dev)
    for flow, kw in [('realnvp', {}), ('realnvp', {'conditioner_kwargs': {'n_hidden': 8}}),
                     ('realnvp', {'conditioner_kwargs': {'n_hidden': 16}}), ('realnvp', {'conditioner_kwargs': {'n_hidden': 32}}),
                     ('realnvp', {'conditioner_kwargs': {'n_hidden': 64}}), ('nice', {}), ('c-rqnsf', {}),
                     ('c-rqnsf', {'conditioner_kwargs': {'n_hidden': 16}})]:
        best = None
        for rep in range(3):
            torch.manual_seed(1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = sample(SumOfSquares((d,)), strategy='jump_mala', flow=flow, flow_kwargs=kw, x0=x0, n_iterations=10,
                         show_progress=False, seed=0, inner_param_kwargs={'n_iterations': 100},
                         param_kwargs={'store_samples': False})
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print('%-8s %-44s %.3f ms per outer step (MALA part ~0.33)  var %.4f' % (flow, kw, best / 10 * 1e3, float(out.variance.mean())), flush=True)

if __name__ == '__main__':
    main()
