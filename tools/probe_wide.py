"""Wide events on the matrix cores (csrc/mfma_wide.hip): time of the streamed forward / inverse / NeuTra-gradient kernels at
d = 256 / 512, conditioner 128 x 2, 65536 chains, next to what served these shapes before -- the one-chain-per-lane VALU
flow kernels (NFMC_FLOW_NO_MFMA=1; no kernel at d = 512) and torch autograd through the restatement for the gradient."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nfmc_amd import hip, flow_training  # noqa: E402
from nfmc_amd.flows import Flow, RealNVP  # noqa: E402
from nfmc_amd.potentials import Funnel  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    dev = hip.require_gpu()
    n = int(os.environ.get('N', 65536))
    for d in [int(v) for v in os.environ.get('D', '256,512').split(',')]:   # D=256: one shape (counter passes)
        torch.manual_seed(d)
        f = Flow(RealNVP((d,), n_layers=2, conditioner_kwargs={'n_hidden': 128, 'n_layers': 2}))
        x = torch.randn(n, d, device=dev)
        st, _keep = f.bijection.packed(dev)
        pd = Funnel((d,), 3.0).descriptor(dev)
        u = torch.empty(n, device=dev)
        g = torch.empty(n, d, device=dev)

        def grad():
            hip.check(hip.lib().nfmc_neutra_potential_grad_f32(C.byref(st), C.byref(pd), hip.ptr(x), n, hip.ptr(u), hip.ptr(g),
                                                               hip.stream()), 'grad')
        # multiply-adds of one gradient: inverse sweep (W1, Wh, W3) + reverse sweep (W1, Wh, W3, W3^T, Wh^T, W1^T) per layer
        macs = 2 * n * (3 * (d // 2) * 128 + 3 * 128 * 128 + 3 * 128 * d)
        t = timed(grad)
        print('d=%d  NeuTra gradient (mfma_wide)   %8.3f ms  %6.1f TFLOP/s' % (d, t, 2 * macs / t / 1e9))
        t = timed(lambda: f.bijection.forward(x))
        print('d=%d  forward (mfma_wide)           %8.3f ms  %6.1f TFLOP/s' % (d, t, 2 * 2 * n * ((d // 2) * 128 + 128 * 128 + 128 * d) / t / 1e9))
        t = timed(lambda: f.bijection.inverse(x))
        print('d=%d  inverse (mfma_wide)           %8.3f ms' % (d, t))
        os.environ['NFMC_FLOW_NO_MFMA'] = '1'
        try:
            t = timed(lambda: f.bijection.forward(x))
            print('d=%d  forward (VALU tile kernels)   %8.3f ms' % (d, t))
        except Exception as e:   # d = 512: the wave tile and the hidden buffer exceed the LDS
            print('d=%d  forward (VALU tile kernels)   no kernel (%s)' % (d, type(e).__name__))
        del os.environ['NFMC_FLOW_NO_MFMA']
        # one NeuTra-HMC trajectory (L = 10) of 32768 chains through the sampler: the fused trajectory kernel (round 4)
        from nfmc_amd.sample import create_sampler
        nt = 32768
        smp = create_sampler(Funnel((d,), 3.0), strategy='neutra_hmc', flow='realnvp',
                             flow_kwargs={'conditioner_kwargs': {'n_hidden': 128, 'n_layers': 2}},
                             inner_kernel_kwargs={'n_leapfrog_steps': 10, 'step_size': 0.02},
                             param_kwargs={'n_iterations': 20, 'store_samples': False})
        smp.kernel.flow.load_state_dict(f.state_dict())
        smp.seed = 0
        z0 = 0.5 * x[:nt]
        t = timed(lambda: smp.sample(z0, show_progress=False), 3)
        print('d=%d  neutra_hmc, L = 10, %d chains: %8.3f ms per trajectory (20 in one call)' % (d, nt, t / 20))
        # the flow-proposal Metropolis step (imh: one inverse pass + U + accept per transition) on the streamed kernel (round 4)
        from nfmc_amd.potentials import SumOfSquares
        imh = create_sampler(SumOfSquares((d,)), strategy='imh', flow='realnvp',
                             flow_kwargs={'conditioner_kwargs': {'n_hidden': 128, 'n_layers': 2}},
                             param_kwargs={'n_iterations': 20, 'store_samples': False})
        imh.kernel.flow.load_state_dict(f.state_dict())
        imh.seed = 0
        t = timed(lambda: imh.sample(z0, show_progress=False), 3)
        print('d=%d  imh, %d chains: %8.3f ms per transition (20 in one call; forward pass of the first state included)' % (d, nt, t / 20))
        nb = min(n, 8192)
        f.to(dev)

        def autograd():
            z = x[:nb].clone().requires_grad_(True)
            xx, ld = flow_training.inverse_torch(f.bijection, z)
            x0 = xx[:, 0]
            uu = 0.5 * x0 ** 2 / 9.0 + 0.5 * torch.exp(-x0) * (xx[:, 1:] ** 2).sum(1) + 0.5 * (d - 1) * x0 - ld
            torch.autograd.grad(uu.sum(), z)
        t = timed(autograd, 3)
        print('d=%d  gradient by torch autograd    %8.3f ms for %d chains (x %d for %d)' % (d, t, nb, n // nb, n))


if __name__ == '__main__':
    main()
