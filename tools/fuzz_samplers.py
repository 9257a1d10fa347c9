"""Randomised shapes through the native-stream parity checks (one-off sweep on a GPU box; tests/test_gpu_parity.py pins fixed
lists).  A case draws a strategy, an event size, a chain count, a flow (coupling layers, conditioner width and depth, NICE or
RealNVP) and a potential, runs the package's sampler on the GPU and the CPU restatement (oracle/samplers.py) on the same
Philox streams, and compares the stored samples: the share of chains that follow the oracle to 3e-4 at EVERY stored step must
exceed 0.95 (a near-tie flip of one accept decision changes the whole later trajectory of that chain), acceptance counts within
2 %, and the flow passes (forward, inverse, log-density) of the case's flow against the oracle's.

usage: python tools/fuzz_samplers.py [seed] [budget_seconds]
"""
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def flows(d, nl, nh, cl, nice, seed, spread):
    from nfmc_amd.flows import Flow, NICE, RealNVP
    from oracle import flow as oflow
    ck = {'n_layers': cl}
    if nh is not None:
        ck['n_hidden'] = nh
    of = oflow.perturb_(oflow.Flow((oflow.NICE if nice else oflow.RealNVP)((d,), n_layers=nl, conditioner_kwargs=ck)), seed, spread, 0.75)
    f = Flow((NICE if nice else RealNVP)((d,), n_layers=nl, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    return of, f


def check_flow(of, f, d, n, gen):
    x = torch.randn(n, d, generator=gen)
    with torch.no_grad():
        z_o, ld_o = of.bijection.forward(x)
        xi_o, ldi_o = of.bijection.inverse(x)
        lp_o = of.log_prob(x)
    z, ld = f.bijection.forward(x)
    xi, ldi = f.bijection.inverse(x)
    tol = 1e-4 * max(1.0, d / 64)
    np.testing.assert_allclose(z.cpu().numpy(), z_o.numpy(), atol=tol, rtol=2e-5)
    np.testing.assert_allclose(ld.cpu().numpy(), ld_o.numpy(), atol=2 * tol, rtol=2e-5)
    np.testing.assert_allclose(xi.cpu().numpy(), xi_o.numpy(), atol=tol, rtol=5e-5)
    np.testing.assert_allclose(ldi.cpu().numpy(), ldi_o.numpy(), atol=2 * tol, rtol=2e-5)
    np.testing.assert_allclose(f.log_prob(x).cpu().numpy(), lp_o.numpy(), atol=4 * tol, rtol=2e-5)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
    from nfmc_amd.containers import NFMCKernel
    from nfmc_amd.potentials import DiagonalGaussian, Funnel, SumOfSquares
    from nfmc_amd.samplers import imh, jump, mcmc, neutra
    from oracle import potentials as opot, samplers as osamp
    rnd = random.Random(seed)
    t0 = time.time()
    done = failed = 0
    while time.time() - t0 < budget:
        strategy = rnd.choice(['mala', 'hmc', 'jump_mala', 'jump_hmc', 'imh', 'neutra_hmc', 'neutra_hmc'])
        d = rnd.choice([rnd.randint(2, 40), rnd.randint(41, 130), rnd.randint(131, 512), rnd.choice([64, 128, 256, 512, 96, 160])])
        n = rnd.choice([rnd.randint(1, 70), rnd.randint(71, 300), rnd.randint(301, 1200)])
        nl, cl = rnd.randint(1, 3), rnd.randint(1, 2)
        nh = rnd.choice([None, rnd.randint(1, 8), rnd.randint(9, 32), rnd.choice([33, 64, 100, 128])])
        nice = rnd.random() < 0.15
        kind = rnd.choice(['sum', 'sum', 'funnel', 'diag'])
        if strategy == 'neutra_hmc' and d > 160 and (nh or 0) > 32 and d % 32:
            d = d // 32 * 32          # wide conditioner off the 32-multiples: composed from torch ops, covered by the suite
        if d * n > 150000:            # the oracle leg (autograd through the flow per leapfrog step) stays within seconds
            n = max(1, 150000 // d)
        T = rnd.randint(2, 4)
        K = rnd.randint(2, 5)
        L = rnd.randint(2, 6)
        sseed = rnd.randint(1, 1 << 30)
        gen = torch.Generator().manual_seed(sseed)
        case = '%s d=%d n=%d layers=%d H=%s x%d nice=%s pot=%s T=%d K=%d L=%d seed=%d' % (strategy, d, n, nl, nh, cl, nice, kind, T, K, L, sseed)
        try:
            if kind == 'sum':
                pot, opo = SumOfSquares((d,)), opot.sum_squares
            elif kind == 'funnel':
                pot, opo = Funnel((d,), 3.0), opot.funnel(3.0)
            else:
                mu, sg = torch.linspace(-0.5, 0.5, d), torch.linspace(0.6, 1.7, d)
                pot = DiagonalGaussian((d,), mu, sg)
                opo = pot          # its torch form is the definition
            x0 = 0.7 * torch.randn(n, d, generator=gen)
            tol = 3e-4 * max(1.0, d / 128)
            if strategy in ('mala', 'hmc'):
                h = 0.5 * d ** (-1 / 3) if strategy == 'mala' else 0.3 * d ** (-1 / 4)
                if strategy == 'mala':
                    s = mcmc.MALA((d,), pot, mcmc.LangevinKernel(event_size=d, step_size=h), mcmc.LangevinParameters(n_iterations=T * K))
                    tr = osamp.mcmc_sample(x0, opo, 'langevin', T * K, h, noise=osamp.PhiloxNoise(sseed))
                else:
                    s = mcmc.HMC((d,), pot, mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h), mcmc.HMCParameters(n_iterations=T))
                    tr = osamp.mcmc_sample(x0, opo, 'hmc', T, h, n_leapfrog=L, noise=osamp.PhiloxNoise(sseed))
                s.seed = sseed
                out = s.sample(x0, show_progress=False)
                acc_g, acc_o, att = out.statistics.n_accepted_trajectories, tr.n_accepted, out.statistics.n_attempted_trajectories
            else:
                of, f = flows(d, nl, nh, cl, nice, sseed % 1000, 0.2)
                check_flow(of, f, d, min(n, 200), gen)
                if strategy in ('jump_mala', 'jump_hmc'):
                    h = 0.5 * d ** (-1 / 3) if strategy == 'jump_mala' else 0.3 * d ** (-1 / 4)
                    if strategy == 'jump_mala':
                        s = jump.JumpMALA((d,), pot, NFMCKernel((d,), flow=f), jump.JumpNFMCParameters(n_iterations=T),
                                          mcmc.LangevinKernel(event_size=d, step_size=h), mcmc.LangevinParameters(n_iterations=K))
                        tr = osamp.jump_sample(x0, opo, of, 'langevin', T, K, h, noise=osamp.PhiloxNoise(sseed))
                    else:
                        s = jump.JumpHMC((d,), pot, NFMCKernel((d,), flow=f), jump.JumpNFMCParameters(n_iterations=T),
                                         mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h), mcmc.HMCParameters(n_iterations=K))
                        tr = osamp.jump_sample(x0, opo, of, 'hmc', T, K, h, n_leapfrog=L, noise=osamp.PhiloxNoise(sseed))
                    s.seed = sseed
                    out = s.sample(x0, show_progress=False)
                    acc_g, acc_o, att = out.statistics.n_accepted_jumps, tr.n_accepted_jumps, out.statistics.n_attempted_jumps
                elif strategy == 'imh':
                    s = imh.FixedIMH((d,), pot, imh.IMHKernel((d,), flow=f), imh.IMHParameters(n_iterations=T * K))
                    s.seed = sseed
                    out = s.sample(x0, show_progress=False)
                    tr = osamp.imh_sample(x0, opo, of, T * K, noise=osamp.PhiloxNoise(sseed))
                    acc_g, acc_o, att = out.statistics.n_accepted_trajectories, tr.n_accepted, out.statistics.n_attempted_trajectories
                else:
                    h = 0.2 * d ** (-1 / 4)
                    imd = torch.linspace(0.8, 1.3, d)
                    s = neutra.NeuTraHMC((d,), pot, mcmc.HMCKernel(event_size=d, n_leapfrog_steps=L, step_size=h, inv_mass_diag=imd.clone()),
                                         mcmc.HMCParameters(), neutra.NeuTraKernel((d,), flow=f), neutra.NeuTraParameters(n_iterations=T))
                    s.seed = sseed
                    z0 = 0.5 * x0
                    out = s.sample(z0, show_progress=False)
                    tr = osamp.neutra_hmc_sample(z0, opo, of, T, h, imd, L, noise=osamp.PhiloxNoise(sseed))
                    acc_g, acc_o, att = out.statistics.n_accepted_trajectories, tr.n_accepted, out.statistics.n_attempted_trajectories
            want = tr.stacked()
            got = out.samples.reshape(want.shape)
            assert torch.isfinite(got).all(), 'non-finite samples'
            same = (got - want).abs().amax(dim=(0, 2)) < tol
            share = float(same.float().mean())
            assert share > 0.95 or (n < 40 and int((~same).sum()) <= 2), 'share of chains following the oracle %.3f' % share
            assert abs(acc_g - acc_o) <= max(2, int(0.02 * att)), 'accepted %d vs %d of %d' % (acc_g, acc_o, att)
            done += 1
            print('ok    %s  share %.3f acc %d/%d  (%.0f s)' % (case, share, acc_g, att, time.time() - t0), flush=True)
        except Exception:
            failed += 1
            print('FAIL  %s' % case, flush=True)
            traceback.print_exc(limit=4)
    print('cases %d  failed %d' % (done + failed, failed))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
