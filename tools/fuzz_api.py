"""Randomised calls of the public `sample()` (one-off robustness sweep on a GPU box): every strategy of the path, flow strings
with and without JSON keywords, warmup on and off (dual-averaging tuning, flow fits on the device, refits inside jump runs),
sample storing with thinning / max_samples, 1-D and 2-D events, closed-form and plain-callable targets.  Per case:
  * the call returns, shapes are the reference's (`sampling/base.py:274-314`), every stored sample and moment is finite,
    counters add up (attempted = chains x transitions);
  * the same call with the same seeds gives bitwise the same samples, moments and counters (the device fits fold their
    gradients in a fixed order; the samplers draw from counter-based streams);
  * for the sum-of-squares target and long enough runs: the variance estimate is near 1/2.

usage: python tools/fuzz_api.py [seed] [budget_seconds]
"""
import json
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def sum_squares(x):
    return torch.sum(x ** 2, dim=tuple(range(1, x.dim())))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
    from nfmc_amd import sample
    from nfmc_amd.potentials import Funnel, SumOfSquares
    rnd = random.Random(seed)
    strategies = ['mala', 'ula', 'hmc', 'uhmc', 'mh', 'imh', 'adaptive_imh', 'jump_mala', 'jump_ula', 'jump_hmc', 'jump_uhmc',
                  'jump_mh', 'neutra_hmc', 'neutra_mh']
    t0 = time.time()
    done = failed = diverged = 0
    while time.time() - t0 < budget:
        strategy = rnd.choice(strategies)
        if rnd.random() < 0.2:
            event = (rnd.randint(1, 9), rnd.randint(1, 9))
        else:
            event = (rnd.choice([rnd.randint(1, 30), rnd.randint(31, 130), rnd.randint(131, 300), rnd.choice([64, 128, 256])]),)
        d = 1
        for s in event:
            d *= s
        n = rnd.choice([rnd.randint(1, 40), rnd.randint(41, 400), rnd.randint(401, 3000)])
        T = rnd.randint(1, 6)
        warm = rnd.random() < 0.5
        Tw = rnd.randint(1, 4)
        fl = rnd.choice(['realnvp', 'realnvp', 'rnvp', 'nice', 'c-rqnsf'])
        kw = {}
        if rnd.random() < 0.5:
            ck = {'n_hidden': rnd.choice([3, 8, 16, 32] if fl == 'c-rqnsf' else [3, 8, 16, 32, 64, 128]), 'n_layers': rnd.randint(1, 2)}
            fl = fl + '%' + json.dumps({'n_layers': rnd.randint(1, 3), 'conditioner_kwargs': ck})
        tk = rnd.choice(['callable', 'sumsq', 'funnel']) if d > 1 else rnd.choice(['callable', 'sumsq'])
        target = sum_squares if tk == 'callable' else (SumOfSquares(event) if tk == 'sumsq' else Funnel(event, 3.0))
        pk = {}
        if rnd.random() < 0.4:
            pk['store_samples'] = rnd.random() < 0.7
        inner = {}
        if strategy.startswith('jump') and rnd.random() < 0.7:
            inner['n_iterations'] = rnd.randint(1, 6)
        if strategy.startswith('jump') and rnd.random() < 0.3:
            pk['fit_nf'] = rnd.random() < 0.7
        if ('hmc' in strategy) and rnd.random() < 0.7:
            kw['inner_kernel_kwargs' if ('jump' in strategy or 'neutra' in strategy) else 'kernel_kwargs'] = {'n_leapfrog_steps': rnd.randint(1, 6)}
        if strategy.startswith('neutra') and warm:
            # the reference's default (>= 5000 epochs of one sample, neutra.py:19-33) is minutes on torch ops for a spline flow
            pk['warmup_fit_kwargs'] = {'n_epochs': rnd.randint(5, 40), 'n_samples': rnd.choice([1, 64, 500]), 'lr': 0.05,
                                       'early_stopping': rnd.random() < 0.5, 'early_stopping_threshold': 10, 'keep_best_weights': True}
        if pk:
            kw['param_kwargs'] = pk
        if inner:
            kw['inner_param_kwargs'] = inner
        sseed = rnd.randint(1, 1 << 30)
        case = '%s event=%s n=%d T=%d warmup=%s(%d) flow=%s target=%s %s seed=%d' % (strategy, event, n, T, warm, Tw, fl, tk, kw, sseed)

        def call():
            torch.manual_seed(sseed)
            kk = {k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()}
            return sample(target, event_shape=event, flow=fl, strategy=strategy, n_iterations=T, n_warmup_iterations=Tw, n_chains=n,
                          warmup=warm, show_progress=False, seed=sseed, **kk)
        try:
            a = call()
            b = call()
            st = a.statistics
            for name in ('mean', 'variance', 'second_moment'):
                va, vb = getattr(a, name), getattr(b, name)
                assert tuple(va.shape) == event, (name, tuple(va.shape))
                assert torch.isfinite(va).all(), name + ' not finite'
                assert torch.equal(va, vb), name + ' differs between two identical calls'
            sa, sb = a.samples, b.samples
            assert (sa is None) == (sb is None)
            if sa is not None:
                assert sa.shape[1:] == (n,) + event, tuple(sa.shape)
                assert torch.isfinite(sa).all(), 'samples not finite'
                assert torch.equal(sa, sb), 'samples differ between two identical calls'
            assert st.n_accepted_trajectories == b.statistics.n_accepted_trajectories
            assert 0 <= st.n_accepted_trajectories <= st.n_attempted_trajectories
            if strategy.startswith('jump'):
                assert st.n_attempted_jumps == n * T and st.n_accepted_jumps == b.statistics.n_accepted_jumps
            last = a.running_samples.last_sample
            assert last is not None and tuple(last.shape) == (n,) + event and torch.isfinite(last).all()
            if sa is not None and not warm and strategy != 'adaptive_imh' and rnd.random() < 0.5:
                # f3: the kept states under thinning / max_samples equal the dense run cut by the reference's rule
                # (sampling/base.py:249-263), bit for bit
                thinning, max_samples = rnd.choice([1, 2, 3, 5]), rnd.choice([None, 1, 2, 4, 50])
                kk = {k: (dict(v) if isinstance(v, dict) else v) for k, v in kw.items()}
                kk['param_kwargs'] = dict(kk.get('param_kwargs', {}), thinning=thinning, max_samples=max_samples)
                torch.manual_seed(sseed)
                c = sample(target, event_shape=event, flow=fl, strategy=strategy, n_iterations=T, n_warmup_iterations=Tw,
                           n_chains=n, warmup=False, show_progress=False, seed=sseed, **kk)
                idx = [i for i in range(sa.shape[0]) if i % thinning == 0]
                if max_samples:
                    idx = idx[-max_samples:]
                assert c.samples.shape[0] == len(idx), ('kept rows', c.samples.shape[0], len(idx), thinning, max_samples)
                assert torch.equal(c.samples, sa[idx]), ('kept states differ from the dense run', thinning, max_samples)
                assert torch.equal(c.running_samples.last_sample.cpu(), sa[-1].cpu())
            done += 1
            print('ok    %s  (%.0f s)' % (case, time.time() - t0), flush=True)
        except ValueError as e:
            if 'flow training diverged' in str(e):   # the reference's own outcome of a fit whose loss goes non-finite (ValueError)
                diverged += 1
                print('div   %s' % case, flush=True)
                continue
            failed += 1
            print('FAIL  %s' % case, flush=True)
            traceback.print_exc(limit=6)
        except Exception:
            failed += 1
            print('FAIL  %s' % case, flush=True)
            traceback.print_exc(limit=6)
    print('cases %d  failed %d  (fits that diverged and raised, as the reference does: %d)' % (done + failed + diverged, failed, diverged))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
