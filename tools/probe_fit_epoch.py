"""One epoch of the device fit at the refit shapes (profiles/r04_refit.txt): d, rows, validation rows -> microseconds per
epoch (gradient launch + fold launch, 200 epochs enqueued back to back, HIP events), the shuffled split of the refit buffer,
and a whole `Flow.fit(n_epochs=2)` call as the refit of jump.py:193-201 issues it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd.flow_training import DeviceFit
from nfmc_amd.flows import Flow, RealNVP
from nfmc_amd.tuning import train_val_split


def ev_ms(fn, reps=5):
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b)
        best = t if best is None else min(best, t)
    return best


def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    for d, n, nv, H in ((256, 4096, 4096, None), (256, 2867, 1229, None), (64, 4096, 1024, None), (64, 1024, 0, None), (128, 4096, 4096, 16),
                        (128, 4096, 4096, 128), (128, 1024, 0, 128), (64, 1024, 0, 64)):
        torch.manual_seed(1)
        ck = {'conditioner_kwargs': {'n_hidden': H, 'n_layers': 2}} if H else {}
        f = Flow(RealNVP((d,), **ck)).to(dev)
        x = (torch.randn(n, d, generator=g) * 0.7071).to(dev)
        xv = (torch.randn(max(nv, 1), d, generator=g) * 0.7071).to(dev)
        fit = DeviceFit(f.bijection, dev, n + nv, lr=0.01)
        if nv:
            fit.set_validation(xv[:nv])
        ctl = fit.control(200, False, 50, True)
        fit.run_calls(ctl, x, 0, 5)
        per_epoch = ev_ms(lambda: fit.run_calls(ctl, x, 0, 200)) / 200
        print('d=%d H=%d rows=%d val=%d: %.1f us per epoch (n_params %d)' % (d, f.bijection.n_hidden, n, nv, per_epoch * 1e3,
                                                                                fit.n_params), flush=True)
    # C4's warmup: the variational fit of the 128 x 2 conditioner at d = 128 to the funnel, 1024 latents per epoch
    from nfmc_amd.potentials import Funnel
    pot = Funnel((128,), 3.0)
    for path in ('0', '1'):
        os.environ['NFMC_FIT_TORCH'] = path
        torch.manual_seed(1)
        f = Flow(RealNVP((128,), conditioner_kwargs={'n_hidden': 128, 'n_layers': 2})).to(dev)
        f.variational_fit(lambda v: -pot(v), n_epochs=3, lr=0.01, n_samples=1024, early_stopping=False, potential=pot)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f.variational_fit(lambda v: -pot(v), n_epochs=200, lr=0.01, n_samples=1024, early_stopping=False, potential=pot)
        torch.cuda.synchronize()
        print('Flow.variational_fit, d=128, H=128x2, 200 epochs of 1024 latents (%s): %.1f ms' % (
            'eager torch loop' if path == '1' else 'device', 1e3 * (time.perf_counter() - t0)), flush=True)
    os.environ['NFMC_FIT_TORCH'] = '0'
    # the refit as the sampler issues it: split of a (5, 32768, 256) block + Flow.fit(2 epochs)
    d, K, nch = 256, 5, 32768
    torch.manual_seed(1)
    f = Flow(RealNVP((d,))).to(dev)
    buf = (torch.randn(K, nch, d, generator=g) * 0.7071).to(dev)
    xt, xv = train_val_split(buf, 0.7, 4096, 4096)
    f.fit(x_train=xt, x_val=xv, n_epochs=2, show_progress=False)
    print('train_val_split (5 x 32768 x 256 -> 4096 + 4096 rows): %.1f us' % (1e3 * ev_ms(lambda: train_val_split(buf, 0.7, 4096, 4096))))
    print('Flow.fit(n_epochs=2), 4096 + 4096 rows, GPU time: %.1f us' % (1e3 * ev_ms(lambda: f.fit(x_train=xt, x_val=xv, n_epochs=2, show_progress=False))))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        a, b = train_val_split(buf, 0.7, 4096, 4096)
        f.fit(x_train=a, x_val=b, n_epochs=2, show_progress=False)
    torch.cuda.synchronize()
    print('split + Flow.fit(n_epochs=2), wall per refit over 50: %.1f us' % ((time.perf_counter() - t0) / 50 * 1e6))


if __name__ == '__main__':
    main()
