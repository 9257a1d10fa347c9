"""f1 on the device: the maximum-likelihood (re)fit of the RealNVP proposal (jump.py:139-151,193-201; imh.py:166-170) through
`nfmc_flow_fit_step_f32` (csrc/fit_kernels.hip: hand-written reverse sweep with weight gradients + fused AdamW) against
autograd of the flow restatement and against the eager torch loop it replaces.

Tolerances: gradients of the mean NLL agree with autograd to 2e-4 relative to the largest entry of each parameter tensor
(fp32 sums over up to 700 rows in a different order; hardware exp / rcp / tanh); the first AdamW step to 2e-6 in the
parameters wherever the gradient is not negligible; the loss of each of 25 epochs to 1e-3 (first 8) / 2e-2 relative."""
import copy
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    return torch.device('cuda', 0)


def _flow(d, n_hidden, n_hl, n_layers, seed, nice=False):
    from nfmc_amd.flows import Flow, NICE, RealNVP
    from oracle import flow as oflow
    ck = {'n_hidden': n_hidden, 'n_layers': n_hl}
    of = oflow.perturb_(oflow.Flow((oflow.NICE if nice else oflow.RealNVP)((d,), n_layers=n_layers, conditioner_kwargs=ck)),
                        seed, 0.3, 0.8)
    f = Flow((NICE if nice else RealNVP)((d,), n_layers=n_layers, conditioner_kwargs=ck))
    f.load_state_dict(of.state_dict())
    return of, f


SHAPES = [  # d, n_hidden, hidden layers, coupling layers, rows, NICE
    (6, 4, 2, 2, 50, False), (7, 3, 1, 3, 64, False), (25, 4, 2, 2, 200, False), (64, 4, 2, 2, 333, False),
    (64, 8, 2, 2, 128, False), (64, 16, 1, 2, 70, False), (100, 7, 2, 3, 129, False), (128, 32, 2, 2, 65, False),
    (256, 7, 2, 2, 700, False), (16, 5, 2, 2, 90, True), (300, 7, 2, 2, 150, False), (512, 7, 2, 2, 130, False), (511, 6, 1, 3, 70, False),
    (1, 4, 2, 2, 30, False), (2, 1, 1, 4, 5000, False),
]


@pytest.mark.parametrize('d,H,nhl,nl,n,nice', SHAPES)
def test_nll_gradient_matches_autograd(dev, d, H, nhl, nl, n, nice):
    """One step with lr = 0, weight decay 0 and beta1 = 0 leaves the parameters alone and the first moment equal to the
    gradient: every entry against autograd of the CPU restatement (oracle/flow.py), the batch loss against its value."""
    from nfmc_amd.flow_training import DeviceFit
    of, f = _flow(d, H, nhl, nl, 3 + d, nice)
    x = (torch.randn(n, d, generator=torch.Generator().manual_seed(d)) * 0.8)
    f.to(dev)
    assert DeviceFit.supported(f.bijection, dev)
    fit = DeviceFit(f.bijection, dev, n, lr=0.0)
    fit.opt.beta1, fit.opt.weight_decay = 0.0, 0.0
    before = fit.params.clone()
    fit.step(x.to(dev), 0)
    torch.cuda.synchronize()
    loss_gpu, applied, _val = (float(v) for v in fit.status.cpu())
    assert applied == 1.0 and torch.equal(fit.params, before)          # lr = 0: nothing moved
    loss = -of.log_prob(x).mean()
    loss.backward()
    np.testing.assert_allclose(loss_gpu, float(loss.detach()), rtol=2e-5, atol=2e-5)
    g = copy.deepcopy(f)
    fit.write_back(fit.m, bijection=g.bijection)                        # the gradient, laid out as parameters
    want = dict(of.named_parameters())
    for name, p in g.named_parameters():
        w = want[name].grad
        if w is None or w.numel() == 0:      # d = 1: the source half is empty, W1 has no entries
            continue
        scale = max(float(w.abs().max()), 1e-3)
        np.testing.assert_allclose(p.detach().cpu().numpy(), w.numpy(), atol=2e-4 * scale, rtol=0, err_msg=name)
    # the padded entries of the blob (hidden units beyond n_hidden, alignment gaps) carry no gradient
    # (every slot a parameter entry maps to -- for conditioners presented on the matrix cores both orientations of a matrix)
    used = torch.zeros_like(fit.m, dtype=torch.bool)
    for _p, off, r, c, rs, cs in fit._layout(f.bijection):
        idx = off + torch.arange(r, device=dev)[:, None] * rs + torch.arange(c, device=dev)[None, :] * cs
        used[idx.reshape(-1)] = True
    assert int(used.sum()) > 0
    assert float(fit.m[~used].abs().max() if (~used).any() else 0.0) == 0.0


def test_device_steps_follow_torch_adamw(dev):
    """The fused AdamW step against torch.optim.AdamW on autograd gradients of the same restatement (flow_training.py:
    forward_torch), same data, same start.  AdamW divides by sqrt(v): a parameter whose gradient is ~0 moves by +-lr in a
    direction that rounding decides, and such differences feed back, so the two runs are compared where that cannot
    matter: the first step entry by entry wherever the gradient is not negligible, and the LOSS of every one of 25 epochs
    (flat directions do not move it), not the parameters after 25 epochs."""
    from nfmc_amd.flow_training import DeviceFit, _base_log_prob, forward_torch
    d, n, lr = 64, 1500, 0.02
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(n, d, generator=g) * torch.linspace(0.4, 1.6, d) + 0.3).to(dev)
    _of, fa = _flow(d, 4, 2, 2, 11)
    fa.to(dev)
    fb = copy.deepcopy(fa)
    fit = DeviceFit(fa.bijection, dev, n, lr=lr)
    opt = torch.optim.AdamW(fb.parameters(), lr=lr)
    la, lb = [], []
    for epoch in range(25):
        fit.step(x, epoch)
        la.append(float(fit.status[0]))
        opt.zero_grad()
        z, ld = forward_torch(fb.bijection, x)
        loss = -(_base_log_prob(z) + ld).mean()
        loss.backward()
        if epoch == 0:
            grads = {k: p.grad.detach().clone() for k, p in fb.named_parameters()}
        opt.step()
        lb.append(float(loss.detach()))
        if epoch == 0:
            fit.write_back()
            for (name, pa), (_n, pb) in zip(fa.named_parameters(), fb.named_parameters()):
                gr = grads[name]
                big = gr.abs() > 1e-4 * gr.abs().max()
                assert big.float().mean() > 0.5, name
                np.testing.assert_allclose(pa.detach()[big].cpu().numpy(), pb.detach()[big].cpu().numpy(), atol=2e-6, rtol=0,
                                           err_msg=name)
    la, lb = np.array(la), np.array(lb)
    assert lb[-1] < lb[0] - 1.0                                           # it learns: nats over 64 dimensions
    np.testing.assert_allclose(la[:8], lb[:8], rtol=1e-3)
    np.testing.assert_allclose(la, lb, rtol=2e-2, atol=5e-2)


def test_flow_fit_api_goes_through_the_device_path(dev, monkeypatch):
    """`Flow.fit(x, x_val=..., keep_best_weights=True)` (the call of jump.py:139-149) with the step on the device against
    the eager torch loop (NFMC_FIT_TORCH=1): both end at the same validation NLL."""
    from nfmc_amd import flow_training as ft
    d, n = 64, 1500
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, d, generator=g) * torch.linspace(0.4, 1.6, d) + 0.3
    xv = torch.randn(400, d, generator=g) * torch.linspace(0.4, 1.6, d) + 0.3
    res = []
    for torch_path in ('0', '1'):
        monkeypatch.setenv('NFMC_FIT_TORCH', torch_path)
        calls = []
        orig = ft.DeviceFit.run_calls
        monkeypatch.setattr(ft.DeviceFit, 'run_calls',
                            lambda self, ctl, xx, c0, k, _o=orig, **kw: (calls.extend(range(c0, c0 + k)), _o(self, ctl, xx, c0, k, **kw))[1])
        _of, f = _flow(d, 4, 2, 2, 11)
        n0 = float(-f.log_prob(xv.to(dev)).mean())
        f.fit(x, x_val=xv, n_epochs=25, lr=0.02, early_stopping=False, keep_best_weights=True, show_progress=False)
        res.append((n0, float(-f.log_prob(xv.to(dev)).mean()), len(calls)))
        monkeypatch.setattr(ft.DeviceFit, 'run_calls', orig)
    (n0, na, ca), (_n0, nb, cb) = res
    assert ca == 26 and cb == 0                  # the device path ran (25 epochs + the closing evaluation) / was switched off
    assert na < n0 - 1.0 and nb < n0 - 1.0
    assert abs(na - nb) < 2e-2 * abs(n0 - nb)


RKL_CASES = [  # d, n_hidden, hidden layers, coupling layers, rows, potential
    (6, 4, 2, 2, 40, 'sum'), (7, 3, 1, 3, 64, 'diag'), (25, 4, 2, 2, 130, 'sum'), (64, 8, 2, 2, 100, 'diag'),
    (16, 5, 2, 3, 70, 'funnel'), (128, 4, 2, 2, 65, 'funnel'), (256, 7, 2, 2, 200, 'sum'), (1, 4, 2, 2, 10, 'sum'),
    (400, 7, 2, 2, 90, 'diag'), (512, 8, 2, 2, 66, 'funnel'),
]


@pytest.mark.parametrize('d,H,nhl,nl,n,kind', RKL_CASES)
def test_reverse_kl_gradient_matches_autograd(dev, d, H, nhl, nl, n, kind):
    """The variational-fit step (imh.py:67-72, neutra.py:84-91): loss mean[log q(x) - log p(x)], x = f^-1(z), and its
    gradient with respect to every parameter from nfmc_flow_variational_fit_step_f32 (lr = 0, beta1 = 0: first moment =
    gradient) against autograd through the CPU restatement's inverse pass and the potential's torch form."""
    from nfmc_amd.flow_training import DeviceFit
    from nfmc_amd.potentials import DiagonalGaussian, Funnel, SumOfSquares
    from oracle import flow as oflow
    of, f = _flow(d, H, nhl, nl, 5 + d)
    f.to(dev)
    g0 = torch.Generator().manual_seed(100 + d)
    if kind == 'sum':
        pot = SumOfSquares((d,))
    elif kind == 'diag':
        pot = DiagonalGaussian((d,), torch.linspace(-0.5, 0.5, d), torch.linspace(0.6, 1.7, d))
    else:
        pot = Funnel((d,), 3.0)
    z = torch.randn(n, d, generator=g0)
    fit = DeviceFit(f.bijection, dev, n, lr=0.0)
    fit.opt.beta1, fit.opt.weight_decay = 0.0, 0.0
    fit.step_variational(z.to(dev), pot.descriptor(dev), 0)
    torch.cuda.synchronize()
    loss_gpu, applied, _val = (float(v) for v in fit.status.cpu())
    x, ld = of.bijection.inverse(z)
    loss = (of.base_log_prob(z) - ld + pot(x)).mean()
    loss.backward()
    assert applied == 1.0
    np.testing.assert_allclose(loss_gpu, float(loss.detach()), rtol=3e-5, atol=3e-5)
    gflow = copy.deepcopy(f)
    fit.write_back(fit.m, bijection=gflow.bijection)
    want = dict(of.named_parameters())
    for name, p in gflow.named_parameters():
        w = want[name].grad
        if w is None or w.numel() == 0:      # d = 1: the source half is empty, W1 has no entries
            continue
        scale = max(float(w.abs().max()), 1e-3)
        np.testing.assert_allclose(p.detach().cpu().numpy(), w.numpy(), atol=3e-4 * scale, rtol=0, err_msg=name)


def test_variational_fit_through_the_sampler_warmup(dev, monkeypatch):
    """FixedIMH.warmup (imh.py:60-75) on a target that `resolve_target` recognises: the variational fit runs on the device
    (spied), the proposal improves (IMH acceptance after the warmup far above the unfitted flow's), and with the device
    path switched off (NFMC_FIT_TORCH=1) the eager loop reaches a comparable proposal."""
    from nfmc_amd import flow_training as ft
    from nfmc_amd.sample import create_sampler
    d, n = 16, 512
    acc = {}
    for torch_path in ('0', '1'):
        monkeypatch.setenv('NFMC_FIT_TORCH', torch_path)
        calls = []
        orig = ft.DeviceFit.run_calls
        monkeypatch.setattr(ft.DeviceFit, 'run_calls',
                            lambda self, ctl, z, c0, k, _o=orig, **kw: (calls.append(int(z.shape[0])), _o(self, ctl, z, c0, k, **kw))[1])
        torch.manual_seed(3)
        s = create_sampler(target=lambda x: torch.sum(x ** 2, dim=-1), event_shape=(d,), strategy='imh',
                           param_kwargs={'n_iterations': 200, 'store_samples': False})
        s.seed = 1
        x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(2))
        cold = s.sample(x0, show_progress=False).statistics.acceptance_rate
        s.params.warmup_fit_kwargs.update(n_epochs=150, n_samples=256, lr=0.02)
        w = s.warmup(x0, show_progress=False)
        out = s.sample(w.running_samples.last_sample, show_progress=False)
        acc[torch_path] = (cold, out.statistics.acceptance_rate, len(calls))
        monkeypatch.setattr(ft.DeviceFit, 'run_calls', orig)
    (cold, warm, n_dev), (_c, warm_t, n_t) = acc['0'], acc['1']
    # early stopping (imh.py:27-36 defaults) may end the fit before 150 epochs; the host looks every 64 enqueued epochs
    assert 50 <= n_dev <= 150 and n_t == 0
    assert warm > max(0.3, 3 * cold) and warm_t > max(0.3, 3 * cold), acc
    assert abs(warm - warm_t) < 0.15, acc


def test_device_fit_early_stopping_best_weights_and_divergence(dev):
    """The loop semantics of flow_training._loop on the device path: early stopping counts epochs without improvement of the
    validation loss, the best weights come back, a non-finite loss raises ValueError (jump.py:150) and leaves the weights
    the flow came in with."""
    from nfmc_amd import flow_training as ft
    d = 16
    _of, f = _flow(d, 4, 2, 2, 2)
    f.to(dev)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(256, d, generator=g) * 0.5).to(dev)
    xv = (torch.randn(64, d, generator=g) * 3.0 + 2.0).to(dev)          # a validation set the training set says nothing about
    from nfmc_amd import hip
    calls = []
    orig = ft.DeviceFit.run_calls
    ft.DeviceFit.run_calls = lambda self, ctl, xx, c0, k, **kw: (calls.extend(range(c0, c0 + k)), orig(self, ctl, xx, c0, k, **kw))[1]
    try:
        f.fit(x, x_val=xv, n_epochs=400, lr=0.05, early_stopping=True, early_stopping_threshold=5, keep_best_weights=True,
              show_progress=False)
    finally:
        ft.DeviceFit.run_calls = orig
    st = f.bijection._device_fit.state_after(len(calls))
    # stopped early on the device (one applied step per live epoch), and the host stopped enqueuing at its next look
    assert calls == list(range(len(calls))) and len(calls) % 64 == 0 and len(calls) < 400
    assert st[hip.FIT_STOPPED] == 1.0 and 6 <= st[hip.FIT_APPLIED] < len(calls) and st[hip.FIT_BOOKED] == st[hip.FIT_APPLIED]
    assert st[hip.FIT_SINCE_BEST] == 6.0
    before = copy.deepcopy(f.state_dict())
    bad = x.clone()
    bad[3, 2] = float('nan')
    with pytest.raises(ValueError):
        f.fit(bad, n_epochs=5, show_progress=False)
    for k, v in f.state_dict().items():
        assert torch.equal(v, before[k]), k


def test_refit_inside_a_jump_run_uses_the_device_path(dev, monkeypatch):
    """jump.py:193-201 with fit_nf: the refits of a sampling run go through nfmc_flow_fit_epochs_f32 (spied), the proposal
    improves (jump acceptance rises from ~0 for the unfitted flow) and the statistics stay those of the target."""
    from nfmc_amd import flow_training as ft, sample
    from nfmc_amd.potentials import SumOfSquares
    d, n = 32, 2048
    calls = []
    orig = ft.DeviceFit.run_calls
    monkeypatch.setattr(ft.DeviceFit, 'run_calls',
                        lambda self, ctl, xx, c0, k, **kw: (calls.extend([int(xx.shape[0])] * min(k, ctl.n_epochs - c0)),
                                                            orig(self, ctl, xx, c0, k, **kw))[1])
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(0)) * 0.7071
    torch.manual_seed(1)
    out = sample(SumOfSquares((d,)), strategy='jump_mala', flow='realnvp', x0=x0, n_iterations=8, show_progress=False, seed=0,
                 inner_param_kwargs={'n_iterations': 10},
                 param_kwargs={'store_samples': False, 'fit_nf': True, 'n_jumps_before_training': 2,
                               'flow_fit_kwargs': {'n_epochs': 60, 'lr': 0.02}})
    assert len(calls) >= 60 and max(calls) <= 4096                      # train_val_split caps (sampling/base.py:46-61)
    st = out.statistics
    assert st.jump_acceptance_rate > 0.05, st.jump_acceptance_rate
    np.testing.assert_allclose(out.variance.numpy(), 0.5, rtol=5e-2)
    assert math.isfinite(float(out.mean.abs().max()))


# ================================================================================================ round 4
def test_gradient_with_four_rows_per_wave_matches_autograd(dev):
    """Batches of >= 4096 rows take the S = 4 instantiation of the row-per-wave kernel (four rows per wave tile, their
    outer products accumulated in registers before the workgroup fold) and, beyond 1024 wave tiles, a second pass that
    ADDS into the workgroup's slab: 4200 train + 300 validation rows, d = 64, against autograd of the CPU restatement."""
    _four_rows_case(dev, 64, 6, 4200, 300)


def test_gradient_at_d512_with_two_rows_per_wave_matches_autograd(dev):
    """d = 512 (four registers per half and lane, the accumulators flushed in two rounds) with a batch large enough for the
    two-rows-per-wave instantiation."""
    _four_rows_case(dev, 512, 7, 4100, 60)


def _four_rows_case(dev, d, H, n, nv):
    from nfmc_amd.flow_training import DeviceFit
    of, f = _flow(d, H, 2, 2, 41)
    g0 = torch.Generator().manual_seed(9)
    x = torch.randn(n, d, generator=g0) * 0.8
    xv = torch.randn(nv, d, generator=g0) * 0.9 + 0.1
    f.to(dev)
    fit = DeviceFit(f.bijection, dev, n + nv, lr=0.0)
    fit.opt.beta1, fit.opt.weight_decay = 0.0, 0.0
    fit.set_validation(xv.to(dev))
    fit.step(x.to(dev), 0)
    loss_gpu, applied, val_gpu = (float(v) for v in fit.status.cpu())
    loss = -of.log_prob(x).mean()
    loss.backward()
    assert applied == 1.0
    np.testing.assert_allclose(loss_gpu, float(loss.detach()), rtol=2e-5)
    np.testing.assert_allclose(val_gpu, float(-of.log_prob(xv).mean().detach()), rtol=2e-5)
    g = copy.deepcopy(f)
    fit.write_back(fit.m, bijection=g.bijection)
    want = dict(of.named_parameters())
    for name, p in g.named_parameters():
        w = want[name].grad
        scale = max(float(w.abs().max()), 1e-3)
        np.testing.assert_allclose(p.detach().cpu().numpy(), w.numpy(), atol=2e-4 * scale, rtol=0, err_msg=name)
    # and twice the same launch: bitwise the same gradient (fixed-order folds everywhere)
    first = fit.m.clone()
    fit.step(x.to(dev), 0)
    assert torch.equal(fit.m, first)


@pytest.mark.parametrize('H', [6, 16])
def test_a_shrinking_batch_with_validation_rows_leaves_no_stale_gradients(dev, H):
    """ADVICE r03 (fit_kernels.hip:99): workgroups that only see validation tiles never write the gradient part of their
    slab, so the fold must not add those slabs.  Two steps on ONE NfmcFlowFit -- a large batch, then a small one, both with
    validation rows -- must give the small batch's gradient exactly as a fresh fitter computes it (both kernel families)."""
    from nfmc_amd.flow_training import DeviceFit
    d = 32
    _of, f = _flow(d, H, 2, 2, 17)
    f.to(dev)
    g0 = torch.Generator().manual_seed(3)
    big = (torch.randn(900, d, generator=g0) * 0.7).to(dev)
    small = (torch.randn(70, d, generator=g0) * 0.7).to(dev)
    xv = (torch.randn(500, d, generator=g0) * 0.7).to(dev)

    def fitter():
        ft_ = DeviceFit(f.bijection, dev, 1400, lr=0.0)
        ft_.opt.beta1, ft_.opt.weight_decay = 0.0, 0.0
        ft_.set_validation(xv)
        return ft_
    a = fitter()
    a.step(big, 0)
    a.step(small, 0)
    b = fitter()
    b.step(small, 0)
    assert torch.equal(a.m, b.m)
    assert float(a.m.abs().max()) > 0
    assert torch.equal(a.status, b.status)


def test_enqueued_run_equals_the_step_by_step_loop(dev):
    """nfmc_flow_fit_epochs_f32 (bookkeeping in the fold kernel, nothing read back) against the same epochs driven one
    nfmc_flow_fit_step_f32 at a time with the bookkeeping of flow_training._loop on the host: same weights after every
    epoch bit for bit, same best validation loss, same best weights, same epoch of the early stop."""
    from nfmc_amd import hip
    from nfmc_amd.flow_training import DeviceFit
    d, n, nv, lr, thr, epochs = 32, 700, 200, 0.05, 4, 60
    _of, fa = _flow(d, 6, 2, 2, 23)
    fa.to(dev)
    fb = copy.deepcopy(fa)
    g0 = torch.Generator().manual_seed(11)
    x = (torch.randn(n, d, generator=g0) * 0.6 + 0.2).to(dev)
    xv = (torch.randn(nv, d, generator=g0) * 2.5 - 1.0).to(dev)     # validation the training rows say little about: it turns
    # --- host loop
    a = DeviceFit(fa.bijection, dev, n + nv, lr=lr)
    a.set_validation(xv)
    best, since, applied, best_vec, stop_at = math.inf, 0, 0, a.params.clone(), None
    for c in range(epochs + 1):
        a.step(x, applied, lr=0.0 if c == epochs else lr)
        loss, ok, val = (float(t) for t in a.status.cpu())
        if c > 0:
            if val < best:
                best, since = val, 0
                best_vec.copy_(a.prev)
            else:
                since += 1
                if since > thr:
                    stop_at = c
                    break
        if c == epochs:
            break
        assert ok == 1.0
        applied += 1
    assert stop_at is not None and 5 < stop_at < epochs        # the scenario does stop early
    # --- enqueued run
    b = DeviceFit(fb.bijection, dev, n + nv, lr=lr)
    b.set_validation(xv)
    ctl = b.control(epochs, True, thr, True)
    b.run_calls(ctl, x, 0, epochs + 1)
    st = b.state_after(epochs + 1)
    assert st[hip.FIT_STOPPED] == 1.0 and st[hip.FIT_DIVERGED] == 0.0
    assert st[hip.FIT_APPLIED] == applied == stop_at
    np.testing.assert_allclose(st[hip.FIT_BEST_LOSS], best, rtol=0, atol=0)
    assert torch.equal(b.best, best_vec)
    assert torch.equal(b.params, a.prev)            # the stopping call's step is discarded on both sides


def test_variational_run_skips_or_ends_on_a_nonfinite_epoch(dev):
    """check_for_divergences of Flow.variational_fit (neutra.py:84-91 passes True, imh.py:67-72 the default False): a batch
    of latents with a NaN gives a non-finite loss; the run skips that epoch (weights, AdamW count and moments untouched) or
    ends as diverged, decided on the device."""
    from nfmc_amd import hip
    from nfmc_amd.flow_training import DeviceFit
    from nfmc_amd.potentials import SumOfSquares
    d, n = 16, 128
    _of, f = _flow(d, 4, 2, 2, 5)
    f.to(dev)
    pot = SumOfSquares((d,)).descriptor(dev)
    g0 = torch.Generator().manual_seed(4)
    z = [(torch.randn(n, d, generator=g0)).to(dev) for _ in range(4)]
    z[2][5, 3] = float('nan')
    for skip in (True, False):
        fit = DeviceFit(f.bijection, dev, n, lr=0.02)
        ctl = fit.control(4, False, 50, True, skip_nonfinite=skip)
        w = []
        for c in range(4):
            fit.run_calls(ctl, z[c], c, 1, pot_struct=pot)
            w.append(fit.params.clone())
        st = fit.state_after(4)
        assert not torch.equal(w[0], w[1])
        assert torch.equal(w[1], w[2])                                    # the NaN epoch moved nothing
        if skip:
            assert st[hip.FIT_DIVERGED] == 0.0 and st[hip.FIT_APPLIED] == 3.0 and not torch.equal(w[2], w[3])
        else:
            assert st[hip.FIT_DIVERGED] == 1.0 and st[hip.FIT_APPLIED] == 2.0 and torch.equal(w[2], w[3])
        assert math.isfinite(st[hip.FIT_BEST_LOSS])


def test_resident_fitter_is_reused_and_follows_outside_changes(dev):
    """`DeviceFit.of` keeps one fitter per flow across fits: a second fit reuses it without gathering the parameters (the
    vector is what the flow's kernels use); parameters changed from outside (load_state_dict) are gathered again; the flow's
    kernels see every fit (log_prob after fit = log_prob of the parameters)."""
    from nfmc_amd import flow_training as ft
    from oracle import flow as oflow
    d = 24
    _of, f = _flow(d, 5, 2, 2, 8)
    f.to(dev)
    g0 = torch.Generator().manual_seed(2)
    x = (torch.randn(600, d, generator=g0) * 0.5 + 0.4).to(dev)
    gathers = []
    orig = ft.DeviceFit._scatter

    def spy(self, vec, to_vector, bijection=None):
        gathers.append(bool(to_vector))
        return orig(self, vec, to_vector, bijection)
    ft.DeviceFit._scatter = spy
    try:
        f.fit(x, n_epochs=5, lr=0.02, show_progress=False)
        first = f.bijection._device_fit
        assert gathers == [True, False]
        f.fit(x, n_epochs=5, lr=0.02, show_progress=False)
        assert f.bijection._device_fit is first and gathers == [True, False, False]      # no second gather
        # the kernels' view (pack cache = the trained vector) agrees with the nn.Parameters' view (CPU restatement)
        of2 = oflow.Flow(oflow.RealNVP((d,), conditioner_kwargs={'n_hidden': 5, 'n_layers': 2}))
        of2.load_state_dict({k: v.cpu() for k, v in f.state_dict().items()})
        np.testing.assert_allclose(f.log_prob(x[:50]).cpu().numpy(), of2.log_prob(x[:50].cpu()).detach().numpy(), atol=2e-4)
        sd = {k: v.clone() * 0.5 for k, v in f.state_dict().items()}
        f.load_state_dict(sd)
        f.fit(x, n_epochs=1, lr=0.0, keep_best_weights=False, show_progress=False)
        assert gathers == [True, False, False, True, False]                               # changed outside: gathered again
    finally:
        ft.DeviceFit._scatter = orig
    assert copy.deepcopy(f).bijection.__dict__.get('_device_fit') is None


def test_refit_split_on_the_device_matches_the_oracle_rows(dev):
    """`train_val_split` (tuning.py:44-65) on a GPU buffer: ONE launch gathers rows pi(0 ..) of the keyed permutation; the
    rows against oracle/shuffle.py bit for bit (same seed from torch's CPU generator), sizes by the reference's rule (cut at
    train_pct, caps), train and validation disjoint."""
    from nfmc_amd.tuning import train_val_split
    from oracle import shuffle
    K, n, d = 5, 333, 7
    x = torch.arange(K * n * d, dtype=torch.float32).reshape(K, n, d)
    for caps in ((4096, 4096), (100, 50), (2000, 10)):
        torch.manual_seed(99)
        seed = int(torch.randint(0, 2 ** 62, ()).item())
        torch.manual_seed(99)
        xt, xv = train_val_split(x.to(dev), 0.7, caps[0], caps[1])
        wt, wv = shuffle.train_val_split(x.numpy(), 0.7, caps[0], caps[1], seed)
        cut = int(0.7 * K * n)
        assert xt.shape[0] == min(cut, caps[0]) and xv.shape[0] == min(K * n - cut, caps[1])
        assert np.array_equal(xt.cpu().numpy(), wt) and np.array_equal(xv.cpu().numpy(), wv)
        rows = torch.cat([xt, xv]).cpu()[:, 0]
        assert rows.unique().numel() == rows.numel()
    # 2-D events keep their shape
    xt, xv = train_val_split(torch.randn(3, 40, 4, 2).to(dev), 0.5, 4096, 4096)
    assert xt.shape == (60, 4, 2) and xv.shape == (60, 4, 2)


WIDE_CASES = [  # d, n_hidden, hidden layers, coupling layers, rows, potential / 'ml', NICE
    (128, 128, 2, 2, 300, 'funnel', False), (128, 128, 2, 2, 200, 'ml', False), (64, 40, 1, 3, 130, 'sum', False),
    (64, 64, 2, 2, 129, 'ml', False), (128, 100, 1, 2, 128, 'diag', False), (64, 128, 2, 1, 77, 'ml', True),
]


@pytest.mark.parametrize('d,H,nhl,nl,n,kind,nice', WIDE_CASES)
def test_wide_conditioner_gradients_on_the_matrix_cores_match_autograd(dev, d, H, nhl, nl, n, kind, nice):
    """Conditioners of width 33..128 at d = 64 / 128 (C4's flow: 128 x 2 at d = 128) are fitted on the device too
    (csrc/fit_mfma.hip: weight gradients as batch-axis GEMMs on the matrix cores).  One step with lr = 0, beta1 = 0 leaves the
    first moment equal to the gradient: every entry of every parameter against autograd of the CPU restatement (oracle/flow.py),
    for the reverse-KL loss (neutra.py:84-91, imh.py:67-72) and the maximum-likelihood loss (jump.py:139-151); the two
    orientations of every matrix in the blob carry the same gradient bit for bit."""
    from nfmc_amd.flow_training import DeviceFit
    from nfmc_amd.potentials import DiagonalGaussian, Funnel, SumOfSquares
    of, f = _flow(d, H, nhl, nl, 70 + d + H, nice)
    f.to(dev)
    assert DeviceFit.supported(f.bijection, dev)
    g0 = torch.Generator().manual_seed(300 + d)
    rows = torch.randn(n, d, generator=g0) * (0.8 if kind == 'ml' else 1.0)
    fit = DeviceFit(f.bijection, dev, n, lr=0.0)
    assert fit.wide
    fit.opt.beta1, fit.opt.weight_decay = 0.0, 0.0
    before = fit.params.clone()
    if kind == 'ml':
        fit.step(rows.to(dev), 0)
        loss = -of.log_prob(rows).mean()
    else:
        pot = {'sum': SumOfSquares((d,)), 'funnel': Funnel((d,), 3.0),
               'diag': DiagonalGaussian((d,), torch.linspace(-0.5, 0.5, d), torch.linspace(0.6, 1.7, d))}[kind]
        fit.step_variational(rows.to(dev), pot.descriptor(dev), 0)
        x, ld = of.bijection.inverse(rows)
        loss = (of.base_log_prob(rows) - ld + pot(x)).mean()
    loss.backward()
    loss_gpu, applied, _val = (float(v) for v in fit.status.cpu())
    assert applied == 1.0 and torch.equal(fit.params, before)
    np.testing.assert_allclose(loss_gpu, float(loss.detach()), rtol=3e-5, atol=3e-5)
    gflow = copy.deepcopy(f)
    fit.write_back(fit.m, bijection=gflow.bijection)
    want = dict(of.named_parameters())
    for name, p in gflow.named_parameters():
        w = want[name].grad
        scale = max(float(w.abs().max()), 1e-3)
        np.testing.assert_allclose(p.detach().cpu().numpy(), w.numpy(), atol=2e-4 * scale, rtol=0, err_msg=name)
    # both orientations of every matrix: identical gradients (so AdamW keeps the copies equal)
    for p_, off, r, c, rs, cs in fit._layout(f.bijection):
        if r > 1 and rs != 1:
            a = fit.m[off:off + r * rs].view(r, rs)[:, :c]
            twin = [q for q in fit._layout(f.bijection) if q[0] is p_ and q[4] == 1]
            (_p2, off2, _r2, _c2, _rs2, cs2), = twin
            b = fit.m[off2:off2 + c * cs2].view(c, cs2)[:, :r].t()
            assert torch.equal(a, b)


def test_wide_conditioner_variational_fit_runs_on_the_device_and_learns(dev, monkeypatch):
    """`Flow.variational_fit` of C4's flow (d = 128, conditioner 128 x 2) to the funnel: the run goes through
    nfmc_flow_fit_epochs_f32 (spied), the reverse-KL estimate falls, the flow's kernels see the trained weights (the weight
    blob is the trainable vector), and the eager torch loop (NFMC_FIT_TORCH=1) reaches a comparable loss."""
    from nfmc_amd import flow_training as ft
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.potentials import Funnel
    d = 128
    pot = Funnel((d,), 3.0)
    res = {}
    for torch_path in ('0', '1'):
        monkeypatch.setenv('NFMC_FIT_TORCH', torch_path)
        calls = []
        orig = ft.DeviceFit.run_calls
        monkeypatch.setattr(ft.DeviceFit, 'run_calls',
                            lambda self, ctl, z, c0, k, _o=orig, **kw: (calls.append(k), _o(self, ctl, z, c0, k, **kw))[1])
        torch.manual_seed(1)
        f = Flow(RealNVP((d,), conditioner_kwargs={'n_hidden': 128, 'n_layers': 2})).to(dev)

        def rkl():
            z = torch.randn(4096, d, generator=torch.Generator().manual_seed(5)).to(dev)
            x, ld = f.bijection.inverse(z)
            return float((-0.5 * (z * z).sum(1) - ld + pot(x)).mean())
        l0 = rkl()
        torch.manual_seed(2)
        f.variational_fit(lambda v: -pot(v), n_epochs=60, lr=0.01, n_samples=1024, early_stopping=False, keep_best_weights=True,
                          show_progress=False, potential=pot)
        res[torch_path] = (l0, rkl(), len(calls))
        monkeypatch.setattr(ft.DeviceFit, 'run_calls', orig)
    (l0, la, ca), (_l0, lb, cb) = res['0'], res['1']
    assert ca == 60 and cb == 0
    assert la < l0 - 5.0 and lb < l0 - 5.0, res
    assert abs(la - lb) < 0.25 * abs(l0 - lb), res
