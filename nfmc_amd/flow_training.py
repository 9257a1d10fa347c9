"""Flow training used by the warmup paths next to the hot path:
`Flow.fit` (maximum likelihood on MCMC samples: jump.py:139-151, imh.py:171-175) and
`Flow.variational_fit` (reverse-KL to the target: imh.py:67-72, neutra.py:84-91), with the keyword
surface nfmc passes (`early_stopping`, `early_stopping_threshold`, `keep_best_weights`, `batch_size`,
`n_epochs`, `lr`, `n_samples`, `check_for_divergences`, `show_progress`, `time_limit_seconds`) and the
`ValueError`-on-divergence contract the callers catch (jump.py:150, imh.py:174).

Training needs gradients with respect to the WEIGHTS, which the sampling kernels never compute, so this
module evaluates the same RealNVP spec with differentiable torch ops (on the GPU when there is one).
It is not on the sampling path: `sample()` never calls into it unless `warmup=True` / `fit_nf=True`.
"""
import math
import time
from copy import deepcopy

import torch

from .flows import MIN_SCALE, AffineCoupling, ElementwiseAffine, ReversePermutation, RQSCoupling

RQS_MIN_BIN, RQS_MIN_DERIV = 1e-3, 1e-3


def _coupling_params(layer: AffineCoupling, x_a, min_scale=MIN_SCALE):
    h = x_a
    for lin in layer.conditioner[:-1]:
        h = torch.tanh(lin(h))
    h = layer.conditioner[-1](h)
    if min_scale >= 1.0:    # additive coupling (NICE)
        alpha = torch.ones_like(h[:, :layer.d_b])
    else:
        alpha = torch.exp(h[:, :layer.d_b] / 2 + math.log(1 - min_scale)) + min_scale
    return alpha, h[:, layer.d_b:] / 2


def _rqs(layer: RQSCoupling, x_a, v, bound, inverse):
    """Differentiable rational-quadratic spline coupling of the target half v ('c-rqnsf'):
    returns (transformed v, summed logdet of THIS direction)."""
    K, B = layer.n_bins, float(bound)
    h = x_a
    for lin in layer.conditioner[:-1]:
        h = torch.tanh(lin(h))
    raw = layer.conditioner[-1](h).reshape(v.shape[0], layer.d_b, 3 * K - 1)
    w = RQS_MIN_BIN + (1 - K * RQS_MIN_BIN) * torch.softmax(raw[..., :K], dim=-1)
    hh = RQS_MIN_BIN + (1 - K * RQS_MIN_BIN) * torch.softmax(raw[..., K:2 * K], dim=-1)
    zero, one = torch.zeros_like(w[..., :1]), torch.ones_like(w[..., :1])
    edge = torch.full_like(zero, B)
    cw = torch.cat([(torch.cat([zero, torch.cumsum(w, -1)], -1) * (2 * B) - B)[..., :-1], edge], -1)
    ch = torch.cat([(torch.cat([zero, torch.cumsum(hh, -1)], -1) * (2 * B) - B)[..., :-1], edge], -1)
    dv = torch.cat([one, RQS_MIN_DERIV + torch.nn.functional.softplus(raw[..., 2 * K:]), one], -1)
    inside = (v >= -B) & (v <= B)
    vc = v.clamp(-B, B)
    k = (vc[..., None] >= (ch if inverse else cw)[..., 1:-1]).sum(-1, keepdim=True)
    take = lambda a, off=0: torch.gather(a, -1, k + off)[..., 0]
    x0, x1, y0, y1, d0, d1 = take(cw), take(cw, 1), take(ch), take(ch, 1), take(dv), take(dv, 1)
    bw, bh = x1 - x0, y1 - y0
    s = bh / bw
    dd = d0 + d1 - 2 * s
    if inverse:
        dy = vc - y0
        a, b, c = dy * dd + bh * (s - d0), bh * d0 - dy * dd, -s * dy
        th = 2 * c / (-b - torch.sqrt(b * b - 4 * a * c))
        out = th * bw + x0
    else:
        th = (vc - x0) / bw
        out = y0 + bh * (s * th * th + d0 * th * (1 - th)) / (s + dd * th * (1 - th))
    ld = torch.log(s * s * (d1 * th * th + 2 * s * th * (1 - th) + d0 * (1 - th) ** 2)) - 2 * torch.log(s + dd * th * (1 - th))
    ld = torch.where(inside, -ld if inverse else ld, torch.zeros_like(ld))
    return torch.where(inside, out, v), ld.sum(-1)


def forward_torch(bijection, x):
    """Differentiable x -> (z, logdet) of the RealNVP spec."""
    n = x.shape[0]
    h = x.reshape(n, -1)
    ld = torch.zeros(n, dtype=h.dtype, device=h.device)
    for layer in bijection.layers:
        if isinstance(layer, ElementwiseAffine):
            h = torch.exp(layer.log_scale) * h + layer.shift
            ld = ld + layer.log_scale.sum()
        elif isinstance(layer, ReversePermutation):
            h = h.flip(-1)
        elif isinstance(layer, RQSCoupling):
            out, l = _rqs(layer, h[:, :layer.d_a], h[:, layer.d_a:], bijection.spline_bound, False)
            h = torch.cat([h[:, :layer.d_a], out], dim=1)
            ld = ld + l
        else:
            alpha, beta = _coupling_params(layer, h[:, :layer.d_a], bijection.min_scale)
            h = torch.cat([h[:, :layer.d_a], alpha * h[:, layer.d_a:] + beta], dim=1)
            ld = ld + torch.log(alpha).sum(-1)
    return h, ld


def inverse_torch(bijection, z):
    n = z.shape[0]
    h = z.reshape(n, -1)
    ld = torch.zeros(n, dtype=h.dtype, device=h.device)
    for layer in reversed(list(bijection.layers)):
        if isinstance(layer, ElementwiseAffine):
            h = (h - layer.shift) * torch.exp(-layer.log_scale)
            ld = ld - layer.log_scale.sum()
        elif isinstance(layer, ReversePermutation):
            h = h.flip(-1)
        elif isinstance(layer, RQSCoupling):
            out, l = _rqs(layer, h[:, :layer.d_a], h[:, layer.d_a:], bijection.spline_bound, True)
            h = torch.cat([h[:, :layer.d_a], out], dim=1)
            ld = ld + l
        else:
            alpha, beta = _coupling_params(layer, h[:, :layer.d_a], bijection.min_scale)
            h = torch.cat([h[:, :layer.d_a], (h[:, layer.d_a:] - beta) / alpha], dim=1)
            ld = ld - torch.log(alpha).sum(-1)
    return h, ld


def _base_log_prob(z):
    return -0.5 * (z * z).sum(-1) - 0.5 * z.shape[1] * math.log(2 * math.pi)


def _train_device(flow):
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else flow.get_device()


def _loop(flow, loss_fn, val_fn, n_epochs, lr, early_stopping, early_stopping_threshold, keep_best_weights,
          show_progress, time_limit_seconds, check_for_divergences=True):
    opt = torch.optim.AdamW(flow.parameters(), lr=lr)
    # the weights the flow came in with are the fallback "best": a fit whose every epoch is non-finite leaves them as
    # they were instead of whatever the optimiser last wrote
    best_loss, best_state, since_best = math.inf, (deepcopy(flow.state_dict()) if keep_best_weights else None), 0
    t0 = time.time()
    for epoch in range(int(n_epochs)):
        if time_limit_seconds is not None and time.time() - t0 >= time_limit_seconds:
            break
        opt.zero_grad()
        loss = loss_fn()
        if not torch.isfinite(loss):
            if check_for_divergences:
                raise ValueError('flow training diverged (non-finite loss)')
            continue   # no backward / step on a non-finite loss: it would write NaN into every weight
        loss.backward()
        opt.step()
        with torch.no_grad():
            v = float(val_fn()) if val_fn is not None else float(loss)
        if not math.isfinite(v):
            if check_for_divergences:
                raise ValueError('flow training diverged (non-finite validation loss)')
            continue
        if v < best_loss:
            best_loss, since_best = v, 0
            if keep_best_weights:
                best_state = deepcopy(flow.state_dict())
        else:
            since_best += 1
            if early_stopping and since_best > early_stopping_threshold:
                break
    if keep_best_weights and best_state is not None:
        flow.load_state_dict(best_state)
    return best_loss


def fit(flow, x_train, x_val=None, n_epochs: int = 500, lr: float = 0.05, batch_size='adaptive',
        shuffle: bool = True, show_progress: bool = False, keep_best_weights: bool = True,
        early_stopping: bool = False, early_stopping_threshold: int = 50, time_limit_seconds=None, **_ignored):
    """Maximum-likelihood fit: minimise -mean log q(x_train)."""
    dev = _train_device(flow)
    flow.to(dev)
    xt = x_train.detach().to(dev, torch.float32).reshape(x_train.shape[0], -1)
    xv = x_val.detach().to(dev, torch.float32).reshape(x_val.shape[0], -1) if x_val is not None and len(x_val) else None
    n = xt.shape[0]
    if n == 0:
        return
    bs = n if batch_size == 'adaptive' or batch_size is None else max(1, min(int(batch_size), n))
    gen = torch.Generator(device='cpu').manual_seed(int(torch.randint(0, 2 ** 31, ()).item()))

    def nll(x):
        z, ld = forward_torch(flow.bijection, x)
        return -(_base_log_prob(z) + ld).mean()

    def loss_fn():
        if bs >= n:
            return nll(xt)
        idx = torch.randperm(n, generator=gen)[:bs].to(dev) if shuffle else torch.arange(bs, device=dev)
        return nll(xt[idx])

    val_fn = (lambda: nll(xv)) if xv is not None else None
    _loop(flow, loss_fn, val_fn, n_epochs, lr, early_stopping, early_stopping_threshold, keep_best_weights,
          show_progress, time_limit_seconds)


def variational_fit(flow, log_prob_fn, n_epochs: int = 500, lr: float = 0.05, n_samples: int = 1000,
                    early_stopping: bool = False, early_stopping_threshold: int = 50, keep_best_weights: bool = True,
                    check_for_divergences: bool = False, show_progress: bool = False, time_limit_seconds=None,
                    **_ignored):
    """Reverse-KL fit: minimise E_{z~N(0,I)} [ log q(x) - log p(x) ], x = f^-1(z)."""
    dev = _train_device(flow)
    flow.to(dev)
    d = flow.bijection.d
    event_shape = flow.event_shape
    n_samples = max(int(n_samples), 1)

    def loss_fn():
        z = torch.randn(n_samples, d, device=dev)
        x, ld = inverse_torch(flow.bijection, z)
        log_q = _base_log_prob(z) - ld
        log_p = log_prob_fn(x.reshape(n_samples, *event_shape)).reshape(-1)
        return (log_q - log_p.to(log_q)).mean()

    _loop(flow, loss_fn, None, n_epochs, lr, early_stopping, early_stopping_threshold, keep_best_weights,
          show_progress, time_limit_seconds, check_for_divergences=check_for_divergences)
