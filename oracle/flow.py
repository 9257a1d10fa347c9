"""CPU restatement of the flow half of the path (RealNVP + Flow).  TEST INFRASTRUCTURE.

PARITY UNPINNED for the internals: the reference delegates RealNVP to the third-party
`torchflows` package (imported at nfmc/algorithms/sampling/base.py:6, nfmc/util.py:228-281;
declared without a version in pyproject.toml:23 / setup.py:55 / environment.yml:5), whose
source is not under /root/reference and cannot be installed here.  What IS fixed by the
reference's call sites is the API and the sign/direction conventions, and this file follows
them:

  * `Flow.sample(n, return_log_prob=True)` -> (x, log q(x)), an *inverse* pass from
    z ~ N(0, I)                                   (jump.py:205, imh.py:221)
  * `Flow.log_prob(x)` = N(forward(x); 0, I) + logdet_forward          (jump.py:218, imh.py:214)
  * `bijection.inverse(z)` -> (x, logdet_inverse), added to log p(x) by NeuTra (neutra.py:60-63)
  * `bijection.layers` grows with `n_layers`       (test/test_flow_kwargs.py:18-30)
  * `RealNVP(event_shape, n_layers=..., conditioner_kwargs={'n_layers':..,'n_hidden':..})`
                                                   (test/test_flow_kwargs.py:49)

The build's RealNVP spec (DESIGN.md "RealNVP spec"), forward direction x -> z:

  layers = [ElementwiseAffine] + n_layers * [ReversePermutation, AffineCoupling] + [ElementwiseAffine]
  ElementwiseAffine : z = exp(log_scale) * x + shift ; logdet = sum(log_scale)
  ReversePermutation: z[j] = x[d-1-j]                ; logdet = 0
  AffineCoupling    : d_a = d // 2 (source, unchanged), d_b = d - d_a (target)
                      h = MLP(x[:d_a]) in R^{2 d_b}; u_alpha = h[:d_b], u_beta = h[d_b:]
                      alpha = exp(u_alpha / 2 + log(1 - m)) + m, m = 1e-3 ; beta = u_beta / 2
                      z[d_a:] = alpha * x[d_a:] + beta ; logdet = sum(log alpha)
  MLP               : Linear(d_a, H) tanh [Linear(H, H) tanh]*(n_hl - 1) Linear(H, 2 d_b)
                      defaults H = max(4, int(3 log10 d_a)), n_hl = 2
Events with more than one axis are flattened row-major.
"""
import math
from typing import Sequence

import torch
import torch.nn as nn

MIN_SCALE = 1e-3


def default_hidden(d_a: int) -> int:
    return max(4, int(3 * math.log10(max(d_a, 1))))


class ElementwiseAffine(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.log_scale = nn.Parameter(torch.zeros(d))
        self.shift = nn.Parameter(torch.zeros(d))

    def forward(self, x):
        return torch.exp(self.log_scale) * x + self.shift, self.log_scale.sum().expand(x.shape[0])

    def inverse(self, z):
        return (z - self.shift) * torch.exp(-self.log_scale), (-self.log_scale.sum()).expand(z.shape[0])


class ReversePermutation(nn.Module):
    def forward(self, x):
        return x.flip(-1), torch.zeros(x.shape[0], dtype=x.dtype)

    inverse = forward


class AffineCoupling(nn.Module):
    additive = False   # NICE: scale fixed to 1, the scale half of the conditioner output is inert

    def __init__(self, d, n_hidden=None, n_layers=2):
        super().__init__()
        self.d_a = d // 2
        self.d_b = d - self.d_a
        h = default_hidden(self.d_a) if n_hidden is None else int(n_hidden)
        dims = [self.d_a] + [h] * int(n_layers) + [2 * self.d_b]
        self.conditioner = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])

    def _params(self, x_a):
        h = x_a
        for lin in self.conditioner[:-1]:
            h = torch.tanh(lin(h))
        h = self.conditioner[-1](h)
        u_alpha, u_beta = h[:, :self.d_b], h[:, self.d_b:]
        if self.additive:
            alpha = torch.ones_like(u_alpha)
        else:
            alpha = torch.exp(u_alpha / 2 + math.log(1 - MIN_SCALE)) + MIN_SCALE
        return alpha, u_beta / 2

    def forward(self, x):
        alpha, beta = self._params(x[:, :self.d_a])
        z = torch.cat([x[:, :self.d_a], alpha * x[:, self.d_a:] + beta], dim=1)
        return z, torch.log(alpha).sum(-1)

    def inverse(self, z):
        alpha, beta = self._params(z[:, :self.d_a])
        x = torch.cat([z[:, :self.d_a], (z[:, self.d_a:] - beta) / alpha], dim=1)
        return x, -torch.log(alpha).sum(-1)


class RealNVP(nn.Module):
    def __init__(self, event_shape, n_layers: int = 2, conditioner_kwargs: dict = None, **kwargs):
        super().__init__()
        if isinstance(event_shape, int):
            event_shape = (event_shape,)
        self.event_shape = tuple(event_shape)
        d = int(math.prod(self.event_shape))
        ck = dict(conditioner_kwargs or {})
        ck.update({k: v for k, v in kwargs.items() if k in ('n_hidden',)})
        layers = [ElementwiseAffine(d)]
        for _ in range(n_layers):
            layers += [ReversePermutation(), AffineCoupling(d, ck.get('n_hidden'), ck.get('n_layers', 2))]
        layers.append(ElementwiseAffine(d))
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        b = x.shape[0]
        h = x.reshape(b, -1)
        logdet = torch.zeros(b, dtype=x.dtype)
        for layer in self.layers:
            h, ld = layer.forward(h)
            logdet = logdet + ld
        return h.reshape(x.shape), logdet

    def inverse(self, z):
        b = z.shape[0]
        h = z.reshape(b, -1)
        logdet = torch.zeros(b, dtype=z.dtype)
        for layer in reversed(self.layers):
            h, ld = layer.inverse(h)
            logdet = logdet + ld
        return h.reshape(z.shape), logdet


# ---- rational-quadratic spline coupling ('c-rqnsf', nfmc/util.py:17; Durkan et al. 2019).  Build-defined like the
# affine coupling above (torchflows absent): K bins on [-B, B], identity outside, linear tails (boundary
# derivatives 1).  Per target coordinate t the conditioner emits P = 3K - 1 numbers at indices t*P + (0..P):
#   K unnormalised widths | K unnormalised heights | K - 1 unnormalised interior derivatives
#   widths  w = MIN_BIN + (1 - K MIN_BIN) softmax(.), knots cw_0 = -B, cw_{k+1} = cw_k + 2B w_k (cw_K = B)
#   heights likewise (ch);  derivatives d_0 = d_K = 1, d_k = MIN_DERIV + softplus(.)
# forward  x in bin k: th = (x - cw_k) / (2B w_k), s = h_k / w_k (h, w = bin height / width),
#   y = ch_k + h (s th^2 + d_k th (1 - th)) / (s + (d_k + d_{k+1} - 2 s) th (1 - th))
#   logdet = log(s^2 (d_{k+1} th^2 + 2 s th (1 - th) + d_k (1 - th)^2)) - 2 log(s + (d_k + d_{k+1} - 2 s) th (1 - th))
RQS_BINS, RQS_BOUND, RQS_MIN_BIN, RQS_MIN_DERIV = 8, 5.0, 1e-3, 1e-3


def rqs_params(raw, K=RQS_BINS, B=RQS_BOUND):
    """raw (..., 3K-1) -> knot positions cw, ch (..., K+1) and derivatives d (..., K+1)."""
    uw, uh, ud = raw[..., :K], raw[..., K:2 * K], raw[..., 2 * K:]
    w = RQS_MIN_BIN + (1 - K * RQS_MIN_BIN) * torch.softmax(uw, dim=-1)
    h = RQS_MIN_BIN + (1 - K * RQS_MIN_BIN) * torch.softmax(uh, dim=-1)
    zero = torch.zeros_like(w[..., :1])
    cw = torch.cat([zero, torch.cumsum(w, dim=-1)], dim=-1) * (2 * B) - B
    ch = torch.cat([zero, torch.cumsum(h, dim=-1)], dim=-1) * (2 * B) - B
    cw = torch.cat([cw[..., :-1], torch.full_like(zero, B)], dim=-1)
    ch = torch.cat([ch[..., :-1], torch.full_like(zero, B)], dim=-1)
    one = torch.ones_like(zero)
    d = torch.cat([one, RQS_MIN_DERIV + torch.nn.functional.softplus(ud), one], dim=-1)
    return cw, ch, d


def rqs_apply(v, raw, inverse, K=RQS_BINS, B=RQS_BOUND):
    """elementwise spline of v (...,) with parameters raw (..., 3K-1); returns (out, logdet of THIS direction)."""
    cw, ch, d = rqs_params(raw, K, B)
    inside = (v >= -B) & (v <= B)
    vc = v.clamp(-B, B)
    knots = ch if inverse else cw
    k = (vc[..., None] >= knots[..., 1:-1]).sum(-1, keepdim=True)          # bin index 0..K-1
    take = lambda a, off=0: torch.gather(a, -1, k + off)[..., 0]
    x0, x1, y0, y1 = take(cw), take(cw, 1), take(ch), take(ch, 1)
    d0, d1 = take(d), take(d, 1)
    bw, bh = x1 - x0, y1 - y0
    s = bh / bw
    if inverse:
        dy = vc - y0
        a = dy * (d0 + d1 - 2 * s) + bh * (s - d0)
        b = bh * d0 - dy * (d0 + d1 - 2 * s)
        c = -s * dy
        th = 2 * c / (-b - torch.sqrt(b * b - 4 * a * c))
        out = th * bw + x0
    else:
        th = (vc - x0) / bw
        out = y0 + bh * (s * th * th + d0 * th * (1 - th)) / (s + (d0 + d1 - 2 * s) * th * (1 - th))
    den = s + (d0 + d1 - 2 * s) * th * (1 - th)
    ld = torch.log(s * s * (d1 * th * th + 2 * s * th * (1 - th) + d0 * (1 - th) ** 2)) - 2 * torch.log(den)
    if inverse:
        ld = -ld
    return torch.where(inside, out, v), torch.where(inside, ld, torch.zeros_like(ld))


class RQSCoupling(nn.Module):
    def __init__(self, d, n_hidden=None, n_layers=2, n_bins=RQS_BINS):
        super().__init__()
        self.d_a = d // 2
        self.d_b = d - self.d_a
        self.n_bins = int(n_bins)
        h = default_hidden(self.d_a) if n_hidden is None else int(n_hidden)
        dims = [self.d_a] + [h] * int(n_layers) + [(3 * self.n_bins - 1) * self.d_b]
        self.conditioner = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])

    def _raw(self, x_a):
        h = x_a
        for lin in self.conditioner[:-1]:
            h = torch.tanh(lin(h))
        return self.conditioner[-1](h).reshape(x_a.shape[0], self.d_b, 3 * self.n_bins - 1)

    def forward(self, x):
        out, ld = rqs_apply(x[:, self.d_a:], self._raw(x[:, :self.d_a]), False, self.n_bins)
        return torch.cat([x[:, :self.d_a], out], dim=1), ld.sum(-1)

    def inverse(self, z):
        out, ld = rqs_apply(z[:, self.d_a:], self._raw(z[:, :self.d_a]), True, self.n_bins)
        return torch.cat([z[:, :self.d_a], out], dim=1), ld.sum(-1)


class CRQNSF(RealNVP):
    """[ElementwiseAffine] + n_layers x [ReversePermutation, RQSCoupling] + [ElementwiseAffine]."""

    def __init__(self, event_shape, n_layers: int = 2, conditioner_kwargs: dict = None, n_bins: int = RQS_BINS, **kwargs):
        super().__init__(event_shape, n_layers=n_layers, conditioner_kwargs=conditioner_kwargs, **kwargs)
        d = int(math.prod(self.event_shape))
        ck = dict(conditioner_kwargs or {})
        ck.update({k: v for k, v in kwargs.items() if k in ('n_hidden',)})
        layers = list(self.layers)
        for i, m in enumerate(layers):
            if isinstance(m, AffineCoupling):
                layers[i] = RQSCoupling(d, ck.get('n_hidden'), ck.get('n_layers', 2), n_bins)
        self.layers = nn.ModuleList(layers)


class NICE(RealNVP):
    """Additive couplings (nfmc/util.py:13 'nice'): the same stack, every coupling scale = 1, logdet of the
    couplings = 0.  Build-defined like RealNVP above (torchflows absent)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for m in self.layers:
            if isinstance(m, AffineCoupling):
                m.additive = True


class Flow(nn.Module):
    """Duck-typed stand-in for torchflows.Flow (surface listed in SURVEY.md section 8b)."""

    def __init__(self, bijection: RealNVP):
        super().__init__()
        self.bijection = bijection
        self._latent_source = None  # optional callable n -> z, lets tests replay recorded latents

    @property
    def event_shape(self):
        return self.bijection.event_shape

    def get_device(self):
        return next(self.parameters()).device

    def base_log_prob(self, z):
        zf = z.reshape(z.shape[0], -1)
        return -0.5 * (zf * zf).sum(-1) - 0.5 * zf.shape[1] * math.log(2 * math.pi)

    def log_prob(self, x):
        z, logdet = self.bijection.forward(x)
        return self.base_log_prob(z) + logdet

    def fit(self, x_train, x_val=None, n_epochs: int = 1, lr: float = 0.05, **_ignored):
        return fit_(self, x_train, n_epochs=n_epochs, lr=lr)

    def sample(self, n, return_log_prob=False, no_grad=False):
        if self._latent_source is not None:
            z = self._latent_source(n)
        else:
            z = torch.randn(n, *self.event_shape)
        ctx = torch.no_grad() if no_grad else torch.enable_grad()
        with ctx:
            x, logdet_inv = self.bijection.inverse(z)
            if return_log_prob:
                return x, self.base_log_prob(z) - logdet_inv
        return x


def fit_(flow: Flow, x_train, n_epochs: int = 1, lr: float = 0.05):
    """Maximum-likelihood refit used by the adaptive samplers (imh.py:171-175 calls `flow.fit(x_train, n_epochs=1)`):
    `n_epochs` full-batch AdamW steps on -mean log q(x_train); ValueError on a non-finite loss, which is the
    only error the callers catch.  torchflows' own `fit` is not in /root/reference ("parity unpinned" for its
    optimiser details); this is the build's spec, mirrored by nfmc_amd/flow_training.py."""
    opt = torch.optim.AdamW(flow.parameters(), lr=lr)
    x = x_train.detach().reshape(x_train.shape[0], -1).float()
    for _ in range(int(n_epochs)):
        opt.zero_grad()
        loss = -flow.log_prob(x.reshape(x.shape[0], *flow.event_shape)).mean()
        if not torch.isfinite(loss):
            raise ValueError('flow training diverged (non-finite loss)')
        loss.backward()
        opt.step()
    return flow


def perturb_(flow: Flow, seed: int, scale: float = 0.3, target_std: float = None):
    """Deterministic non-trivial weights for tests/benches (all parameters touched)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in flow.parameters():
            p.add_(scale * (torch.rand(p.shape, generator=g) - 0.5) * (2.0 if p.dim() == 1 else 1.0))
        if target_std is not None:
            # forward maps x ~ N(0, std^2) towards N(0, 1)
            flow.bijection.layers[0].log_scale.add_(-math.log(target_std))
    return flow
