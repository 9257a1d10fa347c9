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
