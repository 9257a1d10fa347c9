"""CPU restatement of the nfmc-owned half of the path.  TEST INFRASTRUCTURE (oracle/__init__.py).

Every function cites the reference lines it follows (paths relative to /root/reference).
Eager fp32 PyTorch-CPU ops in the reference's op order, autograd for grad U exactly like
`nfmc/algorithms/sampling/mcmc/langevin.py:66-68` / `hmc.py:40-48`.  Pinned against the
reference by tests/golden/*.npz (see tests/golden/make_golden.py).

Noise comes from a `NoiseSource` so one restatement serves three uses:
  TorchNoise   torch.randn / torch.rand in the reference's draw order (bitwise the same
               stream as the reference under the same torch.manual_seed)
  ReplayNoise  recorded tensors (fixtures; also what the HIP kernels' replay mode consumes)
  PhiloxNoise  the build's native counter-based stream (oracle/philox.py)
"""
import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np
import torch

from . import philox


# --------------------------------------------------------------------------- noise sources
class TorchNoise:
    def normal(self, n, shape, step, tag):
        return torch.randn(n, *shape)

    def uniform(self, n, step, tag):
        return torch.rand(n)


class ReplayNoise:
    """Consumes recorded draws in order.  `normals`: list of (n,*event) tensors, `uniforms`: list of (n,)."""

    def __init__(self, normals, uniforms):
        self.normals = list(normals)
        self.uniforms = list(uniforms)
        self.i_n = 0
        self.i_u = 0

    def normal(self, n, shape, step, tag):
        v = self.normals[self.i_n]
        self.i_n += 1
        return v.clone()

    def uniform(self, n, step, tag):
        v = self.uniforms[self.i_u]
        self.i_u += 1
        return v.clone()


class RecordingNoise:
    def __init__(self, inner):
        self.inner = inner
        self.normals = []
        self.uniforms = []

    def normal(self, n, shape, step, tag):
        v = self.inner.normal(n, shape, step, tag)
        self.normals.append(v.clone())
        return v

    def uniform(self, n, step, tag):
        v = self.inner.uniform(n, step, tag)
        self.uniforms.append(v.clone())
        return v


class PhiloxNoise:
    def __init__(self, seed, chain_offset=0, rounds=10):
        self.seed = seed
        self.chain_offset = chain_offset
        self.rounds = rounds   # 10: the library's stream; 7: the opt-in Philox4x32-7 stream

    def _ids(self, n):
        return np.arange(self.chain_offset, self.chain_offset + n, dtype=np.uint32)

    def normal(self, n, shape, step, tag):
        d = int(math.prod(shape))
        return torch.from_numpy(philox.normal_field(self.seed, self._ids(n), step, d, tag, self.rounds)).reshape(n, *shape)

    def uniform(self, n, step, tag):
        if tag == philox.TAG_ACCEPT:
            return torch.from_numpy(philox.accept_uniform(self.seed, self._ids(n), step, self.rounds))
        return torch.from_numpy(philox.jump_uniform(self.seed, self._ids(n), step, self.rounds))


# --------------------------------------------------------------------------- streaming moments
class Moments:
    """`MCMCExpectation.update` for f = id and f = x^2 (nfmc/algorithms/sampling/base.py:75-95,154-161)."""

    def __init__(self, event_dims=1):
        self.event_dims = event_dims
        self.n_seen = 0
        self.first = 0.0
        self.second = 0.0

    def update(self, x):
        # x: (k, n, *event) or (n, *event); base.py:82-95
        if x.dim() == self.event_dims + 1:
            x = x[None]
        n_new = x.shape[0] * x.shape[1]
        w_old = self.n_seen / (self.n_seen + n_new)
        w_new = n_new / (self.n_seen + n_new)
        self.first = torch.add(w_old * self.first, w_new * torch.mean(x, dim=(0, 1)))
        self.second = torch.add(w_old * self.second, w_new * torch.mean(x ** 2, dim=(0, 1)))
        self.n_seen += n_new

    @classmethod
    def for_event(cls, event_shape):
        return cls(len(event_shape))

    @property
    def variance(self):
        return self.second - self.first ** 2


@dataclass
class Trace:
    """What a run produced; `samples` is the reference's `out.samples` (stacked per stored step)."""
    samples: List[torch.Tensor] = field(default_factory=list)
    masks: List[torch.Tensor] = field(default_factory=list)
    log_ratios: List[torch.Tensor] = field(default_factory=list)
    uniforms: List[torch.Tensor] = field(default_factory=list)
    n_accepted: int = 0
    n_attempted: int = 0
    n_target_calls: int = 0
    n_target_gradient_calls: int = 0
    n_accepted_jumps: int = 0
    n_attempted_jumps: int = 0
    n_refits: int = 0
    n_divergences: int = 0
    moments: Optional[Moments] = None
    last: Optional[torch.Tensor] = None

    def stacked(self):
        return torch.stack(self.samples, dim=0)


def _value_and_grad(target, x):
    # langevin.py:66-70 / hmc.py:40-48: autograd of target(x).sum()
    x = x.detach().clone().requires_grad_(True)
    u = target(x)
    g, = torch.autograd.grad(u.sum(), x)
    return u.detach(), g.detach()


def _sum_event(v):
    return v.reshape(v.shape[0], -1).sum(-1)


# --------------------------------------------------------------------------- A.1 MALA / ULA
def proposal_potential(x_prime, x, grad_u_x, a_diag, tau):
    """langevin.py:31-42."""
    term = x_prime - x + tau * a_diag.view(1, -1) * grad_u_x
    return (term * (1 / a_diag.view(1, -1)) * term).sum(dim=-1) / (4 * tau)


def langevin_propose(x, target, step_size, inv_mass_diag, adjustment, noise, step):
    """One `Langevin.propose` (langevin.py:61-122).  x: (n, d) (flattened events).

    Returns x_prime, mask, log_ratio (None for ULA), log_u (None for ULA).
    Draw order: randn_like(x) first (:63), rand(n) after the second target call (:106).
    """
    n = x.shape[0]
    eps = noise.normal(n, x.shape[1:], step, philox.TAG_NOISE)
    u_x, grad_u_x = _value_and_grad(target, x)
    grad_term = -step_size / inv_mass_diag[None].square() * grad_u_x           # :74
    noise_term = math.sqrt(2 * step_size) / inv_mass_diag[None] * eps           # :75
    x_prime = x + grad_term + noise_term                                        # :76
    if not adjustment:
        return x_prime, torch.ones(n, dtype=torch.bool), None, None
    u_xp, grad_u_xp = _value_and_grad(target, x_prime)
    a_diag = 1 / inv_mass_diag ** 2
    log_ratio = (-u_xp) - (-u_x) \
        + (-proposal_potential(x, x_prime, grad_u_xp, a_diag, step_size)) \
        - (-proposal_potential(x_prime, x, grad_u_x, a_diag, step_size))        # :88-105, util.py:392
    log_u = torch.log(noise.uniform(n, step, philox.TAG_ACCEPT))                # :106
    return x_prime, log_u < log_ratio, log_ratio, log_u


# --------------------------------------------------------------------------- random-walk MH (f2)
def mh_propose(x, target, inv_mass_diag, adjustment, noise, step):
    """One `MH.propose` (nfmc/algorithms/sampling/mcmc/mh.py:44-73): x' = x + eps * inv_mass_diag, symmetric."""
    n = x.shape[0]
    x_prime = x + noise.normal(n, x.shape[1:], step, philox.TAG_NOISE) * inv_mass_diag[None]        # :50-55
    if not adjustment:
        return x_prime, torch.ones(n, dtype=torch.bool), None, None
    log_ratio = (-target(x_prime)) - (-target(x))                                                    # :58, util.py:392
    log_u = torch.log(noise.uniform(n, step, philox.TAG_ACCEPT))                                      # :59
    return x_prime, log_u < log_ratio, log_ratio, log_u


# --------------------------------------------------------------------------- A.2 HMC / UHMC
def hmc_propose(x, target, step_size, inv_mass_diag, n_leapfrog, adjustment, noise, step):
    """One `HMC.propose` (hmc.py:96-126) with `hmc_trajectory` (:61-77).  x: (n, d)."""
    n = x.shape[0]
    p = noise.normal(n, x.shape[1:], step, philox.TAG_NOISE) * (1 / inv_mass_diag.sqrt())      # :100
    p0 = p
    xq = x
    for _ in range(n_leapfrog):                                                   # :67-71
        p = p - step_size / 2 * _value_and_grad(target, xq)[1]
        xq = xq + step_size * (p * inv_mass_diag)
        p = p - step_size / 2 * _value_and_grad(target, xq)[1]
    if not adjustment:
        return xq, torch.ones(n, dtype=torch.bool), None, None
    h0 = target(x) + 0.5 * _sum_event(p0 ** 2 * inv_mass_diag)                   # :103-106
    h1 = target(xq) + 0.5 * _sum_event(p ** 2 * inv_mass_diag)                   # :107-110
    log_ratio = -h1 - (-h0)                                                       # :111
    log_u = torch.log(noise.uniform(n, step, philox.TAG_ACCEPT))                  # :112
    return xq, log_u < log_ratio, log_ratio.detach(), log_u


# --------------------------------------------------------------------------- inner loop
def mcmc_sample(x0, target, kind, n_iterations, step_size, inv_mass_diag=None, n_leapfrog=20,
                adjustment=True, noise=None, step0=0, store=True):
    """`MCMCSampler.sample` (mcmc/base.py:56-102) for kind in {'langevin', 'hmc', 'mh'}.

    Events are flattened to (n, d) for the arithmetic; samples keep the flattened shape.
    """
    noise = noise or TorchNoise()
    n = x0.shape[0]
    x = x0.detach().clone().reshape(n, -1)
    d = x.shape[1]
    if inv_mass_diag is None:
        inv_mass_diag = torch.ones(d)
    tr = Trace(moments=Moments.for_event((d,)))
    for it in range(n_iterations):
        step = step0 + it
        # The reference's failure channel (langevin.py:111-114, hmc.py:117-120, mh.py:63-66): a ValueError raised by
        # the target inside propose() rejects every chain for this step and counts ONE divergence; the call counters
        # are booked after the try block, whatever happened inside it.  The step's normals were drawn before the first
        # target call, its uniforms are only drawn once every target call has returned.
        try:
            if kind == 'langevin':
                x_prime, mask, lr, lu = langevin_propose(x, target, step_size, inv_mass_diag, adjustment, noise, step)
            elif kind == 'mh':
                x_prime, mask, lr, lu = mh_propose(x, target, inv_mass_diag, adjustment, noise, step)
            else:
                x_prime, mask, lr, lu = hmc_propose(x, target, step_size, inv_mass_diag, n_leapfrog, adjustment, noise, step)
        except ValueError:
            x_prime, mask, lr, lu = x, torch.zeros(n, dtype=torch.bool), None, None
            tr.n_divergences += 1
        if kind == 'langevin':
            calls = grads = n * (2 if adjustment else 1)                         # langevin.py:116-120
        elif kind == 'mh':
            calls, grads = (2 * n if adjustment else 0), 0                       # mh.py:67-71
        else:
            grads = 2 * n_leapfrog * n                                           # hmc.py:122-125
            calls = grads + (2 * n if adjustment else 0)
        x = x.detach().clone()
        x[mask] = x_prime.detach()[mask]                                         # mcmc/base.py:77
        tr.n_accepted += int(mask.sum())
        tr.n_attempted += n
        tr.n_target_calls += calls
        tr.n_target_gradient_calls += grads
        tr.moments.update(x)                                                     # :86
        tr.masks.append(mask)
        if lr is not None:
            tr.log_ratios.append(lr)
            tr.uniforms.append(lu)
        if store:
            tr.samples.append(x.clone())                                          # :90
        tr.last = x.clone()
    return tr


# --------------------------------------------------------------------------- A.3 jump loop
def jump_sample(x0, target, flow, inner_kind, n_outer, n_inner, step_size, inv_mass_diag=None,
                n_leapfrog=20, adjustment=True, adjusted_jumps=True, noise=None, store=True):
    """`JumpNFMC.sample` (nfmc/algorithms/sampling/nfmc/jump.py:156-246), fit_nf=False.

    Transition numbering for PhiloxNoise: outer i, inner k -> i*(n_inner+1)+k; the jump is
    transition i*(n_inner+1)+n_inner.
    """
    noise = noise or TorchNoise()
    n = x0.shape[0]
    event = x0.shape[1:]
    x = x0.detach().clone().reshape(n, -1)
    tr = Trace(moments=Moments.for_event((x.shape[1],)))
    for i in range(n_outer):
        base = i * (n_inner + 1)
        inner = mcmc_sample(x, target, inner_kind, n_inner, step_size, inv_mass_diag, n_leapfrog,
                            adjustment, noise, step0=base, store=True)          # jump.py:178
        tr.n_accepted += inner.n_accepted
        tr.n_attempted += inner.n_attempted
        tr.n_target_calls += inner.n_target_calls
        tr.n_target_gradient_calls += inner.n_target_gradient_calls
        tr.n_divergences += inner.n_divergences                                  # jump.py:183
        tr.masks += inner.masks
        tr.log_ratios += inner.log_ratios
        tr.moments.update(inner.stacked())                                       # :188
        if store:
            tr.samples += inner.samples                                           # :189
        jstep = base + n_inner
        z = noise.normal(n, event, jstep, philox.TAG_LATENT)
        with torch.no_grad():
            x_prime, ld_inv = flow.bijection.inverse(z)                          # flow.sample, :205
            f_x_prime = flow.base_log_prob(z) - ld_inv
        x_prime = x_prime.reshape(n, -1)
        x = inner.last                                                            # :209
        if adjusted_jumps:
            try:                                                                  # :210-227
                u_x = target(x)                                                   # :212
                u_xp = target(x_prime)                                            # :213
                tr.n_target_calls += 2 * n
                with torch.no_grad():
                    f_x = flow.log_prob(x.reshape(n, *event))                     # :218
                log_alpha = (-u_xp) - (-u_x) + f_x - f_x_prime                    # :219-224, util.py:392
                log_u = noise.uniform(n, jstep, philox.TAG_JUMP).log()            # :225
                mask = log_u < log_alpha
                tr.log_ratios.append(log_alpha.detach())
                tr.uniforms.append(log_u)
            except ValueError:                                                    # :226-227: reject all, no divergence
                mask = torch.zeros(n, dtype=torch.bool)
        else:
            mask = torch.ones(n, dtype=torch.bool)
        x = x.clone()
        x[mask] = x_prime[mask]                                                   # :231
        tr.n_attempted_jumps += n
        tr.n_accepted_jumps += int(mask.sum())
        tr.masks.append(mask)
        tr.moments.update(x)                                                      # :240
        if store:
            tr.samples.append(x.clone())                                          # :243
        tr.last = x.clone()
    return tr


# --------------------------------------------------------------------------- A.4 FixedIMH
def imh_sample(x0, target, flow, n_iterations, noise=None, store=True):
    """`FixedIMH.sample` (nfmc/algorithms/sampling/nfmc/imh.py:200-255)."""
    noise = noise or TorchNoise()
    n = x0.shape[0]
    event = x0.shape[1:]
    x = x0.detach().clone().reshape(n, -1)
    tr = Trace(moments=Moments.for_event((x.shape[1],)))
    with torch.no_grad():
        f_x = flow.log_prob(x.reshape(n, *event))                                 # :214
        for it in range(n_iterations):
            z = noise.normal(n, event, it, philox.TAG_LATENT)
            x_prime, ld_inv = flow.bijection.inverse(z)                           # :221
            f_xp = flow.base_log_prob(z) - ld_inv
            x_prime = x_prime.reshape(n, -1)
            log_alpha = (-target(x_prime)) - (-target(x)) + f_x - f_xp           # :223-228
            log_u = noise.uniform(n, it, philox.TAG_JUMP).log()                   # :229
            mask = torch.less(log_u, log_alpha)                                   # :230
            x[mask] = x_prime[mask]                                               # :232
            f_x[mask] = f_xp[mask]                                                # :233
            tr.moments.update(x)                                                  # :242
            tr.n_target_calls += 2 * n
            tr.n_accepted += int(mask.sum())
            tr.n_attempted += n
            tr.masks.append(mask.clone())
            tr.log_ratios.append(log_alpha.clone())
            tr.uniforms.append(log_u)
            if store:
                tr.samples.append(x.clone())
            tr.last = x.clone()
    return tr


class TorchHostDraws:
    """Host-side scalar draws of AdaptiveIMH (imh.py:148,156-160) from torch's global generator."""

    def rand(self):
        return float(torch.rand(size=()))

    def randint(self, low, high):
        return int(torch.randint(low=low, high=high, size=()))


class ReplayHostDraws:
    def __init__(self, uniforms, ints):
        self.uniforms, self.ints = list(uniforms), list(ints)

    def rand(self):
        return float(self.uniforms.pop(0))

    def randint(self, low, high):
        k = int(self.ints.pop(0))
        assert low <= k < high
        return k


def bounded_geom_index(p, max_val, u):
    """`sample_bounded_geom` (imh.py:39-45) with the uniform passed in."""
    v = torch.arange(0, max_val + 1)
    pdf = p * (1 - p) ** (max_val - v) / (1 - (1 - p) ** (max_val + 1))
    cdf = torch.cumsum(pdf, dim=0)
    return int(torch.searchsorted(cdf, torch.tensor(u, dtype=cdf.dtype), right=True))


def adaptive_imh_sample(x0, target, flow, n_iterations, adaptation_dropoff=0.9999, train_distribution='uniform',
                        noise=None, host=None, fit_fn=None):
    """`AdaptiveIMH.sample` (nfmc/algorithms/sampling/nfmc/imh.py:103-181): IMH whose proposal flow is refitted
    on one stored state per iteration with probability dropoff^i.  log q is recomputed every iteration
    (the flow moves).  Quirk kept: the 2n target evaluations are booked as gradient calls (:142)."""
    noise = noise or TorchNoise()
    host = host or TorchHostDraws()
    fit_fn = fit_fn or (lambda f, xt: f.fit(xt, n_epochs=1, show_progress=False))
    n = x0.shape[0]
    event = x0.shape[1:]
    x = x0.detach().clone().reshape(n, -1)
    tr = Trace(moments=Moments.for_event((x.shape[1],)))
    for it in range(n_iterations):
        with torch.no_grad():
            z = noise.normal(n, event, it, philox.TAG_LATENT)
            x_prime, ld_inv = flow.bijection.inverse(z)                              # :123
            x_prime = x_prime.reshape(n, -1)
            f_xp = flow.base_log_prob(z) - ld_inv
            f_x = flow.log_prob(x.reshape(n, *event))                                # :128
            log_alpha = (-target(x_prime)) - (-target(x)) + f_x - f_xp              # :125-130
            log_u = noise.uniform(n, it, philox.TAG_JUMP).log()                      # :131
            mask = torch.less(log_u, log_alpha)                                      # :132
            x = x.clone()
            x[mask] = x_prime[mask]                                                  # :133
        tr.moments.update(x)                                                         # :139
        tr.n_target_gradient_calls += 2 * n                                          # :141 (sic)
        tr.n_accepted += int(mask.sum())
        tr.n_attempted += n
        tr.masks.append(mask.clone())
        tr.log_ratios.append(log_alpha.clone())
        tr.samples.append(x.clone())                                                 # :145
        tr.last = x.clone()
        if host.rand() < adaptation_dropoff ** it:                                   # :147-149
            n_samples = len(tr.samples)
            if train_distribution == 'uniform':
                k = host.randint(0, n_samples)                                       # :156
            elif train_distribution == 'bounded_geom_approx':
                k = host.randint(max(0, n_samples - 100), n_samples)                 # :158
            elif train_distribution == 'bounded_geom':
                k = bounded_geom_index(0.025, n_samples - 1, host.rand())            # :160
            else:
                raise ValueError
            x_train = tr.samples[k].reshape(n, *event)
            saved = {kk: v.detach().clone() for kk, v in flow.state_dict().items()}  # :166
            try:
                fit_fn(flow, x_train)                                                # :168
                tr.n_refits += 1
            except ValueError:
                flow.load_state_dict(saved)                                          # :170
    return tr


# --------------------------------------------------------------------------- A.5 NeuTra
def neutra_adjusted_target(flow, target, event):
    """`NeuTra.adjusted_target` (nfmc/algorithms/sampling/nfmc/neutra.py:58-68)."""

    def adjusted(z):
        x, log_det_inverse = flow.bijection.inverse(z.reshape(z.shape[0], *event))
        log_prob = -target(x.reshape(z.shape[0], -1))
        return -(log_prob + log_det_inverse)

    return adjusted


def neutra_hmc_sample(z0, target, flow, n_iterations, step_size, inv_mass_diag=None, n_leapfrog=20,
                      noise=None, store=True):
    """`NeuTra.sample` (neutra.py:109-129): HMC on the adjusted target; samples/moments stay latent
    (the data_transform assigned at neutra.py:122 / mcmc/base.py:63 never reaches the moments: SURVEY App. C #1)."""
    event = z0.shape[1:]
    return mcmc_sample(z0, neutra_adjusted_target(flow, target, event), 'hmc', n_iterations, step_size,
                       inv_mass_diag, n_leapfrog, True, noise, 0, store)


# --------------------------------------------------------------------------- a12 refit buffer
def train_val_split(x, train_pct, max_train_size, max_val_size, perm=None):
    """`train_val_split` (nfmc/algorithms/sampling/tuning.py:44-65); `perm` replays torch.randperm."""
    x_train = x.flatten(0, 1)
    if perm is None:
        perm = torch.randperm(len(x_train))
    x_train = x_train[perm]
    n_train = int(train_pct * len(x_train))
    x_train, x_val = x_train[:n_train], x_train[n_train:]
    return x_train[:max_train_size], x_val[:max_val_size]


# --------------------------------------------------------------------------- A.7 tuning (warmup)
class DualAveraging:
    """`DualAveraging` (tuning.py:15-41)."""

    def __init__(self, initial_step_size, target_acceptance_rate=0.651, kappa=0.75, gamma=0.05, t0=10):
        self.t = t0
        self.error_sum = 0.0
        self.log_step_averaged = math.log(initial_step_size)
        self.mu = math.log(10 * initial_step_size)
        self.kappa, self.gamma, self.target = kappa, gamma, target_acceptance_rate

    def step(self, acceptance_rate_error):
        self.error_sum += float(acceptance_rate_error)
        log_step = self.mu - self.error_sum / (math.sqrt(self.t) * self.gamma)
        eta = self.t ** -self.kappa
        self.log_step_averaged = eta * log_step + (1 - eta) * self.log_step_averaged
        self.t += 1

    @property
    def value(self):
        return math.exp(self.log_step_averaged)
