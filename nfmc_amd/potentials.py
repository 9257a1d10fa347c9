"""Closed-form potential descriptors: callables (usable as `target` exactly like the reference's
`target(x) -> (n,)` callables, nfmc/sample.py:35-37) that ALSO describe themselves to the HIP kernels,
so U and grad U are evaluated in-kernel instead of by `torch.autograd.grad` (langevin.py:66-68).

The reference's own potential objects live in the absent third-party package `potentials`
(nfmc/sample.py:17); `SumOfSquares` is the README/test potential (README.md:45-46, test/util.py:4-5),
`Funnel` is the build's C4 potential.

`recognize(target, event_shape)` lets plain Python callables such as the README's
`lambda x: torch.sum(x**2, dim=1)` take the fused path: it probes the callable, fits
U = sum_j a_j (x_j - b_j)^2 + c and accepts it only if the fit reproduces the callable (and its autograd
gradient) on fresh points at radii from 0.1 to 100 and around the run's own x0 to 1e-5 relative.  It is an
inference from finitely many probes (see its docstring); `fuse='never'` turns it off.
"""
import ctypes as C
import logging
import math
from typing import Optional, Sequence, Tuple, Union

import torch

from . import hip


class Potential:
    """Base: `event_shape`, `__call__(x) -> (n,)` in torch ops, `descriptor(device)` for the kernels."""

    event_shape: Tuple[int, ...]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def descriptor(self, device) -> hip.NfmcPotential:
        raise NotImplementedError

    @property
    def event_size(self):
        return int(math.prod(self.event_shape))


class QuadraticPotential(Potential):
    """U(x) = sum_j a_j (x_j - b_j)^2 with per-coordinate or scalar a, b."""

    def __init__(self, event_shape, a=1.0, b=0.0):
        if isinstance(event_shape, int):
            event_shape = (event_shape,)
        self.event_shape = tuple(event_shape)
        d = self.event_size
        self.a = self._norm(a, d)
        self.b = self._norm(b, d)
        self._dev = {}

    @staticmethod
    def _norm(v, d):
        if isinstance(v, (int, float)):
            return float(v)
        t = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            return float(t)
        if t.numel() != d:
            raise ValueError('potential parameter must be a scalar or have event_size entries')
        return t.contiguous()

    def __call__(self, x):
        n = x.shape[0]
        xf = x.reshape(n, -1)
        a, b = self._params_like(xf)
        return torch.sum(a * (xf - b) ** 2, dim=-1)

    def _params_like(self, xf):
        """a, b next to `xf`: the device copies the descriptor uses are kept (a host-to-device copy per call would also keep
        a fit step that evaluates the potential out of a HIP graph)"""
        if isinstance(self.a, float) and isinstance(self.b, float):
            return self.a, self.b
        if xf.dtype == torch.float32 and xf.is_cuda:
            key = str(xf.device)
            if key not in self._dev:
                self._dev[key] = tuple(None if isinstance(v, float) else v.to(xf.device) for v in (self.a, self.b))
            da, db = self._dev[key]
            return (self.a if da is None else da), (self.b if db is None else db)
        return (self.a if isinstance(self.a, float) else self.a.to(xf)), (self.b if isinstance(self.b, float) else self.b.to(xf))

    def descriptor(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(None if isinstance(v, float) else v.to(device) for v in (self.a, self.b))
        da, db = self._dev[key]
        return hip.NfmcPotential(hip.POT_QUADRATIC, 0, hip.ptr(da), hip.ptr(db),
                                 self.a if isinstance(self.a, float) else 0.0,
                                 self.b if isinstance(self.b, float) else 0.0)


class SumOfSquares(QuadraticPotential):
    """U(x) = sum x^2: the reference's "standard Gaussian potential" (target N(0, I/2))."""

    def __init__(self, event_shape):
        super().__init__(event_shape, 1.0, 0.0)


class DiagonalGaussian(QuadraticPotential):
    """U(x) = sum (x - mu)^2 / (2 sigma^2) (normalising constant dropped, as MH ratios ignore it)."""

    def __init__(self, event_shape, mu=0.0, sigma=1.0):
        s = torch.as_tensor(sigma, dtype=torch.float32)
        super().__init__(event_shape, (1.0 / (2.0 * s * s)) if s.numel() > 1 else 1.0 / (2.0 * float(s) ** 2), mu)


class Funnel(Potential):
    """U(x) = x_0^2/(2 s^2) + sum_{i>=1} [ x_i^2 / (2 e^{x_0}) + x_0/2 ]  (Neal's funnel, s = 3)."""

    def __init__(self, event_shape, scale: float = 3.0):
        if isinstance(event_shape, int):
            event_shape = (event_shape,)
        self.event_shape = tuple(event_shape)
        self.scale = float(scale)

    def __call__(self, x):
        n = x.shape[0]
        xf = x.reshape(n, -1)
        x0 = xf[:, 0]
        return x0 ** 2 / (2 * self.scale ** 2) + 0.5 * torch.exp(-x0) * torch.sum(xf[:, 1:] ** 2, dim=-1) \
            + 0.5 * (xf.shape[1] - 1) * x0

    def descriptor(self, device):
        return hip.NfmcPotential(hip.POT_FUNNEL, 0, None, None, self.scale, 0.0)


_log = logging.getLogger('nfmc_amd')
_announced = set()


def recognize(target, event_shape, rtol: float = 1e-5, x_scale: float = None) -> Optional[Potential]:
    """A QuadraticPotential that reproduces the plain callable `target` on every probe below, or None.

    This is an inference from finitely many evaluations, not a proof: a callable that is quadratic on all probed
    points and something else elsewhere (walls or modes beyond the probed radii), or one that changes between calls,
    would be sampled as the fitted Gaussian.  The probes therefore cover, besides the unit directions used for the
    fit, random points at radii 0.1 ... 100 per coordinate AND at 1x / 3x / 10x the largest |x0| of the run
    (`x_scale`), a repeated evaluation (stateful / stochastic targets differ), and the autograd gradient; any mismatch,
    non-finite value or exception -> None (the sampler then takes the split path, where an exception of the target
    surfaces unchanged).  The first time a callable is rerouted one line is logged on the `nfmc_amd` logger.  Opt out
    with `sample(..., fuse='never')` / `sampler.fuse = False`, or pass a `Potential` to be explicit."""
    if isinstance(target, Potential):
        return target
    d = int(math.prod(event_shape))
    try:
        with torch.no_grad():
            g = torch.Generator().manual_seed(0x5EED)
            z = torch.zeros(1, *event_shape, dtype=torch.float64)
            c0 = target(z).reshape(-1).double()
            eye = torch.eye(d, dtype=torch.float64).reshape(d, *event_shape)
            up = target(eye).reshape(-1).double()
            um = target(-eye).reshape(-1).double()
            if not (torch.isfinite(c0).all() and torch.isfinite(up).all() and torch.isfinite(um).all()):
                return None
            # U(t e_j) = a_j t^2 - 2 a_j b_j t + (c0)  =>  a_j = (U+ + U- - 2 c0)/2, a_j b_j = (U- - U+)/4
            a = (up + um - 2 * c0) / 2
            ab = (um - up) / 4
            b = torch.where(a > 0, ab / torch.where(a > 0, a, torch.ones_like(a)), torch.zeros_like(a))
            const = c0 - (a * b * b).sum()
            if not torch.isfinite(a).all() or not torch.isfinite(b).all() or (a < 0).any():
                return None
            scales = [0.1, 1.0, 7.0, 30.0, 100.0]
            if x_scale is not None and math.isfinite(x_scale) and x_scale > 0:
                scales += [x_scale, 3.0 * x_scale, 10.0 * x_scale]
            for scale in scales:
                x = scale * torch.randn(8, *event_shape, generator=g, dtype=torch.float64)
                want = target(x).reshape(-1).double()
                got = (a * (x.reshape(8, -1) - b) ** 2).sum(-1) + const
                if not torch.allclose(got, want, rtol=rtol, atol=rtol * (1 + want.abs().max())):
                    return None
            if not torch.equal(target(x).reshape(-1).double(), want):   # same input, same answer
                return None
        # gradient check through autograd (catches targets that detach / are piecewise)
        with torch.enable_grad():
            x = torch.randn(4, *event_shape, generator=g, dtype=torch.float64).requires_grad_(True)
            gr, = torch.autograd.grad(target(x).sum(), x)
        want_g = (2 * a * (x.detach().reshape(4, -1) - b)).reshape(x.shape)
        if not torch.allclose(gr, want_g, rtol=rtol, atol=rtol * (1 + want_g.abs().max())):
            return None
    except Exception:
        return None

    def simplify(v):
        return float(v[0]) if bool((v == v[0]).all()) else v.float()

    key = getattr(target, '__qualname__', None) or type(target).__name__
    if key not in _announced:
        _announced.add(key)
        _log.warning('nfmc_amd: target %r reproduces U(x) = sum_j a_j (x_j - b_j)^2 + c on every probe (radii up to %.3g); '
                     'it is evaluated in closed form inside the HIP kernels.  Pass fuse="never" to sample() to keep '
                     'calling the Python callable.', key, max(scales))
    return QuadraticPotential(tuple(event_shape), simplify(a), simplify(b))
