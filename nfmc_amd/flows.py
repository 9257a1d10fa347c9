"""Host-side flow objects: the surface nfmc uses from `torchflows`, backed by the
HIP kernels of libnfmc_hip.so.

    Flow(RealNVP(event_shape, n_layers=2, conditioner_kwargs={'n_hidden': .., 'n_layers': ..}))
    flow.sample(n, return_log_prob=False, no_grad=False)    jump.py:205, imh.py:128,221
    flow.log_prob(x)                                         jump.py:218, imh.py:214
    flow.bijection.forward(x) -> (z, logdet)                 tess.py:113
    flow.bijection.inverse(z) -> (x, logdet)                 neutra.py:60,122
    flow.bijection.layers, .event_shape                      test/test_flow_kwargs.py:18
    flow.to(device), .get_device(), .state_dict(), .load_state_dict(), .parameters()
    flow.fit(x_train, x_val=..., ...), flow.variational_fit(log_prob_fn, ...)   (training: flow_training.py)

torchflows itself is absent from the reference tree and unpinned (pyproject.toml:23), so the RealNVP
here follows the build's own spec (DESIGN.md "RealNVP spec", restated on CPU in oracle/flow.py).
The parameters are ordinary `nn.Parameter`s; `packed(device)` lays them out for the kernels
(include/nfmc_hip.h, NfmcRealNVP) and is cached until a parameter changes.
"""
import ctypes as C
import math
import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import hip

MIN_SCALE = 1e-3


def default_hidden(d_a: int) -> int:
    return max(4, int(3 * math.log10(max(d_a, 1))))


class ElementwiseAffine(nn.Module):
    """z = exp(log_scale) * x + shift."""

    def __init__(self, d):
        super().__init__()
        self.log_scale = nn.Parameter(torch.zeros(d))
        self.shift = nn.Parameter(torch.zeros(d))


class ReversePermutation(nn.Module):
    """z[j] = x[d-1-j]; folded into the kernels' index arithmetic, never materialised."""


class AffineCoupling(nn.Module):
    """Half-split affine coupling; `conditioner` is the MLP d_a -> H (tanh) x n_layers -> 2 d_b."""

    def __init__(self, d, n_hidden=None, n_layers=2):
        super().__init__()
        self.d_a = d // 2
        self.d_b = d - self.d_a
        self.n_hidden = default_hidden(self.d_a) if n_hidden is None else int(n_hidden)
        self.n_layers = int(n_layers)
        if self.n_layers < 1:
            raise ValueError('conditioner needs at least one hidden layer')
        dims = [self.d_a] + [self.n_hidden] * self.n_layers + [2 * self.d_b]
        self.conditioner = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])


RQS_BINS, RQS_BOUND = 8, 5.0


class RQSCoupling(AffineCoupling):
    """Half-split rational-quadratic spline coupling ('c-rqnsf'): the conditioner emits, per target coordinate,
    3K - 1 numbers (K widths | K heights | K - 1 interior derivatives), target-major (oracle/flow.py: rqs_apply)."""

    def __init__(self, d, n_hidden=None, n_layers=2, n_bins=RQS_BINS):
        super().__init__(d, n_hidden, n_layers)
        self.n_bins = int(n_bins)
        dims = [self.d_a] + [self.n_hidden] * self.n_layers + [(3 * self.n_bins - 1) * self.d_b]
        self.conditioner = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])


class RealNVP(nn.Module):
    """[ElementwiseAffine] + n_layers x [ReversePermutation, AffineCoupling] + [ElementwiseAffine]."""

    min_scale = MIN_SCALE   # scale = exp(u/2 + log(1 - min_scale)) + min_scale
    n_bins = 0              # 0: affine couplings; K: spline couplings with K bins (CRQNSF)
    spline_bound = RQS_BOUND
    coupling_class = AffineCoupling

    def __init__(self, event_shape, n_layers: int = 2, conditioner_kwargs: Optional[dict] = None, **kwargs):
        super().__init__()
        if isinstance(event_shape, int):
            event_shape = (event_shape,)
        self.event_shape = tuple(int(s) for s in event_shape)
        self.d = int(math.prod(self.event_shape))
        ck = dict(conditioner_kwargs or {})
        if 'n_hidden' in kwargs:
            ck.setdefault('n_hidden', kwargs['n_hidden'])
        self.n_coupling = int(n_layers)
        layers = [ElementwiseAffine(self.d)]
        for _ in range(self.n_coupling):
            layers += [ReversePermutation(), self._make_coupling(ck)]
        layers.append(ElementwiseAffine(self.d))
        self.layers = nn.ModuleList(layers)
        self._pack_cache = None

    def _make_coupling(self, ck):
        return AffineCoupling(self.d, ck.get('n_hidden'), ck.get('n_layers', 2))

    def __getstate__(self):
        """deepcopy / pickle / torch.save: the device-side caches (ctypes structs holding raw pointers, the module
        list of the version key) are rebuilt on demand, never copied."""
        state = self.__dict__.copy()
        state['_pack_cache'] = None
        state.pop('_mods_cache', None)
        state.pop('_device_fit', None)      # flow_training.DeviceFit: device buffers + ctypes structs of THIS object
        state.pop('_wide_scratch', None)    # workspace of the streamed matrix-core kernels
        return state

    # ------------------------------------------------------------------ packing for the kernels
    @property
    def couplings(self):
        return [m for m in self.layers if isinstance(m, AffineCoupling)]

    @property
    def n_hidden(self):
        c = self.couplings
        return c[0].n_hidden if c else 4

    @property
    def n_hidden_layers(self):
        c = self.couplings
        return c[0].n_layers if c else 1

    def _version_key(self, device):
        # live parameter dicts of a cached module list: `self.parameters()` re-walks the module tree (~50 us for the
        # default flow, once per jump launch); reassigned or in-place updated parameters still change the key
        mods = self.__dict__.get('_mods_cache')
        if mods is None:
            mods = self.__dict__['_mods_cache'] = [m for m in self.modules() if m._parameters]
        return (str(device),) + tuple((p.data_ptr(), p._version) for m in mods for p in m._parameters.values()
                                      if p is not None)

    def default_min_hidden(self) -> int:
        """Width to present to the flow kernels by default: at d = 64 / 128 a conditioner of width 9..32 is faster
        zero-padded to 64 on the matrix cores (flow_mfma.hip) than on the one-chain-per-lane kernels (C3 shape, jump
        per outer step: H = 32 0.57 -> 0.47 ms).  Narrower ones stay on the register-layout kernel."""
        ok = (self.d in (64, 128) and 8 < self.n_hidden <= 32 and self.n_hidden_layers in (1, 2) and self.n_bins == 0
              and os.environ.get('NFMC_FLOW_NO_MFMA') is None)
        return 64 if ok else 0

    def packed(self, device, min_hidden: int = None):
        """(NfmcRealNVP struct, keep-alive tensors) on `device`; rebuilt only when a parameter changed.

        `min_hidden`: present the conditioner as at least that wide (extra hidden units have zero weights, so
        tanh(0) = 0 contributes nothing: same function).  NeuTra uses 64 to reach the matrix-core kernels with
        narrow conditioners (d = 128, H = 8: 6.6 -> ~3 ms per transition)."""
        if min_hidden is None:
            min_hidden = self.default_min_hidden()
        key = self._version_key(device) + (int(min_hidden),)
        cache = self._pack_cache if isinstance(self._pack_cache, dict) else {}
        hit = cache.get(int(min_hidden))
        if hit is not None and hit[0] == key:
            return hit[1]
        lib = hip.lib()
        d, H, nhl = self.d, max(self.n_hidden, int(min_hidden)), self.n_hidden_layers
        hp = int(lib.nfmc_realnvp_padded_hidden(H))
        stride = int(lib.nfmc_coupling_layer_floats(d, H, nhl, self.n_bins))
        out_rows = (3 * self.n_bins - 1) * (d - d // 2) if self.n_bins else 2 * (d - d // 2)
        d_a, d_b = d // 2, d - d // 2
        if hp == 0:
            raise ValueError('RealNVP conditioner width %d is beyond the kernels (max 128)' % H)
        if stride == 0:
            raise ValueError('spline couplings run with conditioners of width <= 32 and %d bins' % RQS_BINS)
        with torch.no_grad():
            # one preallocated host buffer filled by slice copies (torch.cat of ~1e5 floats spins up the CPU
            # thread pool: 30-40 ms per call, most of the host time of a sample() with a wide conditioner)
            # the four ElementwiseAffine vectors ride behind the coupling layers (16-byte aligned), so the whole
            # flow goes to the device in ONE copy (five pageable copies were ~0.1 ms per fresh flow object)
            n_w = max(1, self.n_coupling * stride)
            ea_off = (n_w + 3) // 4 * 4
            d4 = (d + 3) // 4 * 4
            flat = np.zeros(ea_off + 4 * d4, dtype=np.float32)   # numpy: single-threaded, no pool
            for li, cpl in enumerate(self.couplings):
                lin = list(cpl.conditioner)
                cur = [li * stride]

                def put(t, rows=None, cols=None, transpose=False):
                    """copy t (zero-padded to rows x cols, optionally transposed) at the cursor"""
                    t = t.detach().float().cpu().numpy()
                    if t.ndim == 1:
                        n_out = rows if rows is not None else t.shape[0]
                        flat[cur[0]:cur[0] + t.shape[0]] = t
                        cur[0] += n_out
                        return
                    if transpose:   # stored as (cols, rows): element (c, r) = t[r, c]
                        view = flat[cur[0]:cur[0] + rows * cols].reshape(cols, rows)
                        view[:t.shape[1], :t.shape[0]] = t.T
                    else:
                        view = flat[cur[0]:cur[0] + rows * cols].reshape(rows, cols)
                        view[:t.shape[0], :t.shape[1]] = t
                    cur[0] += rows * cols

                if hp > 32:  # matrix-core path: both orientations (csrc/mfma_device.hpp)
                    put(lin[0].weight, hp, d_a)
                    put(lin[0].weight, hp, d_a, transpose=True)
                    put(lin[0].bias, hp)
                    for l in lin[1:-1]:
                        put(l.weight, hp, hp)
                        put(l.weight, hp, hp, transpose=True)
                        put(l.bias, hp)
                    put(lin[-1].weight, 2 * d_b, hp)
                    put(lin[-1].weight, 2 * d_b, hp, transpose=True)
                    put(lin[-1].bias)
                else:        # VALU path: W1T | b1 | [WhT | bh] | W3 | b3
                    put(lin[0].weight, hp, d_a, transpose=True)
                    put(lin[0].bias, hp)
                    for l in lin[1:-1]:
                        put(l.weight, hp, hp, transpose=True)
                        put(l.bias, hp)
                    put(lin[-1].weight, out_rows, hp)
                    put(lin[-1].bias)
                assert cur[0] == (li + 1) * stride, (cur[0], li, stride)
            ea0, ea1 = self.layers[0], self.layers[-1]
            for k, t in enumerate((ea0.log_scale, ea0.shift, ea1.log_scale, ea1.shift)):
                flat[ea_off + k * d4:ea_off + k * d4 + d] = t.detach().float().cpu().numpy()
            blob = torch.from_numpy(flat).to(device)
            keep = [blob] + [blob[ea_off + k * d4:ea_off + k * d4 + d] for k in range(4)]
        st = hip.NfmcRealNVP(d, self.n_coupling, H, nhl, float(self.min_scale), int(self.n_bins), hip.ptr(keep[1]),
                             hip.ptr(keep[2]), hip.ptr(keep[3]), hip.ptr(keep[4]), hip.ptr(keep[0]), stride,
                             float(self.spline_bound), 0)
        # the streamed matrix-core kernels (wide conditioner at d = 32 .. 512 other than 64 / 128) work in a slab the CALLER
        # holds: sized here once per flow object at its bound -- 2 x 256 workgroup slots x 128 rows of d floats (67 MB at
        # d = 256), whatever the number of rows of a call -- and kept across re-packs
        need = int(lib.nfmc_flow_scratch_bytes(C.byref(st), 1 << 40, 1))
        if need:
            ws = self.__dict__.get('_wide_scratch')
            if ws is None or ws.device != torch.device(device) or ws.numel() * 4 < need:
                ws = self.__dict__['_wide_scratch'] = torch.empty(need // 4, dtype=torch.float32, device=device)
            st.scratch, st.scratch_bytes = hip.ptr(ws), need
            keep.append(ws)
        cache[int(min_hidden)] = (key, (st, keep))
        self._pack_cache = cache
        return st, keep

    # ------------------------------------------------------------------ bijection API
    def _prep(self, v):
        dev = hip.require_gpu()
        n = v.shape[0]
        return dev, n, v.detach().to(dev, torch.float32).reshape(n, self.d).contiguous()

    def beyond_kernels(self) -> bool:
        """Shapes the flow kernels do not take (events wider than 512 or of one coordinate, conditioners wider than 128,
        32 for splines):
        the passes are then composed from torch ops on the GPU (flow_training.forward_torch / inverse_torch, the same
        arithmetic the training path differentiates), so every flow strategy still runs -- through the samplers'
        split path -- instead of raising."""
        lim = hip.limits()
        if self.d < 2 and self.n_coupling > 0:   # a one-coordinate event: the source half of every coupling is empty (the
            return True                          # conditioner is its last bias); the kernels start at d = 2
        return self.d > lim.max_d_flow or self.n_hidden > (32 if self.n_bins else lim.max_hidden)

    def _composed(self, v, inverse: bool):
        from .flow_training import forward_torch, inverse_torch
        dev, n, vf = self._prep(v)
        if next(self.parameters()).device != dev:
            self.to(dev)
        with torch.no_grad():
            out, ld = (inverse_torch if inverse else forward_torch)(self, vf)
        return out.reshape(n, *self.event_shape), ld

    def forward(self, x):
        """x -> (z, log|det dz/dx|)  [HIP kernel nfmc_realnvp_forward_f32]."""
        if self.beyond_kernels():
            return self._composed(x, False)
        dev, n, xf = self._prep(x)
        st, _keep = self.packed(dev)
        z = torch.empty_like(xf)
        ld = torch.empty(n, dtype=torch.float32, device=dev)
        hip.check(hip.lib().nfmc_realnvp_forward_f32(C.byref(st), hip.ptr(xf), n, hip.ptr(z), hip.ptr(ld), None,
                                                     hip.stream()), 'nfmc_realnvp_forward_f32')
        return z.reshape(n, *self.event_shape), ld

    def inverse(self, z):
        """z -> (x, log|det dx/dz|)  [HIP kernel nfmc_realnvp_inverse_f32]."""
        if self.beyond_kernels():
            return self._composed(z, True)
        dev, n, zf = self._prep(z)
        st, _keep = self.packed(dev)
        x = torch.empty_like(zf)
        ld = torch.empty(n, dtype=torch.float32, device=dev)
        hip.check(hip.lib().nfmc_realnvp_inverse_f32(C.byref(st), hip.ptr(zf), n, hip.ptr(x), hip.ptr(ld), None, None,
                                                     hip.stream()), 'nfmc_realnvp_inverse_f32')
        return x.reshape(n, *self.event_shape), ld


class NICE(RealNVP):
    """Additive (volume-preserving) couplings behind the same kernels (nfmc/util.py:13 'nice'): the RealNVP
    stack with min_scale = 1, for which scale = exp(u/2 + log 0) + 1 = 1 and log-scale = 0 EXACTLY; the
    scale half of the conditioner output is carried but inert (zero gradient)."""

    min_scale = 1.0


class CRQNSF(RealNVP):
    """Coupling rational-quadratic neural spline flow ('c-rqnsf', nfmc/util.py:17): the RealNVP stack with
    RQSCoupling layers (8 bins on [-5, 5], identity outside).  Same kernels for forward / inverse / sampling / the
    flow-proposal Metropolis step (one chain per lane); NeuTra needs the reverse sweep and is not covered."""

    n_bins = RQS_BINS

    def _make_coupling(self, ck):
        return RQSCoupling(self.d, ck.get('n_hidden'), ck.get('n_layers', 2), self.n_bins)


class Flow(nn.Module):
    """`torchflows.Flow` stand-in over a bijection with a standard-normal base."""

    def __init__(self, bijection: RealNVP):
        super().__init__()
        self.bijection = bijection
        self._sample_calls = 0
        self.seed = None  # native-stream seed for sample(); drawn from torch's global RNG on first use

    @property
    def event_shape(self):
        return self.bijection.event_shape

    def get_device(self):
        return next(self.parameters()).device

    @staticmethod
    def _base_log_prob(z):
        zf = z.flatten(1)
        return -0.5 * (zf * zf).sum(-1) - 0.5 * zf.shape[1] * math.log(2.0 * math.pi)

    def log_prob(self, x):
        if self.bijection.beyond_kernels():
            z, ld = self.bijection.forward(x)
            return self._base_log_prob(z) + ld
        dev, n, xf = self.bijection._prep(x)
        st, _keep = self.bijection.packed(dev)
        lp = torch.empty(n, dtype=torch.float32, device=dev)
        hip.check(hip.lib().nfmc_realnvp_forward_f32(C.byref(st), hip.ptr(xf), n, None, None, hip.ptr(lp),
                                                     hip.stream()), 'nfmc_realnvp_forward_f32')
        return lp

    def sample(self, n, return_log_prob=False, no_grad=False, rng=None):
        """x ~ q (inverse pass of in-kernel z ~ N(0, I)); optionally log q(x).  `rng`: hip.NfmcRng override."""
        dev = hip.require_gpu()
        if rng is None:
            if self.seed is None:
                self.seed = int(torch.randint(0, 2 ** 62, ()).item())
            rng = hip.make_rng(self.seed, 0, self._sample_calls)
            self._sample_calls += 1
        if self.bijection.beyond_kernels():   # latents from the same Philox stream the kernel would draw in place
            z = torch.empty(n, self.bijection.d, dtype=torch.float32, device=dev)
            hip.check(hip.lib().nfmc_philox_normals_f32(C.byref(rng), hip.TAG_LATENT, n, self.bijection.d, hip.ptr(z),
                                                        hip.stream()), 'nfmc_philox_normals_f32')
            x, ld = self.bijection.inverse(z)
            return (x, self._base_log_prob(z) - ld) if return_log_prob else x
        st, _keep = self.bijection.packed(dev)
        x = torch.empty(n, self.bijection.d, dtype=torch.float32, device=dev)
        lq = torch.empty(n, dtype=torch.float32, device=dev) if return_log_prob else None
        hip.check(hip.lib().nfmc_realnvp_inverse_f32(C.byref(st), None, n, hip.ptr(x), None, hip.ptr(lq),
                                                     C.byref(rng), hip.stream()), 'nfmc_realnvp_inverse_f32')
        x = x.reshape(n, *self.event_shape)
        return (x, lq) if return_log_prob else x

    def fit(self, x_train, x_val=None, **kwargs):
        from .flow_training import fit as _fit
        return _fit(self, x_train, x_val, **kwargs)

    def variational_fit(self, log_prob_fn, **kwargs):
        from .flow_training import variational_fit as _vfit
        return _vfit(self, log_prob_fn, **kwargs)
