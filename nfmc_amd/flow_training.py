"""Flow training used by the warmup paths next to the hot path:
`Flow.fit` (maximum likelihood on MCMC samples: jump.py:139-151, imh.py:171-175) and
`Flow.variational_fit` (reverse-KL to the target: imh.py:67-72, neutra.py:84-91), with the keyword
surface nfmc passes (`early_stopping`, `early_stopping_threshold`, `keep_best_weights`, `batch_size`,
`n_epochs`, `lr`, `n_samples`, `check_for_divergences`, `show_progress`, `time_limit_seconds`) and the
`ValueError`-on-divergence contract the callers catch (jump.py:150, imh.py:174).

`fit` of a RealNVP / NICE flow with a narrow conditioner (n_hidden <= 32, d <= 256: every default flow) runs on the device:
`DeviceFit` drives `nfmc_flow_fit_step_f32` (csrc/fit_kernels.hip: hand-written reverse sweep with weight gradients +
fused AdamW), two launches per epoch; so does `variational_fit` when the caller knows the target as a closed-form potential
(`potential=`, which the samplers pass).  Everything else -- spline couplings, wide conditioners, mini-batches, arbitrary
Python targets -- evaluates the same spec with differentiable torch ops (on the GPU when there is one).
None of it is on the sampling path: `sample()` never calls into it unless `warmup=True` / `fit_nf=True`.
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


def _bump_versions(params):
    """A raw kernel wrote into these tensors' memory: advance their autograd version counters, as an in-place torch op
    would have, so that everything keyed on `_version` (RealNVP._version_key: the weight-blob caches) sees the change.
    False when this torch build has no way to do it without one launch per tensor."""
    setter = getattr(torch._C._autograd, '_unsafe_set_version_counter', None)
    if setter is None:
        return False
    try:
        setter(tuple(params), tuple(int(p._version) + 1 for p in params))
        return True
    except Exception:
        return False


class DeviceFit:
    """State of a flow fit on the device (include/nfmc_hip.h: NfmcFlowFit, csrc/fit_kernels.hip, csrc/fit_rows.hpp): the
    flow's parameters as ONE trainable vector in the layout of its weight blob, AdamW moments, per-workgroup partial
    gradients, the run's bookkeeping state and best weights.
      `step(x)` / `step_variational(z, pot)`   one optimiser step (gradient kernel + fold / AdamW kernel); `status` holds the
                                               batch loss before the step and whether it was applied;
      `run_calls(...)`                         calls of a whole epoch loop enqueued without reading anything back per epoch:
                                               best loss, best weights, early stopping and divergence are decided in the
                                               fold kernel (nfmc_flow_fit_epochs_f32);
      `write_back()`                           the vector into the nn.Parameters (one launch) and the flow's pack cache.
    `DeviceFit.of(bijection, ...)` keeps one instance per flow alive across fits (the refit of jump.py:193-201 runs every
    outer iteration): no allocation, and no gather of the parameters when the vector is still what the flow's kernels use."""

    def __init__(self, bijection, device, n_rows, lr):
        import ctypes as C
        from . import hip
        self.C, self.hip, self.bij, self.dev = C, hip, bijection, device
        lib = hip.lib()
        self.d = d = bijection.d
        # the width the flow's kernels are presented with by default (flows.RealNVP.default_min_hidden: a conditioner 9..32 wide
        # at d = 64 / 128 runs zero-padded to 64 on the matrix cores; the padding's weights are zero, get zero gradients and
        # stay zero): the fit trains THAT blob, on the matrix-core fit kernel, and it is what the sampling kernels read
        self.min_hidden = int(bijection.default_min_hidden())
        self.H_true = bijection.n_hidden
        self.H, self.nhl = max(bijection.n_hidden, self.min_hidden), bijection.n_hidden_layers
        self.hp = int(lib.nfmc_realnvp_padded_hidden(self.H))
        # layer stride rounded to 16 bytes: the row-per-wave kernel reads the staged blob with 16-byte LDS loads
        self.layer_stride = (int(lib.nfmc_coupling_layer_floats(d, self.H, self.nhl, 0)) + 3) // 4 * 4
        self.d4 = (d + 3) // 4 * 4
        self.ea_off = (max(1, bijection.n_coupling * self.layer_stride) + 3) // 4 * 4
        self.n_params = self.ea_off + 4 * self.d4
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=device)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.wide = self.hp > 32      # matrix-core layout: every matrix in both orientations (csrc/mfma_device.hpp, fit_mfma.hip)
        self.partial, self.scratch = None, None
        self.status = torch.zeros(3, dtype=torch.float32, device=device)
        self.prev = torch.zeros_like(self.params)     # the parameters before the latest step (what its validation loss is of)
        self.best = torch.zeros_like(self.params)     # runs: the weights of the best monitored loss
        self.run_state = torch.zeros(2 * hip.FIT_STATE_FLOATS, dtype=torch.float32, device=device)
        self.flow_struct = self._struct(self.params)
        self.fit = hip.NfmcFlowFit(self.flow_struct, hip.ptr(self.params), hip.ptr(self.m), hip.ptr(self.v), self.n_params,
                                   self.ea_off, None, 0, hip.ptr(self.status), None, 0, hip.ptr(self.prev),
                                   hip.ptr(self.best), hip.ptr(self.run_state), None, 0)
        self.opt = hip.NfmcAdamW(float(lr), 0.9, 0.999, 1e-8, 0.01, 0)   # torch.optim.AdamW defaults
        self.n_rows = 0
        self.ensure_rows(n_rows)
        self._pieces = None
        self._xv = None
        # the trainable vector, gathered from the nn.Parameters ON THE DEVICE (RealNVP.packed goes through the host)
        self._scatter(self.params, to_vector=True)

    def ensure_rows(self, n_rows):
        """Workspace for fits of up to `n_rows` rows (batch + validation): the per-workgroup partial-gradient slabs and, for
        wide conditioners, the resident waves' activation checkpoints (nfmc_flow_fit_workspace).  Grows, never shrinks."""
        n_rows = max(int(n_rows), 1)
        if n_rows <= self.n_rows:
            return
        C, hip = self.C, self.hip
        nfl = C.c_int64(0)
        sbytes = int(hip.lib().nfmc_flow_fit_workspace(C.byref(self.flow_struct), n_rows, n_rows, self.n_params, C.byref(nfl)))
        nfl = int(nfl.value)
        if self.partial is None or self.partial.numel() < nfl:
            self.partial = torch.zeros(nfl, dtype=torch.float32, device=self.dev)
            self.fit.partial, self.fit.partial_floats = hip.ptr(self.partial), nfl
        if sbytes and (self.scratch is None or self.scratch.numel() * 4 < sbytes):
            self.scratch = torch.empty(sbytes // 4, dtype=torch.float32, device=self.dev)
            self.fit.scratch, self.fit.scratch_bytes = hip.ptr(self.scratch), sbytes
        self.n_rows = n_rows

    @classmethod
    def of(cls, bijection, device, n_rows, lr):
        """The flow's resident fitter (created on first use), ready for a new fit: learning rate set, validation rows
        cleared, the vector equal to the flow's current parameters."""
        fitter = bijection.__dict__.get('_device_fit')
        if (fitter is None or fitter.dev != device or fitter.bij is not bijection or fitter.H_true != bijection.n_hidden
                or fitter.min_hidden != int(bijection.default_min_hidden()) or fitter.nhl != bijection.n_hidden_layers or fitter.d != bijection.d
                or fitter.ea_off < bijection.n_coupling * fitter.layer_stride):
            fitter = cls(bijection, device, n_rows, lr)
            bijection.__dict__['_device_fit'] = fitter
            return fitter
        fitter.opt.lr, fitter.opt.weight_decay, fitter.opt.step = float(lr), 0.01, 0
        fitter.fit.x_val, fitter.fit.n_val, fitter._xv = None, 0, None
        fitter.ensure_rows(n_rows)
        if not fitter._vector_is_current():
            fitter._scatter(fitter.params, to_vector=True)
        return fitter

    def _vector_is_current(self):
        """True when the flow's pack cache still holds THIS vector for the parameters as they are now: nothing touched the
        nn.Parameters since the last write_back, so there is nothing to gather."""
        cache = self.bij._pack_cache if isinstance(self.bij._pack_cache, dict) else {}
        hit = cache.get(self.min_hidden)
        return bool(hit is not None and hit[0] == self.bij._version_key(self.dev) + (self.min_hidden,) and hit[1][1]
                    and hit[1][1][0] is self.params)

    def _struct(self, vec):
        hip, o, d4, bij = self.hip, self.ea_off, self.d4, self.bij
        view = lambda k: hip.ptr(vec[o + k * d4:o + k * d4 + self.d])
        return hip.NfmcRealNVP(self.d, bij.n_coupling, self.H, self.nhl, float(bij.min_scale), 0, view(0), view(1), view(2),
                               view(3), hip.ptr(vec), self.layer_stride, float(bij.spline_bound), 0)

    def _layout(self, bijection):
        """[(parameter tensor, offset in the vector, rows, cols, vector stride of a row, of a column)] in the VALU blob
        layout (include/nfmc_hip.h: W1T (d_a, HP) | b1 | [WhT (HP, HP) | bh] | W3 (2 d_b, HP) | b3 per coupling layer, then
        the four ElementwiseAffine vectors at ea_off)."""
        H, hp, d = self.H_true, self.hp, self.d
        d_a, d_b = d // 2, d - d // 2
        out = []
        if self.wide:
            # W1 (HP, d_a) | W1T (d_a, HP) | b1 | [Wh (HP, HP) | WhT | bh] | W3 (2 d_b, HP) | W3T (HP, 2 d_b) | b3: every Linear
            # weight (out, in) appears as itself and transposed; gathering fills both, scattering reads both (they are equal)
            for li, cpl in enumerate(bijection.couplings):
                lin = list(cpl.conditioner)
                cur = li * self.layer_stride
                mats = [(lin[0], hp, d_a)] + [(l, hp, hp) for l in lin[1:-1]] + [(lin[-1], 2 * d_b, hp)]
                for l, rows_p, cols_p in mats:
                    r, c = l.weight.shape
                    out.append((l.weight, cur, r, c, cols_p, 1))
                    cur += rows_p * cols_p
                    out.append((l.weight, cur, r, c, 1, rows_p))
                    cur += rows_p * cols_p
                    out.append((l.bias, cur, 1, r, 0, 1))
                    cur += rows_p
        for li, cpl in enumerate(bijection.couplings if not self.wide else []):
            lin = list(cpl.conditioner)
            cur = li * self.layer_stride
            out.append((lin[0].weight, cur, H, d_a, 1, hp))             # Linear weight (H, d_a) <-> W1T (d_a, HP)
            cur += d_a * hp
            out.append((lin[0].bias, cur, 1, H, 0, 1))
            cur += hp
            for l in lin[1:-1]:
                out.append((l.weight, cur, H, H, 1, hp))                # (out, in) <-> WhT (in, out)
                cur += hp * hp
                out.append((l.bias, cur, 1, H, 0, 1))
                cur += hp
            out.append((lin[-1].weight, cur, 2 * d_b, H, hp, 1))        # W3 keeps (out, in)
            cur += 2 * d_b * hp
            out.append((lin[-1].bias, cur, 1, 2 * d_b, 0, 1))
        ea0, ea1 = bijection.layers[0], bijection.layers[-1]
        for k, t in enumerate((ea0.log_scale, ea0.shift, ea1.log_scale, ea1.shift)):
            out.append((t, self.ea_off + k * self.d4, 1, d, 0, 1))
        return out

    def _scatter(self, vec, to_vector, bijection=None):
        """nn.Parameters <-> trainable vector in one launch (nfmc_flow_blob_copy_f32); parameters that are not plain
        contiguous float32 tensors on the vector's device take slice copies."""
        bij = self.bij if bijection is None else bijection
        hip = self.hip
        layout = [t for t in self._layout(bij) if t[0].numel() > 0]   # d = 1: the source half is empty, W1 has no entries
        plain = (vec.dtype == torch.float32 and vec.is_contiguous()
                 and all(p.dtype == torch.float32 and p.device == vec.device and p.is_contiguous() for p, *_ in layout))
        with torch.no_grad():
            if plain:
                key = tuple(p.data_ptr() for p, *_ in layout)
                if bijection is None and self._pieces is not None and self._pieces[0] == key:
                    arr = self._pieces[1]
                else:
                    arr = (hip.NfmcBlobPiece * len(layout))(*[
                        hip.NfmcBlobPiece(hip.ptr(p), off, rows, cols, rs, cs) for p, off, rows, cols, rs, cs in layout])
                    if bijection is None:
                        self._pieces = (key, arr)
                if to_vector or _bump_versions([p for p, *_ in layout]):
                    hip.check(hip.lib().nfmc_flow_blob_copy_f32(hip.ptr(vec), arr, len(layout), 1 if to_vector else 0,
                                                                hip.stream()), 'nfmc_flow_blob_copy_f32')
                    return
            for p, off, rows, cols, rs, cs in layout:
                if rows == 1:
                    view = vec[off:off + cols]
                elif rs == 1:      # stored transposed
                    view = vec[off:off + cols * cs].view(cols, cs)[:, :rows].t()
                else:
                    view = vec[off:off + rows * rs].view(rows, rs)[:, :cols]
                if to_vector:
                    view.copy_(p.detach().to(view))
                else:
                    p.copy_(view)

    @staticmethod
    def supported(bijection, device) -> bool:
        from . import hip
        from .flows import RealNVP
        import ctypes as C
        if not isinstance(bijection, RealNVP) or bijection.n_bins != 0 or device.type != 'cuda':
            return False
        if bijection.n_coupling < 1 or any(p.device != device for p in bijection.parameters()):
            return False
        st = hip.NfmcRealNVP(bijection.d, bijection.n_coupling, bijection.n_hidden, bijection.n_hidden_layers,
                             float(bijection.min_scale), 0, None, None, None, None, None, 0, 0.0, 0)   # dimensions only
        return bool(hip.lib().nfmc_flow_fit_supported_f32(C.byref(st)))

    def set_validation(self, xv):
        """Validation rows (n_val, d) float32 on the device: every maximum-likelihood step then also reports their mean NLL
        at the parameters it started from (status[2]), from the same launch."""
        self._xv = xv
        self.fit.x_val = self.hip.ptr(xv)
        self.fit.n_val = int(xv.shape[0])

    def step(self, x, applied_steps, lr=None):
        """One AdamW step on the mean NLL of the rows x (n_rows, d) float32 on the device (lr = 0: evaluate only)."""
        if lr is not None:
            self.opt.lr = float(lr)
            self.opt.weight_decay = 0.01 if lr > 0 else 0.0
        self.opt.step = int(applied_steps) + 1
        self.hip.check(self.hip.lib().nfmc_flow_fit_step_f32(self.C.byref(self.fit), self.hip.ptr(x), int(x.shape[0]),
                                                             self.C.byref(self.opt), self.hip.stream()), 'nfmc_flow_fit_step_f32')

    def step_variational(self, z, pot_struct, applied_steps):
        """One AdamW step on the reverse-KL estimate mean[log q(x) - log p(x)], x = f^-1(z), for latents z (n, d) on the
        device and a closed-form potential descriptor (hip.NfmcPotential) as -log p."""
        self.opt.step = int(applied_steps) + 1
        self.hip.check(self.hip.lib().nfmc_flow_variational_fit_step_f32(
            self.C.byref(self.fit), self.C.byref(pot_struct), self.hip.ptr(z), int(z.shape[0]), self.C.byref(self.opt),
            self.hip.stream()), 'nfmc_flow_variational_fit_step_f32')

    def control(self, n_epochs, early_stopping, early_stopping_threshold, keep_best_weights, skip_nonfinite=False):
        return self.hip.NfmcFitControl(int(n_epochs), int(bool(early_stopping)), int(early_stopping_threshold),
                                       int(bool(keep_best_weights)), int(bool(skip_nonfinite)), 0)

    def run_calls(self, ctl, x, first_call, n_calls, pot_struct=None, epoch_stride=0):
        """Enqueue calls first_call .. first_call + n_calls - 1 of a run on rows x (n, d) (nothing is read back)."""
        self.hip.check(self.hip.lib().nfmc_flow_fit_epochs_f32(
            self.C.byref(self.fit), self.C.byref(pot_struct) if pot_struct is not None else None, self.hip.ptr(x),
            int(x.shape[-2]), int(epoch_stride), self.C.byref(self.opt), self.C.byref(ctl), int(first_call), int(n_calls),
            self.hip.stream()), 'nfmc_flow_fit_epochs_f32')

    def state_after(self, n_calls_done):
        """The run's bookkeeping state after `n_calls_done` calls, as a list of floats (ONE device-to-host copy)."""
        half = (n_calls_done & 1) * self.hip.FIT_STATE_FLOATS
        return self.run_state[half:half + self.hip.FIT_STATE_FLOATS].tolist()

    def state_later(self, n_calls_done):
        """The same without stalling the stream: the state is copied to pinned host memory behind the run's kernels and an
        event recorded; `PendingFit.result()` waits for THAT event only (kernels enqueued after the run keep the GPU busy)."""
        half = (n_calls_done & 1) * self.hip.FIT_STATE_FLOATS
        if self.__dict__.get('_pinned') is None:
            self._pinned = [torch.empty(self.hip.FIT_STATE_FLOATS, dtype=torch.float32).pin_memory() for _ in range(2)]
            self._pin_turn = 0
        self._pin_turn ^= 1
        host = self._pinned[self._pin_turn]
        host.copy_(self.run_state[half:half + self.hip.FIT_STATE_FLOATS], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return PendingFit(host, ev)

    def nll(self, x):
        """Mean NLL of rows x under the CURRENT trainable vector (forward kernel), as a device scalar."""
        n = int(x.shape[0])
        lp = torch.empty(n, dtype=torch.float32, device=self.dev)
        self.hip.check(self.hip.lib().nfmc_realnvp_forward_f32(self.C.byref(self.flow_struct), self.hip.ptr(x), n, None, None,
                                                               self.hip.ptr(lp), self.hip.stream()), 'nfmc_realnvp_forward_f32')
        return -lp.mean()

    def write_back(self, vec=None, bijection=None):
        """The trainable vector (or a saved copy of it) into the flow's nn.Parameters, and -- when it goes into the fitter's
        own flow -- straight into that flow's pack cache: the sampling kernels' next launch takes the trained vector as its
        weight blob without a trip through the host (RealNVP.packed would rebuild it from the parameters)."""
        vec = self.params if vec is None else vec
        if bijection is None and vec is not self.params and vec.dtype == torch.float32:
            self.params.copy_(vec)       # the resident vector stays the one the kernels (and the next fit) use
            vec = self.params
        self._scatter(vec, to_vector=False, bijection=bijection)
        if bijection is None and vec.dtype == torch.float32:
            bij, o, d4 = self.bij, self.ea_off, self.d4
            keep = [vec] + [vec[o + k * d4:o + k * d4 + self.d] for k in range(4)]
            # every other presentation of the weights (e.g. zero-padded to the matrix-core width) is stale now
            bij._pack_cache = {self.min_hidden: (bij._version_key(self.dev) + (self.min_hidden,), (self._struct(vec), keep))}


class PendingFit:
    """Outcome of a device fit whose check was deferred (`Flow.fit(..., defer_check=True)`): `result()` waits for the fit's
    own kernels (not for what was enqueued after them), raises the ValueError of a diverged run (jump.py:150) and returns
    the best monitored loss."""

    def __init__(self, host_state=None, event=None, state=None):
        self._host, self._event, self._state = host_state, event, state

    def result(self):
        from . import hip
        if self._state is None:
            if self._event is not None:
                self._event.synchronize()
            self._state = self._host.tolist() if self._host is not None else None
            self._host = self._event = None
        st = self._state
        if st is not None and st[hip.FIT_DIVERGED]:
            raise ValueError('flow training diverged (non-finite loss)')
        return st[hip.FIT_BEST_LOSS] if st is not None else math.inf


def _run_chunk(time_limit_seconds, early_stopping):
    """Calls enqueued between two looks at the run's state (one 32-byte read each): everything at once when nothing can
    end the run early, 64 epochs when early stopping can (calls after the end are no-ops, but they are still launches), 16
    under a time limit (an epoch is tens of microseconds)."""
    if time_limit_seconds is not None:
        return 16
    return 64 if early_stopping else 1 << 30


def _fit_device(flow, xt, xv, n_epochs, lr, early_stopping, early_stopping_threshold, keep_best_weights, time_limit_seconds,
                defer=False):
    """`_loop` for the full-batch maximum-likelihood fit, on the device end to end.  Same order of events per epoch e --
    loss at the weights w_e, step unless it is not finite, validation at the new weights w_{e+1}, best-so-far / early
    stopping bookkeeping -- but decided in the fold kernel (csrc/fit_kernels.hip: fit_fold_kernel): call c reports the batch
    loss AND the validation loss at w_c (the validation rows ride in the gradient launch), so the bookkeeping of epoch
    c - 1 happens inside call c, and one extra evaluate-only call closes the last epoch.  The host enqueues the calls and
    reads 32 bytes at the end (every 64 epochs with early stopping, every 16 under a time limit)."""
    fitter = DeviceFit.of(flow.bijection, xt.device, xt.shape[0] + (xv.shape[0] if xv is not None else 0), lr)
    hip = fitter.hip
    if xv is not None:
        fitter.set_validation(xv)
    n_epochs = int(n_epochs)
    ctl = fitter.control(n_epochs, early_stopping, early_stopping_threshold, keep_best_weights)
    total = n_epochs + (1 if xv is not None else 0)
    if defer and time_limit_seconds is None and not early_stopping and total > 0:
        # nothing can end the run early: every call and the write-back are enqueued, the state follows the stream to pinned
        # memory, and the caller checks it when it next needs to (the refit of jump.py:193-201: at the next refit)
        fitter.run_calls(ctl, xt, 0, total)
        fitter.write_back(fitter.best if keep_best_weights else None)
        return fitter.state_later(total)
    t0 = time.time()
    done, st = 0, None
    try:
        while done < total:
            if time_limit_seconds is not None and done > 0 and time.time() - t0 >= time_limit_seconds:
                if xv is not None and ctl.n_epochs > done:       # close the last applied epoch: one evaluate-only call
                    ctl.n_epochs = done
                    fitter.run_calls(ctl, xt, done, 1)
                    done += 1
                break
            k = min(_run_chunk(time_limit_seconds, early_stopping), total - done)
            fitter.run_calls(ctl, xt, done, k)
            done += k
            if done < total:
                st = fitter.state_after(done)
                if st[hip.FIT_STOPPED] or st[hip.FIT_DIVERGED]:
                    break
        st = fitter.state_after(done) if done else None
        if st is not None and st[hip.FIT_DIVERGED]:
            raise ValueError('flow training diverged (non-finite loss)')
    finally:
        # whatever happened, the nn.Parameters end up as the best weights seen (or the last ones): callers that catch the
        # ValueError restore their own saved state_dict on top (jump.py:150-151).  Without best-weights bookkeeping the
        # weights after the last epoch that counted are the current vector (an early stop applies no further step).
        fitter.write_back(fitter.best if (keep_best_weights and done) else None)
    return st[hip.FIT_BEST_LOSS] if st is not None else math.inf


def fit(flow, x_train, x_val=None, n_epochs: int = 500, lr: float = 0.05, batch_size='adaptive',
        shuffle: bool = True, show_progress: bool = False, keep_best_weights: bool = True,
        early_stopping: bool = False, early_stopping_threshold: int = 50, time_limit_seconds=None, defer_check: bool = False,
        **_ignored):
    """Maximum-likelihood fit: minimise -mean log q(x_train).
    `defer_check` (beyond the reference's keywords): on the device path, return a `PendingFit` instead of waiting for the
    run -- its `result()` raises the ValueError of a diverged fit; the samplers' per-iteration refit uses it so that the GPU
    is never idle behind a fit."""
    import os
    dev = _train_device(flow)
    resident = flow.bijection.__dict__.get('_device_fit')
    first_param = next(flow.parameters(), None)
    resident = resident is not None and resident.dev == dev and first_param is not None and first_param.device == dev
    if not resident:       # a flow that was fitted here before is on the device already (checked again by the fitter)
        flow.to(dev)
    xt = x_train.detach().to(dev, torch.float32).reshape(x_train.shape[0], -1)
    xv = x_val.detach().to(dev, torch.float32).reshape(x_val.shape[0], -1) if x_val is not None and len(x_val) else None
    n = xt.shape[0]
    if n == 0:
        return PendingFit() if defer_check else None
    bs = n if batch_size == 'adaptive' or batch_size is None else max(1, min(int(batch_size), n))
    if (bs >= n and os.environ.get('NFMC_FIT_TORCH') != '1' and (resident or DeviceFit.supported(flow.bijection, dev))):
        # full-batch fit of a RealNVP the fit kernels cover: every epoch is two launches of libnfmc_hip (fit_kernels.hip)
        out = _fit_device(flow, xt.contiguous(), xv.contiguous() if xv is not None else None, n_epochs, lr, early_stopping,
                          early_stopping_threshold, keep_best_weights, time_limit_seconds, defer=defer_check)
        return out if defer_check else None
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
    return PendingFit() if defer_check else None


def _variational_fit_device(flow, potential, dev, n_epochs, lr, n_samples, early_stopping, early_stopping_threshold,
                            keep_best_weights, check_for_divergences, time_limit_seconds):
    """`_loop` for the reverse-KL fit, on the device end to end: each epoch draws its latents with torch.randn on the
    device (the global CUDA generator, as the torch path does) and enqueues ONE call of the run; best loss, best weights,
    early stopping and the handling of a non-finite epoch (skipped, or the end of the run when `check_for_divergences`)
    are decided in the fold kernel.  The host looks at the run's state every 64 epochs (16 under a time limit)."""
    d = flow.bijection.d
    fitter = DeviceFit.of(flow.bijection, dev, n_samples, lr)
    hip = fitter.hip
    pot = potential.descriptor(dev)
    n_epochs = int(n_epochs)
    ctl = fitter.control(n_epochs, early_stopping, early_stopping_threshold, keep_best_weights,
                         skip_nonfinite=not check_for_divergences)
    look = 16 if time_limit_seconds is not None else 64
    t0 = time.time()
    done, st = 0, None
    try:
        while done < n_epochs:
            if time_limit_seconds is not None and time.time() - t0 >= time_limit_seconds:
                break
            z = torch.randn(n_samples, d, device=dev)
            fitter.run_calls(ctl, z, done, 1, pot_struct=pot)
            done += 1
            if done % look == 0 and done < n_epochs:
                st = fitter.state_after(done)
                if st[hip.FIT_STOPPED] or st[hip.FIT_DIVERGED]:
                    break
        st = fitter.state_after(done) if done else None
        if st is not None and st[hip.FIT_DIVERGED]:
            raise ValueError('flow training diverged (non-finite loss)')
    finally:
        fitter.write_back(fitter.best if (keep_best_weights and done) else None)
    return st[hip.FIT_BEST_LOSS] if st is not None else math.inf


def variational_fit(flow, log_prob_fn, n_epochs: int = 500, lr: float = 0.05, n_samples: int = 1000,
                    early_stopping: bool = False, early_stopping_threshold: int = 50, keep_best_weights: bool = True,
                    check_for_divergences: bool = False, show_progress: bool = False, time_limit_seconds=None,
                    potential=None, **_ignored):
    """Reverse-KL fit: minimise E_{z~N(0,I)} [ log q(x) - log p(x) ], x = f^-1(z).
    `potential` (beyond the reference's keywords): the closed-form descriptor of -log p (nfmc_amd.potentials) when the
    caller has one -- the samplers pass what `resolve_target` found; the step then runs on the device
    (nfmc_flow_variational_fit_step_f32) for flows the fit kernels cover."""
    dev = _train_device(flow)
    flow.to(dev)
    d = flow.bijection.d
    event_shape = flow.event_shape
    n_samples = max(int(n_samples), 1)
    import os
    if (potential is not None and os.environ.get('NFMC_FIT_TORCH') != '1' and hasattr(potential, 'descriptor')
            and int(getattr(potential, 'event_size', -1)) == d and DeviceFit.supported(flow.bijection, dev)):
        _variational_fit_device(flow, potential, dev, n_epochs, lr, n_samples, early_stopping, early_stopping_threshold,
                                keep_best_weights, check_for_divergences, time_limit_seconds)
        return

    def loss_fn():
        z = torch.randn(n_samples, d, device=dev)
        x, ld = inverse_torch(flow.bijection, z)
        log_q = _base_log_prob(z) - ld
        log_p = log_prob_fn(x.reshape(n_samples, *event_shape)).reshape(-1)
        return (log_q - log_p.to(log_q)).mean()

    _loop(flow, loss_fn, None, n_epochs, lr, early_stopping, early_stopping_threshold, keep_best_weights,
          show_progress, time_limit_seconds, check_for_divergences=check_for_divergences)
