"""Shared host logic of the samplers: target resolution, run state on the device, noise bookkeeping."""
import ctypes as C
import math
import time
from typing import Optional

import torch

from .. import hip
from ..potentials import Potential, recognize


def resolve_target(target, event_shape, fuse='auto', x0=None) -> Optional[Potential]:
    """A closed-form descriptor for `target`, or None (-> split path: torch autograd for U, grad U).
    fuse: 'auto' (default) probes plain callables (potentials.recognize; `x0` sets the radii the probes must also
    cover); False / 'never' keeps every plain callable on the split path."""
    if isinstance(target, Potential):
        return target
    if fuse in (False, 'never'):
        return None
    x_scale = None
    if x0 is not None and x0.numel() > 0:
        x_scale = float(x0.detach().abs().max())
    return recognize(target, event_shape, x_scale=x_scale)


class Replay:
    """Recorded noise for parity tests: `normals` (T, n, d) and `uniforms` (T, n) on the device, consumed in
    transition order (one normal field and one uniform vector per transition; unadjusted transitions take
    no uniforms, exactly like the reference's draw order -- SURVEY.md App. A.3)."""

    def __init__(self, normals, uniforms, device):
        self.normals = torch.as_tensor(normals, dtype=torch.float32).to(device).contiguous()
        self.normals = self.normals.reshape(self.normals.shape[0], self.normals.shape[1], -1)
        self.uniforms = None
        if uniforms is not None and torch.as_tensor(uniforms).numel() > 0:
            self.uniforms = torch.as_tensor(uniforms, dtype=torch.float32).to(device).contiguous()
        self.i_n = 0
        self.i_u = 0

    def take_uniforms(self, k=1):
        """The uniforms of k transitions alone.  The split path draws them only once every target call of the step
        has returned, like the reference (langevin.py:106, hmc.py:112, jump.py:225): a step whose target raised never
        consumes its uniforms."""
        un = self.uniforms[self.i_u:self.i_u + k]
        assert un.shape[0] == k, 'replay uniforms exhausted'
        self.i_u += k
        return un

    def take(self, k, with_uniforms=True):
        nz = self.normals[self.i_n:self.i_n + k]
        assert nz.shape[0] == k, 'replay noise exhausted'
        self.i_n += k
        un = None
        if with_uniforms:
            un = self.uniforms[self.i_u:self.i_u + k]
            assert un.shape[0] == k, 'replay uniforms exhausted'
            self.i_u += k
        return nz, un


class Run:
    """Device-side state of one `sample()` call."""

    def __init__(self, sampler, x0):
        self.dev = hip.require_gpu()
        hip.lib()
        self.n = int(x0.shape[0])
        self.event_shape = tuple(x0.shape[1:])
        self.d = int(math.prod(self.event_shape)) if self.event_shape else 1
        shard = sampler.shard
        self.shard = shard
        if shard is not None:
            lo, hi = shard.bounds(self.n)
            x0 = x0[lo:hi]
            self.chain_offset = lo
            self.n_global = self.n
            self.n = hi - lo
        else:
            self.chain_offset = 0
            self.n_global = self.n
        self.x = x0.detach().to(self.dev, torch.float32).reshape(self.n, self.d).contiguous().clone()
        self.stats = hip.DeviceStats(self.d, self.dev)
        if sampler.seed is None:
            seed = int(torch.randint(0, 2 ** 62, ()).item())
            if shard is not None:
                seed = shard.broadcast_int(seed)
        else:
            seed = int(sampler.seed)
        self.seed = seed
        # 10 (default) or 7: the opt-in Philox4x32-7 stream (NfmcRng.rounds).  Kernels without it answer
        # NFMC_EUNSUPPORTED, which surfaces as a ValueError: no launch silently falls back to the other stream.
        self.rounds = int(getattr(sampler, 'rng_rounds', 10) or 10)
        if self.rounds not in (7, 10):
            raise ValueError('rng_rounds must be 10 (Philox4x32-10) or 7 (Philox4x32-7)')
        self.replay = None
        if sampler.replay is not None:
            normals, uniforms = sampler.replay
            self.replay = Replay(normals, uniforms, self.dev)
        self.t0 = time.time()
        self.kernel_events = None   # bench.py: list of (label, start_event, end_event) when kernel timing is on
        tk = getattr(sampler, 'time_kernels', False)
        self.kernel_event_filter = tk if isinstance(tk, str) else None   # a label: time only that kernel
        if tk:
            self.kernel_events = []

    def timed(self, label):
        """Context manager recording HIP events on the launch stream around one kernel call."""
        return _Timed(self, label)

    def rng(self, step0, k=0, adjusted=True):
        """NfmcRng for a launch of k transitions starting at transition `step0`."""
        if self.replay is not None:
            nz, un = self.replay.take(k, with_uniforms=adjusted)
            self._keep = (nz, un)
            return hip.make_rng(self.seed, self.chain_offset, step0, nz, un, rounds=self.rounds)
        return hip.make_rng(self.seed, self.chain_offset, step0, rounds=self.rounds)

    def sync(self):
        torch.cuda.synchronize(self.dev)

    def time_is_up(self, t0, limit_seconds) -> bool:
        """The `time_limit_seconds` test of the sampling loops (mcmc/base.py:70-72).  With sharded chains rank 0's clock
        decides for everybody, so all ranks stop after the same number of steps (their later collectives -- the refit
        all-gather, the statistics all-reduce -- must see the same shapes)."""
        if limit_seconds is None:
            return False
        self.sync()
        up = time.time() - t0 >= limit_seconds
        if self.shard is not None and self.shard.world > 1:
            up = bool(self.shard.broadcast_int(1 if up else 0))
        return up

    def elapsed(self):
        self.sync()
        return time.time() - self.t0


class _Timed:
    def __init__(self, run, label):
        self.run, self.label = run, label

    def _on(self):
        sel = getattr(self.run, 'kernel_event_filter', None)
        return self.run.kernel_events is not None and (sel is None or sel == self.label)

    def __enter__(self):
        if self._on():
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self._on():
            self.e1.record(torch.cuda.current_stream())
            self.run.kernel_events.append((self.label, self.e0, self.e1))


def chunks(total, limit=hip.MAX_STEPS_PER_CALL):
    done = 0
    while done < total:
        k = min(limit, total - done)
        yield done, k
        done += k


def imd_tensor(kernel, dev):
    """inv_mass_diag on the device, or None when it is all ones (the kernels' scalar fast path)."""
    imd = kernel.inv_mass_diag
    if imd is None or bool((imd == 1).all()):
        return None
    return imd.detach().to(dev, torch.float32).contiguous()


class _NoBar:
    """Stands in for a disabled tqdm bar (constructing one costs ~15 us per sample() call even with disable=True)."""

    def __init__(self, iterable=None):
        self._it = iterable

    def __iter__(self):
        return iter(self._it)

    def update(self, n=1):
        pass

    def close(self):
        pass

    def set_postfix_str(self, s):
        pass


def progress(show, iterable=None, **kw):
    """tqdm when a progress bar is wanted, else a no-op with the same few methods."""
    if not show:
        return _NoBar(iterable)
    from tqdm import tqdm
    return tqdm(iterable, **kw) if iterable is not None else tqdm(**kw)
