"""nfmc_amd: MI355X-native drop-in for the hot path of davidnabergoj/nfmc.

    from nfmc_amd import sample                       # == `from nfmc import sample` (nfmc/__init__.py:1)
    out = sample(target, event_shape=(64,), strategy='jump_mala', flow='realnvp', n_chains=65536)

The chain state, the flow and all statistics live on the GPU; the per-transition work runs in the
hand-written gfx950 kernels of libnfmc_hip.so (include/nfmc_hip.h).  There is no CPU fallback.
"""
from .sample import create_sampler, sample  # noqa: F401

__version__ = '0.1.0'
