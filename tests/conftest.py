import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """The C-ABI library is a build artefact (git-ignored): compile it when it is missing or older than the
    sources (a no-op otherwise; hipcc cross-compiles gfx950 without a GPU).  Building is not a fallback: the
    product still refuses to run without the library."""
    if os.environ.get('NFMC_LIB'):
        return
    try:
        from nfmc_amd import build as b
        b.build(force=False, verbose=False)
    except Exception as e:   # no hipcc here: the tests that need the library will say so themselves
        sys.stderr.write('conftest: could not build libnfmc_hip.so: %s\n' % e)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        return {k: z[k] for k in z.files}


def golden_flow(fx, d, n_layers=2, n_hidden=None, cond_layers=2):
    """Rebuild the oracle (CPU) flow a fixture was generated with."""
    from oracle import flow as oflow
    ck = {'n_layers': cond_layers}
    if n_hidden is not None:
        ck['n_hidden'] = n_hidden
    f = oflow.Flow(oflow.RealNVP((d,), n_layers=n_layers, conditioner_kwargs=ck))
    sd = {k[len('flow/'):]: torch.from_numpy(v) for k, v in fx.items() if k.startswith('flow/')}
    f.load_state_dict(sd)
    return f


class FailingTarget:
    """A target that raises ValueError on its `fail_calls`-th calls (1-based): the driver of the reference's failure
    channel in the *_fail_* fixtures (tests/golden/make_golden.py uses the same counting)."""

    def __init__(self, base, fail_calls):
        self.base, self.fail_calls, self.calls = base, set(int(c) for c in fail_calls), 0

    def __call__(self, x):
        self.calls += 1
        if self.calls in self.fail_calls:
            raise ValueError('target failed on call %d' % self.calls)
        return self.base(x)


@pytest.fixture
def golden():
    return load_golden
