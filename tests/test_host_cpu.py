"""CPU: host logic of nfmc_amd and the C-ABI library's symbol table (no compute calls without a GPU)."""
import ctypes
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from nfmc_amd import hip
    header = open(os.path.join(ROOT, 'include', 'nfmc_hip.h')).read()
    declared = set(re.findall(r'\b(nfmc_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    handle = hip.lib()
    for name in declared:
        assert hasattr(handle, name), f'{name} declared in include/nfmc_hip.h but not exported'
    assert declared == {s[0] for s in hip.SYMBOLS}, 'hip.SYMBOLS and the header disagree'
    lim = hip.limits()
    assert lim.abi_version == hip.NFMC_ABI_VERSION == 4 and lim.max_d_sampler >= 256 and lim.max_steps_per_call == hip.MAX_STEPS_PER_CALL
    assert hip.lib().nfmc_error_string(-5).decode().startswith('statistics scratch')
    assert hip.lib().nfmc_stats_scratch_bytes(64) > 0 and hip.lib().nfmc_stats_scratch_bytes(5000) == 0
    assert hip.lib().nfmc_realnvp_padded_hidden(5) == 8 and hip.lib().nfmc_realnvp_padded_hidden(100) == 128
    assert hip.lib().nfmc_realnvp_layer_floats(64, 4, 2) == 32 * 4 + 4 + 16 + 4 + 64 * 4 + 64
    # NeuTra scratch of the matrix-core path: momentum + gradient (n, d), U~ + H0 (n,), and one activation-checkpoint
    # area per resident (workgroup slot <= 256, wave) of 24 KB per coupling layer at d = 128, H = 128 x 2 -- bounded in n
    sb = hip.lib().nfmc_neutra_scratch_bytes
    assert sb(1000, 64, 8, 2, 2) == 0                                        # VALU path: no scratch
    assert sb(65536, 128, 128, 2, 2) == 4 * (2 * 65536 * 128 + 2 * 65536 + 256 * 8 * 2 * 24 * 256)
    assert sb(128, 128, 128, 2, 2) == 4 * (2 * 128 * 128 + 2 * 128 + 1 * 8 * 2 * 24 * 256)
    assert sb(10 ** 6, 128, 128, 2, 2) - sb(65536, 128, 128, 2, 2) == 4 * (10 ** 6 - 65536) * (2 * 128 + 2)
    assert sb(65536, 64, 40, 1, 3) == 4 * (2 * 65536 * 64 + 2 * 65536 + 256 * 8 * 3 * (4 + 4) * 256)   # H = 40 -> 64: 4 + 4 tiles


def test_struct_sizes_match_the_header():
    """Compile a tiny C program against include/nfmc_hip.h and compare sizeof() with the ctypes mirrors."""
    from nfmc_amd import hip
    names = ['NfmcPotential', 'NfmcRng', 'NfmcStats', 'NfmcSampleStore', 'NfmcTune', 'NfmcJumpTail', 'NfmcMalaArgs', 'NfmcHmcArgs', 'NfmcRealNVP', 'NfmcFlowMhArgs',
             'NfmcNeutraHmcArgs', 'NfmcSelectArgs', 'NfmcLimits', 'NfmcAdamW', 'NfmcFlowFit', 'NfmcFitControl']
    src = '#include <stdio.h>\n#include "nfmc_hip.h"\nint main(){' + ''.join(
        f'printf("%zu\\n", sizeof({n}));' for n in names) + 'return 0;}'
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, 's.c')
        open(c, 'w').write(src)
        exe = os.path.join(td, 's')
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), c, '-o', exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    for n, sz in zip(names, sizes):
        assert ctypes.sizeof(getattr(hip, n)) == sz, n


def test_a_library_of_another_abi_version_is_refused(tmp_path):
    """`hip.lib()` compares nfmc_limits().abi_version with the version its ctypes structs mirror BEFORE binding any other
    symbol: a stale or side-built library selected through NFMC_LIB would read pointers at the wrong offsets."""
    src = tmp_path / 'fake.c'
    src.write_text('typedef struct {int abi_version, a, b, c, d, e;} L;\n'
                   'int nfmc_limits(L* o) { o->abi_version = 1; o->a = o->b = o->c = o->d = o->e = 0; return 0; }\n')
    so = tmp_path / 'libnfmc_hip.old.so'
    subprocess.check_call(['gcc', '-shared', '-fPIC', str(src), '-o', str(so)])
    code = ('import sys; sys.path.insert(0, %r)\nfrom nfmc_amd import hip\n'
            'try:\n    hip.lib()\nexcept RuntimeError as e:\n    assert "ABI version 1" in str(e), e; print("refused")\n' % ROOT)
    out = subprocess.check_output([sys.executable, '-c', code], env=dict(os.environ, NFMC_LIB=str(so)))
    assert out.decode().strip().endswith('refused')


def test_samplers_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from nfmc_amd import sample
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        sample(lambda x: torch.sum(x ** 2, dim=1), event_shape=(5,), strategy='mala', n_chains=4, n_iterations=2,
               show_progress=False)
    from nfmc_amd.flows import Flow, RealNVP
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        Flow(RealNVP((6,))).log_prob(torch.randn(3, 6))


def test_product_does_not_import_the_oracle():
    for root, _dirs, files in os.walk(os.path.join(ROOT, 'nfmc_amd')):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f


@pytest.mark.parametrize('strategy', ['mala', 'ula', 'hmc', 'uhmc', 'mh', 'imh', 'fixed_imh', 'jump_mala', 'jump_ula',
                                      'jump_hmc', 'jump_uhmc', 'jump_mh', 'neutra_hmc', 'neutra_mh'])
def test_create_sampler_plumbing(strategy):
    """nfmc/sample.py:20-240 keyword plumbing and defaults."""
    from nfmc_amd.sample import create_sampler
    s = create_sampler(lambda x: torch.sum(x ** 2, dim=-1), (10,), strategy=strategy,
                       param_kwargs={'n_iterations': 7, 'store_samples': False})
    assert s.params.n_iterations == 7 and s.params.store_samples is False
    if strategy.startswith('jump_'):
        inner = s.inner_sampler
        assert inner.params.n_iterations == (5 if strategy == 'jump_hmc' else 100)   # sample.py:161-162, base.py:31
        assert s.params.adjusted_jumps and not s.params.fit_nf and s.params.n_jumps_before_training == 10
        assert inner.params.adjustment == (strategy in ('jump_mala', 'jump_hmc', 'jump_mh'))
    if strategy in ('mh', 'jump_mh'):
        k = s.inner_sampler if strategy == 'jump_mh' else s
        assert k.params.imd_adjustment == 1e-5 and k.params.tune_step_size is False   # mh.py:20-25
    if 'mala' in strategy or 'ula' in strategy:
        k = s.inner_sampler.kernel if strategy.startswith('jump') else s.kernel
        assert abs(k.step_size - 10 ** (-1 / 3)) < 1e-12                                  # langevin.py:16-18
        assert torch.equal(k.inv_mass_diag, torch.ones(10))
    if strategy in ('hmc', 'uhmc'):
        assert s.kernel.n_leapfrog_steps == 20 and s.kernel.step_size == 0.01               # hmc.py:13, base.py:109
    if strategy in ('imh', 'fixed_imh', 'neutra_hmc') or strategy.startswith('jump'):
        assert len(s.kernel.flow.bijection.layers) == 6


def test_unsupported_strategy_and_flow_are_errors():
    from nfmc_amd.sample import create_sampler
    with pytest.raises(ValueError, match='Unsupported sampling strategy'):
        create_sampler(lambda x: x.sum(-1), (4,), strategy='dlmc')
    with pytest.raises(ValueError, match="outside this build's path"):
        create_sampler(lambda x: x.sum(-1), (4,), strategy='imh', flow='maf')
    with pytest.raises(ValueError):
        create_sampler(lambda x: x.sum(-1), (4,), strategy='imh', flow=None)


def test_flow_strings():
    """nfmc/util.py:189-215; test/test_flow_kwargs.py."""
    from nfmc_amd.util import parse_flow_string, create_flow_object, metropolis_acceptance_log_ratio
    from conftest import load_golden
    fx = load_golden('util')
    p = parse_flow_string('realnvp%{"n_layers": 10, "conditioner_kwargs": {"n_layers": 5, "n_hidden": 100}}')
    assert p['name'] == 'realnvp' and p['kwargs']['n_layers'] == int(fx['parsed_n_layers'])
    assert p['kwargs']['conditioner_kwargs']['n_hidden'] == int(fx['parsed_n_hidden'])
    assert parse_flow_string('rnvp') == {'name': 'rnvp', 'kwargs': {}, 'hash': hash('rnvp')}
    basic = create_flow_object('realnvp', (100,))
    adv = create_flow_object('real_nvp%{"n_layers": 10, "conditioner_kwargs": {"n_layers": 5, "n_hidden": 100}}', (100,))
    assert len(adv.bijection.layers) > len(basic.bijection.layers)
    assert adv.bijection.n_hidden == 100 and adv.bijection.n_hidden_layers == 5
    assert basic.event_shape == (100,)
    a, b, c, d = (torch.tensor(v) for v in ([1.0, -2.0], [0.5, 3.0], [0.25, 0.0], [-1.0, 4.0]))
    np.testing.assert_allclose(metropolis_acceptance_log_ratio(a, b, c, d).numpy(), fx['log_ratio'])


def test_metropolization_sign_convention():
    """test/test_metropolization.py:30."""
    from nfmc_amd.util import metropolis_acceptance_log_ratio as r
    t = lambda x: torch.sum(x ** 2, dim=-1)
    q = lambda x: torch.sum(x ** 2 / (2 * 100 ** 2), dim=-1)
    x0, x1 = torch.tensor([[-100.0, -100.0]]), torch.tensor([[0.0, 0.0]])
    assert r(-t(x0), -t(x1), -q(x0), -q(x1)) > r(-q(x0), -q(x1), -t(x0), -t(x1))


def test_flow_state_dict_is_interchangeable_with_the_oracle_flow():
    from nfmc_amd.flows import Flow, RealNVP
    from oracle import flow as oflow
    a = Flow(RealNVP((9,), n_layers=3, conditioner_kwargs={'n_hidden': 6, 'n_layers': 3}))
    b = oflow.Flow(oflow.RealNVP((9,), n_layers=3, conditioner_kwargs={'n_hidden': 6, 'n_layers': 3}))
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    a.load_state_dict(b.state_dict())
    assert a.event_shape == (9,) and a.get_device().type == 'cpu'


def test_statistics_equal_the_reference_streaming_formula():
    """MCMCExpectation (sums / n_seen) == the reference's running update (base.py:88-95) == oracle.Moments."""
    from nfmc_amd.containers import MCMCStatistics
    from oracle.samplers import Moments
    torch.manual_seed(0)
    st, mo = MCMCStatistics((5,)), Moments(1)
    for k in (1, 3, 2):
        x = torch.randn(k, 11, 5)
        st.expectations.update(x)
        mo.update(x)
    np.testing.assert_allclose(st.running_first_moment.numpy(), mo.first.numpy(), atol=1e-6)
    np.testing.assert_allclose(st.running_second_moment.numpy(), mo.second.numpy(), atol=1e-6)
    np.testing.assert_allclose(st.running_variance.numpy(), mo.variance.numpy(), atol=1e-6)
    st.update_counters(n_accepted_trajectories=3, n_attempted_trajectories=4)
    assert st.acceptance_rate == 0.75


def test_samples_store_semantics():
    """thinning / max_samples / last_sample of base.py:234-263."""
    from nfmc_amd.containers import MCMCSamples, MCMCOutput
    s = MCMCSamples((2,), thinning=2, max_samples=3)
    xs = torch.arange(7 * 4 * 2, dtype=torch.float32).reshape(7, 4, 2)
    s.add(xs[:3])
    s.add(xs[3])
    s.add(xs[4:])
    kept = [0, 2, 4, 6][-3:]
    assert torch.equal(s.as_tensor(), xs[kept])
    assert torch.equal(s.last_sample, xs[6]) and s.n_samples == 3
    off = MCMCSamples((2,), store_samples=False)
    off.add(xs[:2])
    assert off.n_samples == 0 and torch.equal(off.last_sample, xs[1])
    with pytest.raises(ValueError):
        s.add(torch.zeros(4, 3))
    o = MCMCOutput((2,), store_samples=False)
    assert o.samples is None


def test_train_val_split_matches_golden():
    from nfmc_amd.tuning import train_val_split, DualAveraging, DualAveragingParams
    from conftest import load_golden
    fx = load_golden('train_val_split')
    torch.manual_seed(19)
    _ = torch.randn(5, 7, 3)  # the generator drew x before the permutation
    tr, va = train_val_split(torch.from_numpy(fx['x']), 0.7, 16, 4)
    np.testing.assert_array_equal(tr.numpy(), fx['train'])
    np.testing.assert_array_equal(va.numpy(), fx['val'])
    fx = load_golden('tuning')
    da = DualAveraging(0.25, DualAveragingParams())
    vals = []
    for e in fx['da_errors']:
        da.step(float(e))
        vals.append(da.value)
    np.testing.assert_allclose(vals, fx['da_values'], rtol=1e-12)


def test_shard_bounds_cover_all_chains():
    from nfmc_amd.dist import Shard
    for n, w in [(10, 3), (262144, 8), (7, 8), (100, 1)]:
        spans = [Shard(rank=r, world=w).bounds(n) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_potential_recognition():
    from nfmc_amd.potentials import recognize, SumOfSquares, Funnel
    from oracle import potentials as opot
    p = recognize(lambda x: torch.sum(x ** 2, dim=1), (25,))
    assert p is not None and p.a == 1.0 and p.b == 0.0
    p = recognize(lambda x: torch.sum((x - 1.5) ** 2 * torch.arange(1, 4.), dim=-1), (3,))
    np.testing.assert_allclose(p.a.numpy(), [1, 2, 3], rtol=1e-6)
    assert recognize(lambda x: torch.sum(x ** 4, dim=1), (5,)) is None
    assert recognize(lambda x: torch.sum(x.abs(), dim=1), (5,)) is None
    assert recognize(opot.funnel(3.0), (6,)) is None
    # quadratic near the origin, something else far out (a wall at |x| = 25): the far probes catch it
    assert recognize(lambda x: torch.sum(x ** 2, dim=1) + 1e3 * (x.abs().amax(dim=1) > 25).float(), (5,)) is None
    # ... and one whose other regime starts only at the scale of the run's own x0
    wall = lambda x: torch.sum(x ** 2, dim=1) * (1 + (x.abs().amax(dim=1) > 2000).float())
    assert recognize(wall, (5,)) is not None and recognize(wall, (5,), x_scale=800.0) is None
    # stateful / stochastic targets differ between two evaluations of the same points
    calls = []
    def drifting(x):
        calls.append(1)
        return torch.sum(x ** 2, dim=1) * (1 + 1e-3 * len(calls))
    assert recognize(drifting, (5,)) is None
    assert recognize(lambda x: torch.sum(x ** 2, dim=1) + 1e-3 * torch.rand(x.shape[0], dtype=x.dtype), (5,)) is None
    x = torch.randn(7, 6)
    np.testing.assert_allclose(Funnel((6,), 3.0)(x).numpy(), opot.funnel(3.0)(x).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(SumOfSquares((6,))(x).numpy(), opot.sum_squares(x).numpy(), rtol=1e-6)


def test_fuse_never_keeps_the_callable_and_rerouting_is_logged(caplog):
    import logging
    from nfmc_amd import potentials
    from nfmc_amd.samplers.common import resolve_target
    f = lambda x: torch.sum(x ** 2, dim=1)
    assert resolve_target(f, (4,), 'never') is None and resolve_target(f, (4,), False) is None
    potentials._announced.clear()
    with caplog.at_level(logging.WARNING, logger='nfmc_amd'):
        assert resolve_target(f, (4,), 'auto', x0=torch.randn(10, 4)) is not None
        assert resolve_target(f, (4,), 'auto') is not None
    msgs = [r.getMessage() for r in caplog.records if 'closed form' in r.getMessage()]
    assert len(msgs) == 1 and 'fuse="never"' in msgs[0]      # logged once
    import inspect
    from nfmc_amd import sample
    assert "fuse" in inspect.getdoc(sample)


GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(sys.argv[2]), int(sys.argv[3])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[4], RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group('gloo', rank=rank, world_size=world)
from nfmc_amd.dist import Shard
from nfmc_amd.samplers.jump import JumpNFMCStatistics
from nfmc_amd.tuning import train_val_split
sh = Shard()
assert (sh.rank, sh.world) == (rank, world)
# C2: statistics merge
n, d = 10, 3
x = torch.arange(n * d, dtype=torch.float32).reshape(n, d)
lo, hi = sh.bounds(n)
st = JumpNFMCStatistics((d,))
st.expectations.update(x[lo:hi])
st.update_counters(n_accepted_trajectories=hi - lo, n_attempted_trajectories=2 * (hi - lo), n_accepted_jumps=rank + 1,
                   n_attempted_jumps=5, n_target_calls=7)
sh.merge_statistics(st)
assert torch.allclose(st.running_first_moment, x.mean(0)), st.running_first_moment
assert torch.allclose(st.running_second_moment, (x ** 2).mean(0))
assert st.n_accepted_trajectories == n and st.n_attempted_trajectories == 2 * n
assert st.n_accepted_jumps == sum(range(1, world + 1)) and st.n_attempted_jumps == 5 * world and st.n_target_calls == 7 * world
# seed broadcast
assert sh.broadcast_int(1234 + rank) == 1234
# C1: refit buffer all-gather, same rows on every rank
torch.manual_seed(100 + rank)
local = torch.randn(4, 6, d) + 10 * rank
tr, va = train_val_split(local, 0.7, 6, 2, shard=sh)
assert tr.shape == (5, d) and va.shape[0] <= 2, (tr.shape, va.shape)
both = torch.cat([tr, va])
gathered = [torch.empty_like(both) for _ in range(world)]
dist.all_gather(gathered, both)
assert all(torch.equal(g, gathered[0]) for g in gathered)
assert (both.mean(1) > 5).any() and (both.mean(1) < 5).any()   # rows from both ranks
# the gathered buffer is shuffled again (seed from rank 0) before the 70 % cut: train AND validation mix both ranks
local = torch.randn(8, 16, d) + 10 * rank
tr, va = train_val_split(local, 0.7, 64, 64, shard=sh)
assert tr.shape[0] == 64 and va.shape[0] == 128 - int(0.7 * 128), (tr.shape, va.shape)
for part in (tr, va):
    assert (part.mean(1) > 5).any() and (part.mean(1) < 5).any(), 'rank-ordered cut'
# uneven blocks (5 chains over 2 ranks: 3 + 2) and fewer rows than the share: the row count is agreed by an
# all-reduce MIN, so the all-gather sees equal sizes on every rank
lo, hi = sh.bounds(5)
local = torch.randn(3, hi - lo, d) + 10 * rank          # 9 rows on rank 0, 6 on rank 1
tr, va = train_val_split(local, 0.7, 4096, 4096, shard=sh)
assert tr.shape[0] + va.shape[0] == 12 and tr.shape[0] == int(0.7 * 12), (tr.shape, va.shape)
both = torch.cat([tr, va])
gathered = [torch.empty_like(both) for _ in range(world)]
dist.all_gather(gathered, both)
assert all(torch.equal(g, gathered[0]) for g in gathered)
assert sh.all_reduce_min_int(7 - rank) == 7 - (world - 1)
# update_kernel's all-reduced tuning statistics: both ranks end with the kernel of the unsharded run
from nfmc_amd.samplers import mcmc
xs = torch.randn(10, d, generator=torch.Generator().manual_seed(9)) * 2
ms = torch.rand(10, generator=torch.Generator().manual_seed(10)) > 0.4
def tuned(shard, rows, mask):
    smp = mcmc.MALA((d,), lambda v: (v ** 2).sum(-1))
    smp.shard = shard
    smp.update_kernel({'x': rows, 'mask': mask})
    return smp.kernel.inv_mass_diag, smp.kernel.step_size
lo, hi = sh.bounds(10)
imd_s, h_s = tuned(sh, xs[lo:hi], ms[lo:hi])
imd_1, h_1 = tuned(None, xs, ms)
assert torch.allclose(imd_s, imd_1, atol=1e-6) and abs(h_s - h_1) < 1e-6, (imd_s, imd_1, h_s, h_1)
dist.barrier()
dist.destroy_process_group()
print('ok', rank)
'''


def test_two_rank_gloo_collectives(tmp_path):
    """N > 1 path on CPU: world_size 2, gloo, one process per rank (C1 all-gather, C2 all-reduce, seed bcast)."""
    script = tmp_path / 'worker.py'
    script.write_text(GLOO_WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = str(s.getsockname()[1])
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), '2', port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'ok' in o


def test_training_restatement_matches_oracle_flow_and_learns():
    """flow_training evaluates the same RealNVP spec with differentiable torch ops (CPU here)."""
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd import flow_training as ft
    from oracle import flow as oflow
    torch.manual_seed(0)
    of = oflow.perturb_(oflow.Flow(oflow.RealNVP((7,), n_layers=3, conditioner_kwargs={'n_hidden': 6})), 3, 0.4)
    f = Flow(RealNVP((7,), n_layers=3, conditioner_kwargs={'n_hidden': 6}))
    f.load_state_dict(of.state_dict())
    x = torch.randn(40, 7)
    with torch.no_grad():
        z, ld = ft.forward_torch(f.bijection, x)
        zo, ldo = of.bijection.forward(x)
        xi, ldi = ft.inverse_torch(f.bijection, x)
        xio, ldio = of.bijection.inverse(x)
    np.testing.assert_allclose(z.numpy(), zo.numpy(), atol=1e-5)
    np.testing.assert_allclose(ld.numpy(), ldo.numpy(), atol=1e-5)
    np.testing.assert_allclose(xi.numpy(), xio.numpy(), atol=1e-5)
    np.testing.assert_allclose(ldi.numpy(), ldio.numpy(), atol=1e-5)
    if torch.cuda.is_available():
        return
    # maximum likelihood: fit N(1.5, 0.3^2) data, the NLL must drop well below the initial one
    g = Flow(RealNVP((4,)))
    data = 1.5 + 0.3 * torch.randn(512, 4)

    def nll():
        with torch.no_grad():
            zz, l = ft.forward_torch(g.bijection, data)
            return float(-(ft._base_log_prob(zz) + l).mean())
    before = nll()
    g.fit(data[:400], x_val=data[400:], n_epochs=150, lr=0.05, early_stopping=True, early_stopping_threshold=50)
    assert nll() < before - 2.0
    # reverse KL towards N(0, I/2): U = sum x^2
    v = Flow(RealNVP((3,)))
    v.variational_fit(lambda t: -torch.sum(t ** 2, dim=-1), n_epochs=200, lr=0.05, n_samples=256)
    with torch.no_grad():
        xs, _ = ft.inverse_torch(v.bijection, torch.randn(4000, 3))
    assert abs(float(xs.var(0).mean()) - 0.5) < 0.12


def test_adaptive_imh_host_logic():
    """Strategy routing and the host-side draws of AdaptiveIMH (imh.py:39-45,147-160)."""
    from nfmc_amd.sample import create_sampler
    from nfmc_amd.samplers import imh
    from nfmc_amd.util import get_supported_samplers
    from oracle import samplers as osamp
    assert 'adaptive_imh' in get_supported_samplers()
    s = create_sampler(lambda x: torch.sum(x ** 2, dim=-1), event_shape=(5,), strategy='adaptive_imh', flow='realnvp',
                       param_kwargs={'n_iterations': 7, 'store_samples': False})
    assert isinstance(s, imh.AdaptiveIMH) and s.params.n_iterations == 7 and s.params.store_samples is True
    for u in (0.0, 0.013, 0.4, 0.77, 0.999999):
        for m in (0, 3, 120):
            assert imh.bounded_geom_index(0.025, m, u) == osamp.bounded_geom_index(0.025, m, u)
    h = imh.HostDraws(None, [0.25, 0.5], [3])
    assert (h.rand(), h.randint(0, 5), h.rand()) == (0.25, 3, 0.5)
    with pytest.raises(ValueError):
        imh.IMHParameters(train_distribution='nope')


def test_nice_flow_string_and_training_restatement():
    """'nice' (nfmc/util.py:13): additive couplings; differentiable restatement equals the oracle's NICE."""
    from nfmc_amd.flows import NICE
    from nfmc_amd.util import create_flow_object, get_supported_normalizing_flows, is_flow_supported
    from nfmc_amd import flow_training as ft
    from oracle import flow as oflow
    assert is_flow_supported('nice') and 'nice' in get_supported_normalizing_flows()
    f = create_flow_object('nice%{"n_layers": 3}', (6,))
    assert isinstance(f.bijection, NICE) and f.bijection.min_scale == 1.0 and len(f.bijection.layers) == 2 + 2 * 3
    torch.manual_seed(0)
    of = oflow.perturb_(oflow.Flow(oflow.NICE((6,), n_layers=3)), 3, 0.4)
    f.load_state_dict(of.state_dict())
    x = torch.randn(30, 6)
    with torch.no_grad():
        z, ld = ft.forward_torch(f.bijection, x)
        zo, ldo = of.bijection.forward(x)
        xi, ldi = ft.inverse_torch(f.bijection, z)
    np.testing.assert_allclose(z.numpy(), zo.numpy(), atol=1e-6)
    np.testing.assert_allclose(ld.numpy(), ldo.numpy(), atol=1e-6)
    np.testing.assert_allclose(xi.numpy(), x.numpy(), atol=1e-5)
    np.testing.assert_allclose(ldi.numpy(), -ld.numpy(), atol=1e-6)
    const = float(of.bijection.layers[0].log_scale.sum() + of.bijection.layers[-1].log_scale.sum())
    assert float((ldo - const).abs().max()) < 1e-6


def _integration_stub_namespace():
    """The ctypes stub INTEGRATION.md shows for the reference, executed as written (library path filled in)."""
    import re
    from nfmc_amd import hip
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = re.search(r"```python\n(# nfmc/algorithms/sampling/mcmc/_hip\.py.*?)```", text, re.S).group(1)
    block = block.replace("C.CDLL('libnfmc_hip.so')", "C.CDLL(%r)" % hip.LIB_PATH)
    ns = {}
    exec(compile(block, 'INTEGRATION.md', 'exec'), ns)
    return ns


def test_integration_stub_structs_match_the_header():
    import ctypes as C
    from nfmc_amd import hip
    ns = _integration_stub_namespace()
    for name in ('NfmcPotential', 'NfmcRng', 'NfmcStats', 'NfmcMalaArgs'):
        assert C.sizeof(ns[name]) == C.sizeof(getattr(hip, name)), name
        assert [f[0] for f in ns[name]._fields_] == [f[0] for f in getattr(hip, name)._fields_], name


def test_rqs_flow_string_and_training_restatement():
    """'c-rqnsf' (nfmc/util.py:17): flow string, layer count, differentiable restatement equals the oracle's spline."""
    from nfmc_amd.flows import CRQNSF, RQSCoupling
    from nfmc_amd.util import create_flow_object, is_flow_supported
    from nfmc_amd import flow_training as ft
    from oracle import flow as oflow
    assert is_flow_supported('c-rqnsf') and is_flow_supported('c-rqsnsf')
    f = create_flow_object('c-rqnsf%{"n_layers": 3, "conditioner_kwargs": {"n_hidden": 6}}', (7,))
    assert isinstance(f.bijection, CRQNSF) and len(f.bijection.layers) == 8
    assert sum(isinstance(m, RQSCoupling) for m in f.bijection.layers) == 3
    assert f.bijection.layers[2].conditioner[-1].out_features == 23 * 4
    torch.manual_seed(0)
    of = oflow.perturb_(oflow.Flow(oflow.CRQNSF((7,), n_layers=3, conditioner_kwargs={'n_hidden': 6})), 3, 1.0)
    f.load_state_dict(of.state_dict())
    x = torch.randn(50, 7) * 2.5
    with torch.no_grad():
        z, ld = ft.forward_torch(f.bijection, x)
        zo, ldo = of.bijection.forward(x)
        xi, ldi = ft.inverse_torch(f.bijection, z)
        xio, ldio = of.bijection.inverse(z)
    np.testing.assert_allclose(z.numpy(), zo.numpy(), atol=1e-6)
    np.testing.assert_allclose(ld.numpy(), ldo.numpy(), atol=1e-5)
    np.testing.assert_allclose(xi.numpy(), xio.numpy(), atol=1e-6)
    np.testing.assert_allclose(ldi.numpy(), ldio.numpy(), atol=1e-5)
    np.testing.assert_allclose(xi.numpy(), x.numpy(), atol=5e-3)      # fp32 round trip through three spline layers
    # the restatement is differentiable in the weights (what Flow.fit needs)
    zz, ll = ft.forward_torch(f.bijection, x)
    loss = (0.5 * (zz * zz).sum(-1) - ll).mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in f.parameters())


def test_default_kernel_width_policy():
    """At d = 64 / 128 conditioners of width 9..32 are presented zero-padded to 64 (matrix-core flow kernels); narrow
    ones, other shapes and spline flows are left alone (flows.py: default_min_hidden)."""
    from nfmc_amd.flows import CRQNSF, RealNVP
    mk = lambda d, h, cls=RealNVP, cl=2: cls((d,), conditioner_kwargs={'n_hidden': h, 'n_layers': cl})
    assert mk(64, 16).default_min_hidden() == 64 and mk(128, 32).default_min_hidden() == 64
    assert mk(64, 8).default_min_hidden() == 0 and mk(64, 4).default_min_hidden() == 0
    assert mk(100, 16).default_min_hidden() == 0 and mk(64, 16, cl=3).default_min_hidden() == 0
    assert mk(64, 64).default_min_hidden() == 0          # already wide: nothing to pad
    assert mk(64, 16, CRQNSF).default_min_hidden() == 0


def test_metropolis_log_ratio_ordering():
    """Mirror of the reference's test/test_metropolization.py."""
    from nfmc_amd.util import metropolis_acceptance_log_ratio
    target = lambda x: 0.5 * torch.sum(x ** 2, dim=1)
    proposal = lambda x: 0.5 * torch.sum((x / 3.0) ** 2, dim=1) + x.shape[1] * float(np.log(3.0))
    x0, x1 = torch.tensor([[-100.0, -100.0]]), torch.tensor([[0.0, 0.0]])
    forward = metropolis_acceptance_log_ratio(-target(x0), -target(x1), -proposal(x0), -proposal(x1))
    inverse = metropolis_acceptance_log_ratio(-proposal(x0), -proposal(x1), -target(x0), -target(x1))
    assert forward > inverse


def test_sample_store_slabs_equal_stepwise_reference_semantics():
    """MCMCSamples fed whole launches (slabs of k steps) keeps exactly the rows the reference's step-by-step
    `add` keeps (base.py:234-263: every `thinning`-th offered row, the newest `max_samples`, `last_sample` always),
    whatever the slab sizes."""
    import random
    from nfmc_amd.containers import MCMCSamples
    rng = random.Random(0)
    for trial in range(60):
        thinning = rng.choice([1, 1, 2, 3, 5])
        max_samples = rng.choice([None, None, 1, 4, 7])
        store = rng.random() > 0.15
        total = rng.randint(0, 40)
        steps = torch.arange(total, dtype=torch.float32).reshape(total, 1, 1).expand(total, 3, 2).contiguous()
        s = MCMCSamples((2,), store_samples=store, thinning=thinning, max_samples=max_samples)
        kept, seen, i = [], 0, 0
        while i < total:
            k = rng.randint(1, 9)
            slab = steps[i:i + k]
            s.add(slab if len(slab) > 1 or rng.random() > 0.5 else slab[0])   # (k, n, *e) or a single (n, *e) state
            for row in slab:                                                    # the reference, one step at a time
                if store:
                    if seen % thinning == 0:
                        kept.append(row)
                        if max_samples is not None and len(kept) > max_samples:
                            kept.pop(0)
                    seen += 1
            i += len(slab)
        if total:
            assert torch.equal(s.last_sample, steps[-1]) and torch.equal(s[-1], steps[-1])
        want = torch.stack(kept) if kept else None
        got = s.as_tensor()
        assert s.n_samples == len(kept)
        if want is None:
            assert got.shape[0] == 0
        else:
            assert torch.equal(got, want), (trial, thinning, max_samples)


def test_flow_copy_state_and_kernel_shape_limits():
    """Host logic around the flow object: the device caches never travel with a copy / pickle, and shapes beyond the
    kernels (nfmc_limits: d > 512, conditioners wider than 128, 32 for splines) are recognised without a GPU."""
    import copy, io
    from nfmc_amd.flows import CRQNSF, Flow, RealNVP
    f = Flow(RealNVP((6,)))
    f.bijection._pack_cache = {0: ('key', object())}        # what a used flow carries (ctypes structs on a GPU box)
    key = f.bijection._version_key('cpu')
    assert '_mods_cache' in f.bijection.__dict__
    twin = copy.deepcopy(f)
    assert twin.bijection._pack_cache is None and '_mods_cache' not in twin.bijection.__dict__
    assert twin.bijection._version_key('cpu') != key          # its own parameter storage
    assert all(torch.equal(a, b) for a, b in zip(f.parameters(), twin.parameters()))
    buf = io.BytesIO()
    torch.save(f, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert back.bijection._pack_cache is None
    with torch.no_grad():
        f.bijection.layers[0].shift.add_(1.0)
    assert f.bijection._version_key('cpu') != key             # in-place update invalidates the packed blob
    assert not RealNVP((512,)).beyond_kernels() and RealNVP((513,)).beyond_kernels()
    assert not RealNVP((16,), conditioner_kwargs={'n_hidden': 128}).beyond_kernels()
    assert RealNVP((16,), conditioner_kwargs={'n_hidden': 129}).beyond_kernels()
    assert not CRQNSF((16,), conditioner_kwargs={'n_hidden': 32}).beyond_kernels()
    assert CRQNSF((16,), conditioner_kwargs={'n_hidden': 33}).beyond_kernels()


def test_bench_launcher_starts_one_process_per_gpu():
    """`python bench.py --gpus 2` without torchrun: the parent starts two fresh ranks before touching any GPU, they
    rendezvous (gloo here, nccl = RCCL on the GPU box: same code up to the backend name), and exactly ONE JSON line
    comes back on stdout, carrying n_gpus = 2 and the world size the backend itself reported."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--rehearse',
                        '--reps', '3'], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['world_size_reported_by_backend'] == 2 and len(line['per_rank_ms']) == 2
    # under a launcher that already set WORLD_SIZE (torchrun) the same file is one rank and spawns nothing
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "'WORLD_SIZE' not in os.environ and args.gpus > 1" in src


def test_device_sample_store_plans_exactly_the_reference_rows():
    """DeviceSampleStore decides BEFORE every launch which transitions are kept and in which ring row
    (NfmcSampleStore: stride, countdown, ring_rows, row).  A Python model of the kernels' cursor (csrc/common.hpp
    StoreCursor) fed those structs must leave exactly the rows the reference's step-by-step MCMCSamples.add keeps
    (base.py:249-263: every `thinning`-th offered state by global index, the newest `max_samples`), for any launch
    sizes, early stops, and mixes of kernel launches with host-side dense adds -- in at most max_samples rows."""
    import random
    from nfmc_amd.containers import DeviceSampleStore, MCMCSamples
    rng = random.Random(1)
    for trial in range(200):
        thinning = rng.choice([1, 1, 2, 3, 7])
        max_samples = rng.choice([None, None, 1, 2, 5, 16])
        total = rng.choice([1, 2, 9, 40, 101])
        offered = total if rng.random() > 0.3 else rng.randrange(0, total + 1)   # time limit: stop early
        n, d = 2, 3
        states = torch.arange(total * n * d, dtype=torch.float32).reshape(total, n, d) + 1
        st = DeviceSampleStore(n, d, 'cpu', total, thinning, max_samples)
        assert st.rows <= (max_samples or total) and st.rows <= -(-total // thinning)
        done = 0
        while done < offered:
            k = min(rng.choice([1, 1, 3, 8, 512]), offered - done)
            if rng.random() < 0.3:
                st.add_dense(states[done:done + k])
            else:
                stride, countdown, ring_rows, row = st.plan(k)   # what a kernel launch of k transitions receives
                assert 0 <= countdown < stride and 0 <= row < ring_rows
                for s_ in range(k):                      # StoreCursor::next
                    if countdown > 0:
                        countdown -= 1
                        continue
                    st.buf[row] = states[done + s_]
                    row = 0 if row + 1 == ring_rows else row + 1
                    countdown = stride - 1
            done += k
        # the reference, one state at a time
        idx = [i for i in range(offered) if i % thinning == 0]
        if max_samples:
            idx = idx[-max_samples:]
        want = states[idx] if idx else torch.empty(0, n, d)
        got = st.ordered()
        assert got.shape == want.shape and torch.equal(got, want), (trial, thinning, max_samples, total, offered)
        ms = MCMCSamples((d,), thinning=thinning, max_samples=max_samples)
        ms.adopt_store(st)
        assert ms.n_samples == len(idx) and ms.seen_samples == offered
        if idx:
            assert torch.equal(ms.as_tensor(), want)


def test_flow_fit_survives_non_finite_losses():
    """variational_fit defaults to check_for_divergences=False: a non-finite loss must not reach backward()/step()
    (it would poison every weight), and a fit whose every epoch is non-finite leaves the incoming weights in place."""
    from nfmc_amd import flow_training as ft
    from nfmc_amd.flows import Flow, RealNVP
    torch.manual_seed(0)
    f = Flow(RealNVP((4,)))
    before = {k: v.clone() for k, v in f.state_dict().items()}
    ft.variational_fit(f, lambda x: torch.full((x.shape[0],), float('nan')), n_epochs=3, n_samples=16)
    after = f.state_dict()
    assert all(torch.equal(before[k], after[k]) for k in before)
    calls = []
    def sometimes_nan(x):
        calls.append(1)
        lp = -0.5 * (x ** 2).sum(-1)
        return lp * float('nan') if len(calls) == 2 else lp
    ft.variational_fit(f, sometimes_nan, n_epochs=5, n_samples=64, lr=0.01)
    assert all(torch.isfinite(v).all() for v in f.state_dict().values())
    with pytest.raises(ValueError):
        ft.variational_fit(f, lambda x: torch.full((x.shape[0],), float('nan')), n_epochs=2, n_samples=8,
                           check_for_divergences=True)


# ------------------------------------------------------------------------------------------------ code-object metadata
def _gfx950_kernels(path):
    """Kernel metadata (AMDGPU msgpack note) of every gfx950 code object bundled in a shared library."""
    import re
    import struct
    import msgpack
    data = open(path, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    for m in re.finditer(magic, data):
        b0 = m.start()
        n, = struct.unpack_from('<Q', data, b0 + len(magic))
        p = b0 + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if 'gfx950' not in triple or not size:
                continue
            elf = data[b0 + off:b0 + off + size]
            assert elf[:4] == b'\x7fELF'
            shoff, = struct.unpack_from('<Q', elf, 0x28)
            shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
            for i in range(shnum):
                sh = elf[shoff + i * shentsize: shoff + (i + 1) * shentsize]
                if struct.unpack_from('<I', sh, 4)[0] != 7:   # SHT_NOTE
                    continue
                q, size_n = struct.unpack_from('<QQ', sh, 0x18)
                end = q + size_n
                while q < end:
                    namesz, descsz, ntype = struct.unpack_from('<III', elf, q)
                    q += 12
                    name = elf[q:q + namesz]
                    q += (namesz + 3) & ~3
                    desc = elf[q:q + descsz]
                    q += (descsz + 3) & ~3
                    if ntype == 32 and name.startswith(b'AMDGPU'):
                        for k in msgpack.unpackb(desc, raw=False, strict_map_key=False).get('amdhsa.kernels', []):
                            yield k


def test_hand_counted_memory_waits_see_no_compiler_inserted_memory_operations():
    """`imh_scan_kernel` (csrc/imh_parallel.hpp) requests its records with inline `global_load_dwordx3` and waits with
    `s_waitcnt vmcnt(N)`, N counted by hand from the memory operations between request and use.  A register spill would put
    scratch stores / loads in between that the count does not know about.  The shipped code object must therefore show no
    scratch and no spills for these kernels (a lane has up to 512 registers there: one wave per SIMD)."""
    from nfmc_amd import build
    lib = build.build(verbose=False)
    seen = 0
    for k in _gfx950_kernels(lib):
        if 'imh_scan_kernel' not in k['.name']:
            continue
        seen += 1
        assert k['.private_segment_fixed_size'] == 0, k['.name']
        assert k.get('.vgpr_spill_count', 0) == 0 and k.get('.sgpr_spill_count', 0) == 0, k['.name']
        assert k['.vgpr_count'] <= 512
    assert seen >= 2   # with and without the per-step outputs, in the affine and the spline unit


def test_refit_split_permutation_matches_the_oracle_and_is_a_permutation():
    """The refit buffer's shuffled split (tuning.py:44-65) on the device gathers rows pi(0), pi(1), ... of a keyed
    pseudo-random permutation (csrc/fit_support.hip).  The library's HOST evaluation of pi (`nfmc_rows_sample_index`: the
    same inline function the kernel calls, integer arithmetic only) against oracle/shuffle.py index for index, pi a
    bijection of [0, N) at sizes around the powers of two (cycle walking), different seeds give different orders, and the
    first rows are spread over the whole buffer (a shuffle, not a rotation)."""
    from nfmc_amd import hip
    from oracle import shuffle
    lib = hip.lib()
    for n, seed in ((1, 5), (2, 9), (3, 1), (7, 2), (64, 3), (65, 4), (1000, 2 ** 61 + 12345), (4097, 77)):
        got = [int(lib.nfmc_rows_sample_index(n, seed, i)) for i in range(n)]
        assert got == shuffle.permutation_prefix(n, seed, n).tolist(), (n, seed)
        assert sorted(got) == list(range(n)), (n, seed)
    n = 163840   # the C5 refit buffer: 5 x 32768 pooled rows
    a = [int(lib.nfmc_rows_sample_index(n, 11, i)) for i in range(2048)]
    b = [int(lib.nfmc_rows_sample_index(n, 12, i)) for i in range(2048)]
    assert a == shuffle.permutation_prefix(n, 11, 2048).tolist()
    assert len(set(a)) == 2048 and a != b
    counts = np.bincount(np.asarray(a) * 16 // n, minlength=16)          # 128 expected per sixteenth
    assert counts.min() > 80 and counts.max() < 180, counts.tolist()
    assert lib.nfmc_rows_sample_index(n, 11, n) == -1 and lib.nfmc_rows_sample_index(0, 1, 0) == -1


def test_streamed_kernels_workspace_is_bounded_and_caller_supplied():
    """include/nfmc_hip.h: the library never allocates.  The streamed matrix-core kernels (wide conditioner at d = 256 / 512
    ...) take their slab from NfmcRealNVP.scratch; its size is bounded by the workgroup slots (256 x 128 rows of d floats, x 2
    with the gradient), not by n, and 0 for every shape that runs register- or LDS-resident."""
    import ctypes as C
    from nfmc_amd import hip
    lib = hip.lib()

    def need(d, H, n, grad):
        st = hip.NfmcRealNVP(d, 2, H, 2, 1e-3, 0, None, None, None, None, None, 0, 0.0, 0)
        return int(lib.nfmc_flow_scratch_bytes(C.byref(st), n, grad))
    assert need(256, 128, 100, 0) == 128 * 256 * 4 and need(256, 128, 100, 1) == 2 * 128 * 256 * 4      # one workgroup slot
    assert need(256, 128, 10 ** 9, 1) == need(256, 128, 256 * 128, 1) == 2 * 256 * 128 * 256 * 4       # bounded by 256 slots
    assert need(512, 64, 70000, 0) == 256 * 128 * 512 * 4
    for d, H in ((128, 128), (64, 64), (256, 8), (256, 32), (100, 128)):
        assert need(d, H, 4096, 1) == 0, (d, H)
    # the fused wide trajectory's scratch: gradient and U~ at the state + three slab rows (position, gradient, momentum) per
    # lane group of a workgroup slot
    sb = lib.nfmc_neutra_scratch_bytes
    assert sb(1000, 256, 128, 2, 2) == 4 * (1000 * 256 + 1000 + 3 * 8 * 128 * 256)
