"""CPU: the oracle (oracle/) against the golden vectors recorded from the reference itself.

Fixtures were produced by tests/golden/make_golden.py running the reference's own modules;
here the oracle replays the recorded noise and must reproduce the reference's outputs.
Tolerance: fp32 eager ops in the same order => 2e-6 abs on O(1) states; counters exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, golden_flow, FailingTarget
from oracle import samplers as osamp
from oracle import potentials as opot
from oracle import philox

ATOL = 2e-6


def _noise(fx):
    normals = [torch.from_numpy(v) for v in fx['noise/normals']] if fx['noise/normals'].ndim > 1 else []
    uniforms = [torch.from_numpy(v) for v in fx['noise/uniforms']] if fx['noise/uniforms'].ndim > 1 else []
    return osamp.ReplayNoise(normals, uniforms)


def _check(tr, fx, jump=False):
    np.testing.assert_allclose(tr.stacked().numpy(), fx['exp/samples'], atol=ATOL, rtol=0)
    np.testing.assert_allclose(tr.moments.first.numpy(), fx['exp/first_moment'], atol=ATOL, rtol=0)
    np.testing.assert_allclose(tr.moments.second.numpy(), fx['exp/second_moment'], atol=ATOL, rtol=0)
    c = fx['exp/counters']
    assert (tr.n_accepted, tr.n_attempted, tr.n_target_calls, tr.n_target_gradient_calls) == (c[0], c[1], c[3], c[4])
    if jump:
        assert (tr.n_accepted_jumps, tr.n_attempted_jumps) == tuple(fx['exp/jump_counters'])


@pytest.mark.parametrize('name,adjust,pot', [('mala_d6', True, 'sumsq'), ('mala_d7_mass', True, 'sumsq'),
                                             ('ula_d6', False, 'sumsq'), ('mala_funnel_d5', True, 'funnel')])
def test_langevin(name, adjust, pot):
    fx = load_golden(name)
    target = opot.sum_squares if pot == 'sumsq' else opot.funnel(3.0)
    tr = osamp.mcmc_sample(torch.from_numpy(fx['x0']), target, 'langevin', fx['exp/samples'].shape[0],
                           float(fx['step_size']), torch.from_numpy(fx['inv_mass_diag']), adjustment=adjust,
                           noise=_noise(fx))
    _check(tr, fx)


@pytest.mark.parametrize('name,adjust', [('hmc_d5', True), ('hmc_d6_mass', True), ('uhmc_d5', False)])
def test_hmc(name, adjust):
    fx = load_golden(name)
    tr = osamp.mcmc_sample(torch.from_numpy(fx['x0']), opot.sum_squares, 'hmc', fx['exp/samples'].shape[0],
                           float(fx['step_size']), torch.from_numpy(fx['inv_mass_diag']),
                           n_leapfrog=int(fx['n_leapfrog']), adjustment=adjust, noise=_noise(fx))
    _check(tr, fx)


@pytest.mark.parametrize('name,adjust', [('mh_d5', True), ('rw_d6', False)])
def test_random_walk_mh(name, adjust):
    fx = load_golden(name)
    tr = osamp.mcmc_sample(torch.from_numpy(fx['x0']), opot.sum_squares, 'mh', fx['exp/samples'].shape[0], 0.01,
                           torch.from_numpy(fx['inv_mass_diag']), adjustment=adjust, noise=_noise(fx))
    _check(tr, fx)


@pytest.mark.parametrize('name,kind,adjust', [('mala_fail_d5', 'langevin', True), ('ula_fail_d5', 'langevin', False),
                                              ('hmc_fail_d5', 'hmc', True), ('mh_fail_d5', 'mh', True)])
def test_target_failure_rejects_the_step_and_counts_a_divergence(name, kind, adjust):
    """langevin.py:111-114, hmc.py:117-120, mh.py:63-66: ValueError from the target -> x' = x, nobody accepts,
    n_divergences += 1, the step's uniforms are never drawn; the run continues."""
    fx = load_golden(name)
    target = FailingTarget(opot.sum_squares, fx['fail_calls'])
    tr = osamp.mcmc_sample(torch.from_numpy(fx['x0']), target, kind, fx['exp/samples'].shape[0],
                           float(fx['step_size']) if 'step_size' in fx else 0.01, torch.from_numpy(fx['inv_mass_diag']),
                           n_leapfrog=int(fx['n_leapfrog']) if 'n_leapfrog' in fx else 20, adjustment=adjust,
                           noise=_noise(fx))
    _check(tr, fx)
    assert tr.n_divergences == int(fx['exp/counters'][2]) == len(fx['fail_calls'])


def test_jump_with_failing_target():
    """jump.py:183 (inner divergences are carried over) and :226-227 (a failing jump rejects, books no calls)."""
    fx = load_golden('jump_mala_fail_d6')
    flow = golden_flow(fx, 6)
    target = FailingTarget(opot.sum_squares, fx['fail_calls'])
    tr = osamp.jump_sample(torch.from_numpy(fx['x0']), target, flow, 'langevin', int(fx['n_outer']),
                           int(fx['n_inner']), float(fx['step_size']), noise=_noise(fx))
    _check(tr, fx, jump=True)
    assert tr.n_divergences == int(fx['exp/counters'][2]) == 1


def test_jump_mala():
    fx = load_golden('jump_mala_d6')
    flow = golden_flow(fx, 6)
    tr = osamp.jump_sample(torch.from_numpy(fx['x0']), opot.sum_squares, flow, 'langevin', int(fx['n_outer']),
                           int(fx['n_inner']), float(fx['step_size']), noise=_noise(fx))
    _check(tr, fx, jump=True)


def test_jump_hmc():
    fx = load_golden('jump_hmc_d8')
    flow = golden_flow(fx, 8, int(fx['flow_n_layers']), int(fx['flow_n_hidden']), int(fx['flow_cond_layers']))
    tr = osamp.jump_sample(torch.from_numpy(fx['x0']), opot.sum_squares, flow, 'hmc', int(fx['n_outer']),
                           int(fx['n_inner']), float(fx['step_size']), n_leapfrog=int(fx['n_leapfrog']),
                           noise=_noise(fx))
    _check(tr, fx, jump=True)


@pytest.mark.parametrize('name,d,nl', [('imh_d6', 6, 2), ('imh_d7_odd', 7, 3)])
def test_imh(name, d, nl):
    fx = load_golden(name)
    flow = golden_flow(fx, d, nl)
    tr = osamp.imh_sample(torch.from_numpy(fx['x0']), opot.sum_squares, flow, int(fx['n_iterations']), noise=_noise(fx))
    _check(tr, fx)


@pytest.mark.parametrize('name,dist,d', [('adaptive_imh_d6', 'uniform', 6), ('adaptive_imh_geom_d5', 'bounded_geom', 5)])
def test_adaptive_imh(name, dist, d):
    """AdaptiveIMH.sample (imh.py:103-181) incl. the per-iteration one-epoch refits: host draws replayed."""
    fx = load_golden(name)
    flow = golden_flow(fx, d)
    host = osamp.ReplayHostDraws(fx['noise/host_uniforms'], fx['noise/host_ints'])
    tr = osamp.adaptive_imh_sample(torch.from_numpy(fx['x0']), opot.sum_squares, flow, int(fx['n_iterations']),
                                   adaptation_dropoff=float(fx['adaptation_dropoff']), train_distribution=dist,
                                   noise=_noise(fx), host=host)
    _check(tr, fx)
    assert tr.n_refits > 0 and not host.uniforms and not host.ints
    for k, v in flow.state_dict().items():
        np.testing.assert_allclose(v.numpy(), fx['flow_final/' + k], atol=ATOL, rtol=0)


def test_neutra_hmc():
    fx = load_golden('neutra_hmc_d6')
    flow = golden_flow(fx, 6)
    tr = osamp.neutra_hmc_sample(torch.from_numpy(fx['x0']), opot.sum_squares, flow, int(fx['n_iterations']),
                                 float(fx['step_size']), n_leapfrog=int(fx['n_leapfrog']), noise=_noise(fx))
    _check(tr, fx)


def test_train_val_split():
    fx = load_golden('train_val_split')
    tr, va = osamp.train_val_split(torch.from_numpy(fx['x']), 0.7, 16, 4, perm=torch.from_numpy(fx['perm']))
    np.testing.assert_array_equal(tr.numpy(), fx['train'])
    np.testing.assert_array_equal(va.numpy(), fx['val'])


def test_dual_averaging():
    fx = load_golden('tuning')
    da = osamp.DualAveraging(0.25)
    vals = []
    for e in fx['da_errors']:
        da.step(float(e))
        vals.append(da.value)
    np.testing.assert_allclose(vals, fx['da_values'], rtol=1e-12)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        r = philox.philox4x32_10(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert tuple(int(v) for v in r) == exp


def test_philox_streams_are_sane():
    z = philox.normal_field(7, np.arange(4096), 3, 64, philox.TAG_NOISE)
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert not np.allclose(z, philox.normal_field(7, np.arange(4096), 4, 64, philox.TAG_NOISE))
    # sharding independence: chains 100..199 drawn alone equal the slice of a bigger draw
    np.testing.assert_array_equal(philox.normal_field(7, np.arange(100, 200), 9, 10, 0),
                                  philox.normal_field(7, np.arange(0, 300), 9, 10, 0)[100:200])
    u = philox.accept_uniform(7, np.arange(10000), 5)
    assert 0 < u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.02


def test_metropolis_log_ratio_convention():
    fx = load_golden('util')
    a, b, c, d = (np.array(v, np.float32) for v in ([1.0, -2.0], [0.5, 3.0], [0.25, 0.0], [-1.0, 4.0]))
    np.testing.assert_allclose(b - a + c - d, fx['log_ratio'])


def test_rqs_flow_known_answers():
    """'c-rqnsf' (build-defined spline couplings, parity unpinned): the known answers of SURVEY 8c on the CPU spec:
    round trip, logdet antisymmetry, logdet = slogdet of the autograd Jacobian, identity outside [-B, B]."""
    from oracle import flow as oflow
    torch.manual_seed(3)
    d = 6
    f = oflow.perturb_(oflow.Flow(oflow.CRQNSF((d,), n_layers=3, conditioner_kwargs={'n_hidden': 5})), 4, 1.5).double()
    x = torch.randn(200, d, dtype=torch.float64) * 2.0
    with torch.no_grad():
        z, ld = f.bijection.forward(x)
        xb, ldi = f.bijection.inverse(z)
    np.testing.assert_allclose(xb.numpy(), x.numpy(), atol=1e-10)
    np.testing.assert_allclose(ldi.numpy(), -ld.numpy(), atol=1e-10)
    for i in range(5):
        J = torch.autograd.functional.jacobian(lambda v: f.bijection.forward(v[None])[0][0], x[i])
        np.testing.assert_allclose(float(torch.linalg.slogdet(J)[1]), float(ld[i]), atol=1e-9)
    # the spline really bends the map (not an affine flow in disguise) and is the identity beyond the bound
    c = f.bijection.layers[2]
    far = torch.full((3, d), 7.5, dtype=torch.float64)
    out, l0 = c.forward(far)
    assert torch.equal(out, far) and float(l0.abs().max()) == 0.0
    mid = torch.linspace(-4, 4, 50, dtype=torch.float64)[:, None].repeat(1, d)
    out, _ = c.forward(mid)
    second = out[2:, -1] - 2 * out[1:-1, -1] + out[:-2, -1]
    assert float(second.abs().max()) > 1e-4
    assert len(f.bijection.layers) == 2 + 2 * 3


def test_philox_7_round_stream_known_answers():
    """The opt-in stream (NfmcRng.rounds = 7): Philox4x32-7 known-answer vectors of the Random123 distribution
    (kat_vectors: `philox4x32 7 ...`), next to the 10-round ones the default stream is pinned with."""
    z, f = np.uint32(0), np.uint32(0xffffffff)
    got = [int(v) for v in philox.philox4x32_10(z, z, z, z, 0, 0, rounds=7)]
    assert got == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]
    got = [int(v) for v in philox.philox4x32_10(f, f, f, f, 0xffffffff, 0xffffffff, rounds=7)]
    assert got == [0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662]
    got = [int(v) for v in philox.philox4x32_10(z, z, z, z, 0, 0, rounds=10)]
    assert got == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    a = philox.normal_field(5, np.arange(4, dtype=np.uint32), 3, 8, 0, rounds=7)
    b = philox.normal_field(5, np.arange(4, dtype=np.uint32), 3, 8, 0, rounds=10)
    assert np.isfinite(a).all() and not np.allclose(a, b)
