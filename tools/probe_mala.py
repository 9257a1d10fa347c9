"""GPU probe: time the fused MALA / HMC / flow-MH kernels across layout configs (tuning aid)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import hip
from nfmc_amd.potentials import SumOfSquares

dev = torch.device('cuda', 0)
lib = hip.lib()


def time_mala(n, d, K, cfg, reps=5, adjust=1, store=False):
    if cfg:
        os.environ['NFMC_SAMPLER_CFG'] = cfg
    else:
        os.environ.pop('NFMC_SAMPLER_CFG', None)
    x = torch.randn(n, d, device=dev) * 0.7
    st = hip.DeviceStats(d, dev)
    pot = SumOfSquares((d,))
    samples = torch.empty(K, n, d, device=dev) if store else None
    a = hip.NfmcMalaArgs()
    a.x, a.n, a.d, a.n_steps = hip.ptr(x), n, d, K
    a.step_size, a.adjust = d ** (-1 / 3), adjust
    a.pot = pot.descriptor(dev)
    a.stats = st.struct()
    a.samples = hip.dense_store(samples if store else None)
    ts = []
    for r in range(reps + 1):
        a.rng = hip.make_rng(1, 0, r * K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.check(lib.nfmc_mala_steps_f32(C.byref(a), hip.stream()), 'mala')
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    return t, n * K / t


def time_hmc(n, d, K, L, cfg, reps=3):
    if cfg:
        os.environ['NFMC_SAMPLER_CFG'] = cfg
    else:
        os.environ.pop('NFMC_SAMPLER_CFG', None)
    x = torch.randn(n, d, device=dev) * 0.7
    st = hip.DeviceStats(d, dev)
    pot = SumOfSquares((d,))
    a = hip.NfmcHmcArgs()
    a.x, a.n, a.d, a.n_steps = hip.ptr(x), n, d, K
    a.step_size, a.n_leapfrog, a.adjust = 0.01, L, 1
    a.pot = pot.descriptor(dev)
    a.stats = st.struct()
    ts = []
    for r in range(reps + 1):
        a.rng = hip.make_rng(1, 0, r * K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.check(lib.nfmc_hmc_steps_f32(C.byref(a), hip.stream()), 'hmc')
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    return t, n * K / t


def time_flow_mh(n, d, K, nh=None, reps=3):
    from nfmc_amd.flows import Flow, RealNVP
    from nfmc_amd.samplers.jump import launch_flow_mh

    class R:  # minimal Run stand-in
        pass
    f = Flow(RealNVP((d,), conditioner_kwargs={'n_hidden': nh} if nh else None))
    run = R()
    run.dev, run.n, run.d = dev, n, d
    run.x = torch.randn(n, d, device=dev) * 0.7
    run.rng = lambda step0, k=0, adjusted=True: hip.make_rng(1, 0, step0)
    import contextlib
    run.timed = lambda label: contextlib.nullcontext()
    st = hip.DeviceStats(d, dev)
    logq = torch.empty(n, device=dev)
    pot = SumOfSquares((d,))
    ts = []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        launch_flow_mh(run, f, pot, logq, K, r * K, False, True, st.struct())
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    return t, n * K / t


if __name__ == '__main__':
    print('device', torch.cuda.get_device_name(0), flush=True)
    for cfg in ['16,4', '8,8', '4,16', None]:
        try:
            t, r = time_mala(65536, 64, 100, cfg)
            print(f'mala n=65536 d=64 K=100 cfg={cfg}: {t*1e3:.3f} ms  {r/1e9:.3f} G chain-steps/s', flush=True)
        except Exception as e:
            print('cfg', cfg, 'failed', e, flush=True)
    os.environ.pop('NFMC_SAMPLER_CFG', None)
    t, r = time_mala(65536, 64, 100, None, adjust=0)
    print(f'ula  n=65536 d=64 K=100: {t*1e3:.3f} ms  {r/1e9:.3f} G/s', flush=True)
    t, r = time_mala(65536, 64, 100, None, store=True)
    print(f'mala+store n=65536 d=64 K=100: {t*1e3:.3f} ms  {r/1e9:.3f} G/s', flush=True)
    t, r = time_mala(65536, 64, 1, None, reps=10)
    print(f'mala K=1: {t*1e6:.1f} us  {r/1e9:.3f} G/s', flush=True)
    for n in (8192, 262144, 1048576):
        t, r = time_mala(n, 64, 100, None)
        print(f'mala n={n} d=64 K=100: {t*1e3:.3f} ms  {r/1e9:.3f} G/s', flush=True)
    for d, cfgs in ((128, ['16,8']), (256, ['16,16']), (25, [None]), (32, ['4,8'])):
        for cfg in cfgs:
            t, r = time_mala(65536, d, 50, cfg)
            print(f'mala n=65536 d={d} K=50 cfg={cfg}: {t*1e3:.3f} ms  {r/1e9:.3f} G/s', flush=True)
    t, r = time_hmc(32768, 256, 5, 20, None)
    print(f'hmc n=32768 d=256 K=5 L=20: {t*1e3:.3f} ms  {r/1e6:.2f} M traj/s', flush=True)
    t, r = time_hmc(65536, 128, 5, 10, None)
    print(f'hmc n=65536 d=128 K=5 L=10: {t*1e3:.3f} ms  {r/1e6:.2f} M traj/s', flush=True)
    for (n, d, K, nh) in [(65536, 64, 1, None), (8192, 64, 100, None), (32768, 256, 1, None), (65536, 64, 4, 32)]:
        t, r = time_flow_mh(n, d, K, nh)
        print(f'flow_mh n={n} d={d} K={K} H={nh}: {t*1e3:.3f} ms  {r/1e6:.2f} M chain-steps/s', flush=True)
