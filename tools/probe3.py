import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_mala import time_mala, time_hmc, time_flow_mh
for d, cfgs in ((64, ['8,8', '16,4']), (128, ['8,16', '16,8']), (256, ['8,32', '16,16']), (32, ['4,8', '8,4'])):
    for cfg in cfgs:
        try:
            t, r = time_mala(65536, d, 100, cfg)
            print(f'mala n=65536 d={d} K=100 cfg={cfg}: {t*1e3:.3f} ms  {r/1e9:.3f} G/s', flush=True)
        except Exception as e:
            print('skip', d, cfg, e)
for cfg in ['8,32', '16,16']:
    t, r = time_hmc(32768, 256, 5, 20, cfg)
    print(f'hmc n=32768 d=256 K=5 L=20 cfg={cfg}: {t*1e3:.3f} ms  {r/1e6:.2f} M traj/s', flush=True)
for cfg in ['8,16', '16,8']:
    t, r = time_hmc(65536, 128, 5, 10, cfg)
    print(f'hmc n=65536 d=128 K=5 L=10 cfg={cfg}: {t*1e3:.3f} ms  {r/1e6:.2f} M traj/s', flush=True)
for (n, d, K, nh) in [(65536, 64, 1, None), (8192, 64, 100, None), (32768, 256, 1, None)]:
    t, r = time_flow_mh(n, d, K, nh)
    print(f'flow_mh n={n} d={d} K={K} H={nh}: {t*1e3:.3f} ms  {r/1e6:.2f} M chain-steps/s', flush=True)
