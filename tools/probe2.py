import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_mala import time_mala, time_hmc
print('lib', os.environ.get('NFMC_LIB', 'default'), flush=True)
for cfg in ['16,4', '8,8', '4,16']:
    for n in (65536, 1048576):
        t, r = time_mala(n, 64, 100, cfg)
        print(f'mala n={n} d=64 K=100 cfg={cfg}: {t*1e3:.3f} ms  {r/1e9:.3f} G chain-steps/s', flush=True)
t, r = time_hmc(32768, 256, 5, 20, None)
print(f'hmc n=32768 d=256 K=5 L=20: {t*1e3:.3f} ms  {r/1e6:.2f} M traj/s', flush=True)
