import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_mala import time_mala, time_hmc, time_flow_mh
for (n, d, K, nh) in [(65536, 64, 1, None), (8192, 64, 100, None), (32768, 256, 1, None), (65536, 64, 4, 8), (65536, 25, 4, None), (65536, 64, 4, 16)]:
    t, r = time_flow_mh(n, d, K, nh)
    print(f'flow_mh n={n} d={d} K={K} H={nh}: {t*1e3:.3f} ms  {r/1e6:.2f} M chain-steps/s', flush=True)
t, r = time_mala(65536, 64, 100, None)
print(f'mala n=65536 d=64 K=100: {t*1e3:.3f} ms  {r/1e9:.3f} G/s', flush=True)
