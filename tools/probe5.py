import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_mala import time_flow_mh
for cfg in [None, '8,8', '4,16']:
    if cfg: os.environ['NFMC_FLOWB_CFG'] = cfg
    for (n, d, K) in [(65536, 64, 1), (65536, 64, 8), (8192, 64, 100)]:
        t, r = time_flow_mh(n, d, K, None)
        print(f'flow_mh cfg={cfg} n={n} d={d} K={K}: {t*1e3:.3f} ms  {r/1e6:.1f} M chain-steps/s', flush=True)
for cfg in ['8,32', '4,64']:
    os.environ['NFMC_FLOWB_CFG'] = cfg
    t, r = time_flow_mh(32768, 256, 1, None)
    print(f'flow_mh cfg={cfg} n=32768 d=256 K=1: {t*1e3:.3f} ms  {r/1e6:.1f} M chain-steps/s', flush=True)
    t, r = time_flow_mh(32768, 256, 8, None)
    print(f'flow_mh cfg={cfg} n=32768 d=256 K=8: {t*1e3:.3f} ms  {r/1e6:.1f} M chain-steps/s', flush=True)
