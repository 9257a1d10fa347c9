import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_mala import time_flow_mh
for gc in ['256', '512', '768', '1024', '2048']:
    os.environ['NFMC_FLOWB_GRID'] = gc
    for (n, d, K) in [(65536, 64, 1), (32768, 256, 1), (8192, 64, 100)]:
        t, r = time_flow_mh(n, d, K, None, reps=5)
        print(f'flow_mh grid<={gc} n={n} d={d} K={K}: {t*1e3:.3f} ms', flush=True)
