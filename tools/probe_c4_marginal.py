"""C4's target is the funnel: x_0 ~ N(0, 3^2).  NeuTra's own statistics are moments of the latent z (the reference's quirk), so
the bench line reports the x-space marginal of the chains' LAST states (parity.funnel_x0_marginal) -- after W + K trajectories
from z ~ N(0, I), still far from stationarity.  This probe runs the same sampler (fitted flow, h = 0.3, L = 10) on 8192 chains
for longer and prints mean / variance of x_0 over the chains every 500 trajectories."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda', 0)
cfg = dict(bench.CONFIGS['C4'])
cfg['_flow_state'] = bench.fitted_flow_state('C4', cfg, dev)
n = 8192
z = bench.initial_state(cfg, n).to(dev)
total = 0
for block in range(8):
    s = bench.build_sampler(cfg, 500, flow_state=cfg['_flow_state'])
    s.seed = block
    out = s.sample(z, show_progress=False)
    z = out.running_samples.last_sample.reshape(n, -1).to(dev)
    total += 500
    with torch.no_grad():
        x = s.kernel.flow.bijection.inverse(z)[0]
    x0 = x[:, 0].double()
    rest = x[:, 1:].double()
    print('after %4d trajectories: x0 mean %+.3f var %.3f (target 0, 9)   acceptance %.3f   mean var of x_1.. %.2f' % (
        total, float(x0.mean()), float(x0.var()), out.statistics.acceptance_rate, float(rest.var(dim=0).mean())), flush=True)
