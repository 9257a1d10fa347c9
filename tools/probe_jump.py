"""The jump kernel alone (nfmc_flow_mh_steps_f32, one transition per launch): `python tools/probe_jump.py N D [REPS]`
launches it REPS times at the production settings (statistics on, log q not cached, adjusted).  Host-side launch cost
(~20 us of ctypes per call) exceeds the kernel for small shapes, so time it with the profiler, not with events:
tools/ab_jump.sh runs this under `rocprofv3 --kernel-trace --stats` for several libraries / knob settings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import hip
from nfmc_amd.flows import Flow, RealNVP
from nfmc_amd.potentials import SumOfSquares
from nfmc_amd.samplers.common import Run
from nfmc_amd.samplers.jump import launch_flow_mh


class _S:
    shard = None; seed = 0; replay = None; time_kernels = False


def main():
    n, d = int(sys.argv[1]), int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    dev = torch.device('cuda', 0)
    torch.manual_seed(1)
    flow = Flow(RealNVP((d,)))
    x0 = torch.randn(n, d, generator=torch.Generator().manual_seed(0)) * 0.7071
    run = Run(_S(), x0.to(dev))
    pot = SumOfSquares((d,))
    logq = torch.zeros(n, device=dev)
    for i in range(reps):
        launch_flow_mh(run, flow, pot, logq, 1, i, False, True, run.stats.struct(defer=True, attempted=n, jump=True))
    torch.cuda.synchronize()
    print('variance of the state %.4f' % float(run.x.var()))


if __name__ == '__main__':
    main()
