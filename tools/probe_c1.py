"""C1 (README example: jump_mala, d=25, 100 chains, 200 outer x (100 MALA + 1 jump)): wall time, with and without the
fused jump tail, and a host-side profile of one call (C1_PROFILE=1)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nfmc_amd import sample
from nfmc_amd.samplers.jump import JumpNFMC


def call(store=True):
    torch.manual_seed(0)
    return sample(lambda x: torch.sum(x ** 2, dim=1), event_shape=(25,), strategy='jump_mala', flow='realnvp',
                  n_chains=int(os.environ.get('C1_N', '100')), n_iterations=200, show_progress=False, seed=0,
                  param_kwargs={'store_samples': store})


def timed(label, **kw):
    best = None
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = call(**kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print('%-28s %.2f ms  var %.4f  jump acc %.4f' % (label, best * 1e3, float(out.variance.mean()),
                                                      out.statistics.jump_acceptance_rate), flush=True)


def main():
    timed('separate jump launch')
    timed('separate, no sample store', store=False)
    JumpNFMC.fuse_jump_tail = True
    timed('fused jump tail')
    timed('fused, no sample store', store=False)
    JumpNFMC.fuse_jump_tail = False
    if os.environ.get('C1_PROFILE') == '1':
        pr = cProfile.Profile()
        pr.enable()
        call()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(25)


if __name__ == '__main__':
    main()
