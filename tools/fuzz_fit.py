"""Randomised shapes through the device-fit gradient checks of tests/test_gpu_fit.py (one-off sweep on a GPU box; the
test suite pins a fixed list).  Every case is the test's own comparison -- loss and every parameter gradient of
nfmc_flow_fit_step_f32 / nfmc_flow_variational_fit_step_f32 against autograd of the CPU restatement -- at a shape drawn from
the supported domain: d 1..512, conditioner width 1..128 where a fit kernel exists, 1-2 hidden layers, 1-4 coupling layers,
1..6000 rows (>= 4096 rows switch the row kernels to four rows per wave).

usage: python tools/fuzz_fit.py [seed] [budget_seconds]
"""
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
    import test_gpu_fit as T
    from nfmc_amd.flow_training import DeviceFit
    from nfmc_amd.flows import Flow, RealNVP
    dev = torch.device('cuda', 0)
    rnd = random.Random(seed)
    t0 = time.time()
    done = failed = skipped = 0
    while time.time() - t0 < budget:
        d = rnd.choice([rnd.randint(1, 40), rnd.randint(41, 130), rnd.randint(131, 512), rnd.choice([64, 128, 256, 512, 255, 257])])
        H = rnd.choice([rnd.randint(1, 8), rnd.randint(1, 8), rnd.randint(9, 32), rnd.choice([33, 48, 64, 100, 128])])
        nhl, nl = rnd.randint(1, 2), rnd.randint(1, 4)
        n = rnd.choice([rnd.randint(1, 70), rnd.randint(71, 700), rnd.randint(701, 3000), rnd.randint(4096, 6000)])
        if d * n > 1.2e6:          # keeps the autograd leg of a case within a few seconds
            n = max(1, int(1.2e6 // d))
        nice = rnd.random() < 0.15
        kind = rnd.choice(['ml', 'ml', 'sum', 'diag', 'funnel'])
        if kind == 'funnel' and d < 2:
            kind = 'sum'
        probe = Flow(RealNVP((d,), n_layers=nl, conditioner_kwargs={'n_hidden': H, 'n_layers': nhl})).to(dev)
        if not DeviceFit.supported(probe.bijection, dev):
            skipped += 1
            continue
        case = 'd=%d H=%d nhl=%d nl=%d n=%d nice=%s kind=%s' % (d, H, nhl, nl, n, nice, kind)
        try:
            if H > 32:
                T.test_wide_conditioner_gradients_on_the_matrix_cores_match_autograd(dev, d, H, nhl, nl, n, kind, nice)
            elif kind == 'ml':
                T.test_nll_gradient_matches_autograd(dev, d, H, nhl, nl, n, nice)
            else:
                T.test_reverse_kl_gradient_matches_autograd(dev, d, H, nhl, nl, n, kind)
            done += 1
            print('ok    %s  (%.0f s)' % (case, time.time() - t0), flush=True)
        except Exception:
            failed += 1
            print('FAIL  %s' % case, flush=True)
            traceback.print_exc(limit=3)
    print('cases %d  failed %d  unsupported shapes skipped %d' % (done + failed, failed, skipped))
    return 1 if failed else 0


if __name__ == '__main__':
    sys.exit(main())
