#!/usr/bin/env python
"""Headline benchmark: chain-steps/s of jump_mala + RealNVP on a synthetic Gaussian target (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md section 8d "C3"): strategy jump_mala, U = sum x^2, d = 64,
65536 chains PER GPU (weak scaling: global chain ids continue across ranks), 100 inner MALA transitions per
jump (h = 64^(-1/3) = 0.25, unit mass), adjusted jumps with the build's default RealNVP (weights seed 1),
store_samples=False.  One bench "step" = one outer iteration = 101 Markov transitions of every chain.
metric value = n_chains_total * 101 * K / t, t = max over ranks of the wall time of `sampler.sample(x0)`
(x0 already resident in HBM) bracketed by barrier + synchronize.

roofline: the dominant kernel is `mala_kernel`.  achieved = (8*d bytes per chain-transition [SURVEY 8d] *
n_local * 100 transitions per launch) / mean launch duration from HIP events recorded on the launch stream
during the timed region.  The kernel keeps the state in registers for the 100 transitions of a launch, so its
real HBM traffic (`traffic`, PMC-measured, profiles/) is ~1% of the algorithmic figure and the kernel is
VALU-issue bound (profiles/: SQ_ACTIVE_INST_VALU ~ 100% of the launch); frac is quoted against the HBM roof
because that is the per-transition bound SURVEY 8d names -- frac > 1 means the fused kernel runs faster than
ANY kernel that streams the state once per transition could (it says nothing about being "over peak").

cpu_baseline: the CPU oracle (oracle/samplers.py: the reference's op sequence, eager PyTorch + autograd) on
one outer iteration of the same workload (65536 chains, 100 MALA + 1 jump), rank 0, N = 1 only.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = 64
N_PER_GPU = 65536
K_INNER = 100
HBM_PEAK_GBS = 8000.0


def build_sampler(n_outer, flow_seed=1):
    from nfmc_amd.sample import create_sampler
    from nfmc_amd.potentials import SumOfSquares
    torch.manual_seed(flow_seed)
    s = create_sampler(SumOfSquares((D,)), strategy='jump_mala', flow='realnvp',
                       param_kwargs={'n_iterations': n_outer, 'store_samples': False},
                       inner_param_kwargs={'n_iterations': K_INNER})
    return s


def cpu_baseline():
    """One outer iteration of the same workload on the host cores with the CPU oracle (port of the reference).
    Eager PyTorch on tensors of 16 MiB scales badly past a few dozen threads (measured on the bench box: 2.2e6
    chain-steps/s at 16-32 threads, 5.8e5 at 128), so the thread count is calibrated on a short run first and the
    best one is used and reported as `cores`."""
    from oracle import flow as oflow
    from oracle import potentials as opot
    from oracle import samplers as osamp
    torch.manual_seed(1)
    flow = oflow.Flow(oflow.RealNVP((D,)))
    torch.manual_seed(0)
    x0 = torch.randn(N_PER_GPU, D)
    h = D ** (-1 / 3)
    osamp.jump_sample(x0[:1024], opot.sum_squares, flow, 'langevin', 1, 5, h, store=False)  # warm
    avail = torch.get_num_threads()
    best_t, best_dt = avail, None
    for nt in sorted({t for t in (8, 16, 32, 64, avail) if t <= avail}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        osamp.jump_sample(x0, opot.sum_squares, flow, 'langevin', 1, 4, h, store=False)
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = nt, dt
    torch.set_num_threads(best_t)
    t0 = time.perf_counter()
    osamp.jump_sample(x0, opot.sum_squares, flow, 'langevin', 1, K_INNER, h, store=False)
    dt = time.perf_counter() - t0
    torch.set_num_threads(avail)
    return {'value': N_PER_GPU * (K_INNER + 1) / dt, 'unit': 'chain-steps/s', 'cores': best_t,
            'kind': 'port',
            'sample': f'1 outer iteration (100 MALA + 1 jump) of {N_PER_GPU} chains, d={D}, oracle/samplers.py '
                      f'jump_sample in {dt:.1f} s on {best_t} threads (best of 8/16/32/64/{avail} on a 4-step calibration)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='(experiment) no HIP events in the timed region')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # NFMC_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, barriers, all-reduce) with one rank (rehearsal)
    distributed = world > 1 or os.environ.get('NFMC_BENCH_FORCE_DIST') == '1'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # stdout carries exactly one JSON line: RCCL prints its init banner ("RCCL version : ...", debug level VERSION,
        # which the bench box sets) with printf, so file descriptor 1 points at stderr while the communicator is built
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
            dist.barrier()
        finally:
            import ctypes
            ctypes.CDLL(None).fflush(None)   # the banner sits in C stdio's buffer when stdout is a pipe
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from nfmc_amd.dist import Shard
    shard = Shard(rank=rank, world=world) if distributed else None

    n_total = N_PER_GPU * world
    gen = torch.Generator(device='cpu').manual_seed(0)
    x0 = (torch.randn(n_total, D, generator=gen) * 0.7071).to(dev)   # resident in HBM before the timed region

    def run(n_outer, time_kernels=False):
        s = build_sampler(n_outer)
        s.seed = 0
        s.shard = shard
        s.time_kernels = time_kernels
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = s.sample(x0, show_progress=False)
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        return time.perf_counter() - t0, out

    # one-time initialisation outside the timed region whatever W is: library load, code-object load of the two
    # kernels (first launch), allocator warm-up -- ONE outer step of the same path at the same per-GPU size, so that
    # every launch a profiler sees has the same shape (per-kernel averages agree with the HIP-event mean below)
    prime = build_sampler(1)
    prime.seed = 0
    prime.sample(x0[:N_PER_GPU], show_progress=False)
    torch.cuda.synchronize(dev)
    if args.warmup > 0:
        run(args.warmup)
    # CPython's cyclic collector must not fire inside the timed region: with torch imported a full (generation 2)
    # collection takes ~35 ms on this host -- five times the 20 timed steps -- and its trigger depends on allocation
    # counts, i.e. on luck.  Collect now, keep it off while timing (reference counting still frees everything the
    # run allocates), turn it back on afterwards.
    gc.collect()
    gc.disable()
    # HIP events only around the dominant kernel's launches: every event pair costs ~6 us of stream time
    dt, out = run(args.steps, time_kernels=False if args.no_kernel_events else 'mala_steps')
    gc.enable()
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant-kernel duration from the HIP events recorded on the launch stream during the timed region
    ev = [(l, a.elapsed_time(b)) for (l, a, b) in (out.kernel_events or [])]
    mala_ms = [ms for (l, ms) in ev if l == 'mala_steps']
    jump_ms = [ms for (l, ms) in ev if l == 'flow_mh_steps']
    mean_mala = sum(mala_ms) / max(1, len(mala_ms))
    alg_bytes_per_launch = 8 * D * N_PER_GPU * K_INNER
    achieved = alg_bytes_per_launch / (mean_mala * 1e-3) / 1e9 if mala_ms else None
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get('mala_kernel_hbm_bytes_per_launch')
        except Exception:
            traffic = None

    # what actually binds the kernel, from the committed PMC passes of this same command (profiles/): VALU wave
    # instructions per coordinate-transition and the share of the launch the VALU was issuing
    valu = None
    ppath = os.path.join(ROOT, 'profiles', 'r01_bench_pmc_summary.json')
    if os.path.exists(ppath):
        try:
            pm = json.load(open(ppath))['mala_kernel']
            valu = {'insts_per_coordinate_transition': pm['SQ_INSTS_VALU'] * 64 / (N_PER_GPU * D * K_INNER),
                    'busy_frac': (pm['SQ_ACTIVE_INST_VALU'] / 256) / (pm['GRBM_GUI_ACTIVE'] / 8),
                    'source': 'profiles/r01_bench_pmc_summary.json (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE)'}
        except Exception:
            valu = None

    if rank == 0:
        value = n_total * (K_INNER + 1) * args.steps / dt
        st = out.statistics
        line = {
            'metric': 'chain-steps/sec (n_chains x iters / s), jump_mala + RealNVP, d=64',
            'value': value, 'unit': 'chain-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[2]: jump_mala + realnvp, U=sum x^2 (N(0, I/2)), d=64, '
                                   '65536 chains per GPU, 100 MALA transitions per jump, step 0.25',
                       'n_chains_total': n_total, 'n_chains_per_gpu': N_PER_GPU, 'n_dim': D, 'inner_steps': K_INNER,
                       'transitions_per_step': K_INNER + 1, 'store_samples': False,
                       'sharding': f'chains x{world}, no data-path collective; one statistics all-reduce per sample()'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': (achieved / HBM_PEAK_GBS) if achieved else None, 'traffic': traffic,
                         'kernel': 'mala_kernel', 'mean_launch_ms': mean_mala, 'launches': len(mala_ms),
                         'algorithmic_bytes_per_launch': alg_bytes_per_launch, 'valu': valu,
                         'note': 'state stays in VGPRs for the 100 transitions of a launch: real HBM traffic (traffic) << '
                                 'algorithmic bytes; the kernel is VALU-issue bound (Philox4x32-10 + Box-Muller + MALA '
                                 'arithmetic), so frac vs the HBM roof can exceed 1',
                         'flow_mh_mean_launch_ms': (sum(jump_ms) / len(jump_ms)) if jump_ms else None},
            'parity': {'mean_abs_max': float(out.mean.abs().max()), 'variance_mean': float(out.variance.mean()),
                       'variance_rel_err_max': float(((out.variance - 0.5).abs() / 0.5).max()),
                       'mcmc_acceptance': st.acceptance_rate, 'jump_acceptance': st.jump_acceptance_rate},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
