#!/usr/bin/env python
"""Headline benchmark: chain-steps/s of jump_mala + RealNVP on a synthetic Gaussian target (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W [--config C3|C2|C4|C5] [--reps R] [--fit-nf]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Launching.  Under torchrun (WORLD_SIZE set) this process is one rank.  Without it and with --gpus N > 1 the
parent -- BEFORE any GPU call; it never touches the GPU itself -- starts N fresh child processes of this same file,
one per device (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment), relays
rank 0's JSON line and exits with the children's status.  Ranks rendezvous over RCCL (`nccl` backend);
`--backend gloo --rehearse` runs the same launcher / rendezvous / barrier / max-over-ranks code without any GPU
work (what the CPU test of the launcher uses).

Workload (default C3 = BASELINE.json configs[2]): strategy jump_mala, U = sum x^2, d = 64,
65536 chains PER GPU (weak scaling: global chain ids continue across ranks), 100 inner MALA transitions per jump
(h = 64^(-1/3) = 0.25, unit mass), adjusted jumps with the build's default RealNVP (weights seed 1) FITTED once before
any timing by `Flow.fit` on draws of the target -- the state the reference's warmup=True leaves (`--unfitted-flow` keeps the
random initialisation of rounds 1-2, which accepts ~0.2 % of the jumps; same arithmetic per transition either way),
store_samples=False, x0 ~ N(0, I) from torch.manual_seed(0) on the CPU, uploaded before the timed region.  One bench
"step" = one outer iteration = 101 Markov transitions of every chain.  W untimed warm-up steps carry the state from
x0 to stationarity; then R (>= 5) timed repetitions of EXACTLY K steps each, every repetition bracketed by barrier +
synchronize on both sides and reduced by MAX over ranks; `value` / `ms_per_step` come from the MEDIAN repetition
(median of >= 5 runs), all repetitions are listed in `rep_ms`.  One further repetition of the same K
steps carries HIP events around every launch of the dominant kernel (`events_rep_ms`; the events cost 3-4 % of a
step, so it is not one of the R).
  value = n_chains_total * transitions_per_step * K / t_median.

roofline.  The dominant kernel of C3 is `mala_kernel`: it keeps the state in VGPRs for the 100 transitions of a
launch, so its HBM traffic is ~1 % of the per-transition algorithmic bytes of SURVEY 8d and the roof that BINDS is
the fp32 vector pipe (Philox4x32-10 + Box-Muller + the MALA arithmetic).  So:
  bound     "valu"
  achieved  ALGORITHMIC TFLOP/s: SURVEY 8d's flops per chain-transition (30*d for a Gaussian MALA step; `algorithmic_flops`
            below states the figure of every config) x the chain-transitions of one launch / the mean launch duration
            measured live here with HIP events on the launch stream
  peak      157.3 TFLOP/s, the fp32 vector peak of MI355X_MICROARCH.md (1024 SIMD-32 x 32 lanes x 2 flop x 2.4 GHz:
            one wave64 `v_fma_f32` per SIMD every 2 cycles, no packing)
  frac      achieved / peak
  valu_issue          VALU wave-instructions per second (SQ_INSTS_VALU per launch from the committed rocprofv3 --pmc
                      pass of this same command / live launch time) vs 1024 SIMDs x 2.4 GHz / 2 cycles = 1228.8 G/s
  valu_cost_weighted  the same instructions priced by class: sum over the PMC instruction classes (ADD/MUL/FMA/TRANS_F32,
                      INT32, INT64, CVT, other) of count x the measured issue time of that class on gfx950
                      (tools/ubench.hip, profiles/r03_instruction_cost_ubench.txt), / 1024 SIMDs / launch time: the
                      share of the launch that is pure issue of THIS instruction mix (1 = nothing left but a cheaper mix)
  valu_busy_frac_pmc  SQ_ACTIVE_INST_VALU / 4 per CU vs GRBM_GUI_ACTIVE: the pipe's own busy counter
  hbm_algorithmic     SURVEY 8d's 8*d bytes per chain-transition x transitions per launch / launch time vs 8 TB/s --
                      can exceed 1 for a kernel that fuses K transitions per launch; reported, not the bound
  hbm_real_frac       PMC-measured HBM bytes per launch / launch time / 8 TB/s
  pmc_stale           true when the committed PMC summary was profiled on another build of the library than the one
                      loaded now (source digests differ): the counter-derived fields are then from other code
C4's dominant kernel (`neutra_leapfrog_mfma_kernel`) is bound by fp32 MFMA: achieved TFLOP/s of conditioner GEMMs
(sustained: measured over the whole timed region; flops = the algorithm's minimum, C4_MACS_PER_GRADIENT) vs 157.3.

parity.  moments / acceptance of the timed run vs the analytic N(0, I/2), and -- north_star's "vs the CPU path on
the same seeds" -- a leg that runs ONE outer iteration of the first 8192 chains on the GPU and through the CPU
oracle (oracle/samplers.py, the reference's op sequence) fed the same x0 and the same Philox streams, and reports
the differences of first / second moments and of the acceptance rates.

cpu_baseline: the CPU oracle (eager PyTorch + autograd, torch's own generator like the reference) on a bounded
sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import gc
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
FP32_PEAK_TFLOPS = 157.3
# G wave64-instructions / s: 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md: `v_fma_f32`
# wave64 = 2 cycles of throughput, the rate behind the 157.3 TFLOP/s vector peak), 2.4 GHz
VALU_PEAK_GINST = 1024 * 2.4 / 2.0
N_SIMD = 1024
# issue time per wave-instruction and SIMD at full occupancy on gfx950, ns, by PMC instruction class
# (profiles/r03_instruction_cost_ubench.txt, tools/ubench.hip; the class's representative instructions in the sampler
# kernels: v_add_f32 / v_mul_f32 / v_fma(c)_f32 / v_log,v_sqrt,v_sin,v_cos,v_exp,v_rcp / v_bitop3,v_add_u32,v_and_or /
# v_mad_u64_u32 / v_cvt_f32_u32 / v_cndmask (VOP3), v_mov, DPP moves)
CLASS_NS = {'ADD_F32': 1.02, 'MUL_F32': 0.96, 'FMA_F32': 1.47, 'TRANS_F32': 3.40, 'INT32': 1.30, 'INT64': 2.09,
            'CVT': 1.73, 'OTHER': 1.75}
# (OTHER = everything SQ_INSTS_VALU counts outside the seven class counters: in the sampler kernels mostly v_bitop3_b32
# 1.63-1.85 ns -- the class counters do not count it as INT32 --, v_cndmask_b32 (VOP3) 1.82-1.88, v_mov_b32 1.02, DPP adds 1.79)

# ---------------------------------------------------------------------------------------------------- workloads
CONFIGS = {
    'C3': dict(strategy='jump_mala', d=64, n_per_gpu=65536, inner=100, transitions_per_step=101,
               kernel='mala_kernel', label='mala_steps', bound='valu',
               workload='BASELINE configs[2]: jump_mala + realnvp, U=sum x^2 (N(0, I/2)), d=64, 65536 chains per GPU, '
                        '100 MALA transitions per jump, step 0.25',
               metric='chain-steps/sec (n_chains x iters / s), jump_mala + RealNVP, d=64'),
    'C2': dict(strategy='imh', d=64, n_per_gpu=8192, inner=50, transitions_per_step=50,
               kernel='imh_eval_kernel', label='imh_parallel', bound='valu',
               workload='BASELINE configs[1]: imh + realnvp, U=sum x^2, d=64, 8192 '
                        'chains per GPU; one bench step = 50 independence-MH transitions (20 steps = SURVEY C2\'s T=1000)',
               metric='chain-steps/sec (n_chains x iters / s), imh + RealNVP, d=64'),
    'C4': dict(strategy='neutra_hmc', d=128, n_per_gpu=65536, inner=1, transitions_per_step=1,
               kernel='neutra_leapfrog_mfma_kernel', label='neutra_hmc_steps', bound='mfma',
               workload='BASELINE configs[3]: neutra_hmc + realnvp (conditioner 128 x 2), funnel potential d=128, 65536 '
                        'chains per GPU, L=10 leapfrog steps; one bench step = one trajectory',
               metric='chain-steps/sec (n_chains x iters / s), neutra_hmc + RealNVP, funnel d=128, L=10'),
    'C5': dict(strategy='jump_hmc', d=256, n_per_gpu=32768, inner=5, transitions_per_step=6,
               kernel='hmc_kernel', label='hmc_steps', bound='valu',
               workload='BASELINE configs[4]: jump_hmc + realnvp, U=sum x^2, d=256, 32768 chains per GPU (262144 over 8), '
                        'K=5 HMC trajectories of L=20 (h=0.05) per jump',
               metric='chain-steps/sec (n_chains x iters / s), jump_hmc + RealNVP, d=256'),
}
# per chain and gradient of the adjusted potential: 2 couplings x (the conditioner's 3 GEMMs of the inverse sweep, 40960
# multiply-adds at d = 128, H = 128 x 2, + their 3 transposed products in the reverse sweep).  This is the minimum the
# algorithm needs (SURVEY 8d's 'RealNVP pass' formula, backward with respect to the input = one more pass) and, since
# the reverse sweep reads its activations back from checkpoints, also what the kernel executes (round 1 recomputed the
# activations: 262144 executed against 245760 counted).
C4_MACS_PER_GRADIENT = 163840
PARITY_ROWS = 8192
PMC_STEPS = 3                     # --steps of the PMC passes in tools/profile_bench.sh


FLOW_STATE = {}   # config name -> state_dict of the jump configs' FITTED flow (fitted_flow_state), shared by every leg


def fitted_flow_state(cfg_name, cfg, dev):
    """The proposal flow of the jump configs in the state the reference's `warmup=True` leaves it in: the default RealNVP
    (weights from torch.manual_seed(1)) fitted by maximum likelihood (`Flow.fit`, jump.py:139-149; on the device:
    csrc/fit_kernels.hip) to 4096 + 1024 draws of the target N(0, I/2) from a fixed generator -- ONCE, outside every timed
    region.  An unfitted flow accepts ~0.2 % of the jumps; the arithmetic per transition is the same either way."""
    import torch
    from nfmc_amd.flows import Flow, RealNVP
    if cfg_name not in FLOW_STATE and cfg['strategy'] == 'neutra_hmc':
        # C4: the state NeuTra's warmup leaves (neutra.py:70-107): a variational (reverse-KL) fit of the flow to the funnel,
        # 200 epochs of 1024 latents, on the device (the conditioner is 128 wide: the matrix-core fit kernel, csrc/fit_mfma.hip)
        from nfmc_amd.potentials import Funnel
        d = cfg['d']
        pot = Funnel((d,), 3.0)
        torch.manual_seed(1)
        f = Flow(RealNVP((d,), conditioner_kwargs={'n_hidden': 128, 'n_layers': 2})).to(dev)
        f.variational_fit(lambda v: -pot(v), n_epochs=200, lr=0.01, n_samples=1024, early_stopping=False,
                          keep_best_weights=True, show_progress=False, potential=pot)
        FLOW_STATE[cfg_name] = {k: v.detach().cpu().clone() for k, v in f.state_dict().items()}
    if cfg_name not in FLOW_STATE and cfg['strategy'] == 'imh':
        # C2: the independence sampler's own warmup (imh.py:60-75): a variational fit of the default RealNVP to the target,
        # on the device (fit kernels), 300 epochs of 1024 latents
        from nfmc_amd.potentials import SumOfSquares
        d = cfg['d']
        pot = SumOfSquares((d,))
        torch.manual_seed(1)
        f = Flow(RealNVP((d,))).to(dev)
        f.variational_fit(lambda v: -pot(v), n_epochs=300, lr=0.02, n_samples=1024, early_stopping=False,
                          keep_best_weights=True, show_progress=False, potential=pot)
        FLOW_STATE[cfg_name] = {k: v.detach().cpu().clone() for k, v in f.state_dict().items()}
    if cfg_name not in FLOW_STATE:
        d = cfg['d']
        torch.manual_seed(1)
        f = Flow(RealNVP((d,))).to(dev)
        gen = torch.Generator().manual_seed(7)
        xt = (torch.randn(4096, d, generator=gen) * 0.7071067811865476).to(dev)
        xv = (torch.randn(1024, d, generator=gen) * 0.7071067811865476).to(dev)
        f.fit(xt, x_val=xv, n_epochs=300, lr=0.02, early_stopping=True, early_stopping_threshold=30, keep_best_weights=True,
              show_progress=False)
        FLOW_STATE[cfg_name] = {k: v.detach().cpu().clone() for k, v in f.state_dict().items()}
    return FLOW_STATE[cfg_name]


def neutra_step_size(flow_state):
    """C4's leapfrog step: 0.3 in the latent space of the FITTED flow (acceptance ~0.88: tools/probe_c4_fitted.py); 0.02 with
    the random initialisation of rounds 1-2, which also needs the chains started at 0.5 x0 not to reject everything."""
    return 0.3 if flow_state is not None else 0.02


def build_sampler(cfg, n_steps, fit_nf=False, flow_seed=1, flow_state=None):
    """The sampler of one repetition: K bench steps.  Flow weights from torch.manual_seed(flow_seed), or `flow_state`."""
    import torch
    from nfmc_amd.potentials import Funnel, SumOfSquares
    from nfmc_amd.sample import create_sampler
    d = cfg['d']
    torch.manual_seed(flow_seed)
    st = cfg['strategy']
    if st in ('jump_mala', 'jump_hmc'):
        pk = {'n_iterations': n_steps, 'store_samples': False}
        if fit_nf:
            pk.update(fit_nf=True, n_jumps_before_training=0, flow_fit_kwargs={'n_epochs': 2, 'show_progress': False})
        kw = {'inner_kernel_kwargs': {'n_leapfrog_steps': 20, 'step_size': 0.05}} if st == 'jump_hmc' else {}
        s = create_sampler(SumOfSquares((d,)), strategy=st, flow='realnvp', param_kwargs=pk,
                           inner_param_kwargs={'n_iterations': cfg['inner']}, **kw)
        if flow_state is not None:
            s.kernel.flow.load_state_dict(flow_state)
        return s
    if st == 'imh':
        s = create_sampler(SumOfSquares((d,)), strategy=st, flow='realnvp',
                           param_kwargs={'n_iterations': n_steps * cfg['inner'], 'store_samples': False})
        if flow_state is not None:
            s.kernel.flow.load_state_dict(flow_state)
        else:
            _match_scale_(s.kernel.flow)
        return s
    if st == 'neutra_hmc':
        s = create_sampler(Funnel((d,), 3.0), strategy=st, flow='realnvp',
                           flow_kwargs={'conditioner_kwargs': {'n_hidden': 128, 'n_layers': 2}},
                           inner_kernel_kwargs={'n_leapfrog_steps': 10, 'step_size': neutra_step_size(flow_state)},
                           param_kwargs={'n_iterations': n_steps, 'store_samples': False})
        if flow_state is not None:
            s.kernel.flow.load_state_dict(flow_state)
        return s
    raise ValueError(st)


def _match_scale_(flow, std=0.7071067811865476):
    """C2: an independence sampler whose proposal is an UNFITTED flow accepts ~1 % at d = 64 and measures a chain that
    does not move.  The default-initialised RealNVP is a small random perturbation of the identity map, so shifting its
    first ElementwiseAffine log-scale by -log(std) puts the proposal close to N(0, std^2 I) = the target N(0, I/2):
    roughly the state of a fitted flow, same arithmetic per transition (SURVEY C2's 'perturbed-init variant').  Works
    on the package's and the oracle's flow."""
    import math
    import torch
    with torch.no_grad():
        flow.bijection.layers[0].log_scale.add_(-math.log(std))
        d = int(flow.bijection.layers[0].log_scale.numel())
        for p in flow.parameters():          # the couplings' output layers: a tenth of their random initial size
            if p.dim() == 2 and p.shape[0] == 2 * (d - d // 2):
                p.mul_(0.1)
    return flow


def initial_state(cfg, n_total):
    """x0 ~ N(0, I), f32, from torch.manual_seed(0) on the CPU, so CPU and GPU legs share it.  Only the
    UNFITTED funnel run (--unfitted-flow) starts closer in (0.5 N(0, I)): an untrained flow + N(0, I) in 128 dimensions
    puts most chains where a step of 0.02 rejects, which measures nothing."""
    import torch
    gen = torch.Generator(device='cpu').manual_seed(0)
    x0 = torch.randn(n_total, cfg['d'], generator=gen)
    return 0.5 * x0 if (cfg['strategy'] == 'neutra_hmc' and cfg.get('_flow_state') is None) else x0


# ---------------------------------------------------------------------------------------------------- CPU legs
def _oracle_flow(cfg):
    import torch
    from oracle import flow as oflow
    torch.manual_seed(1)
    if cfg['strategy'] == 'neutra_hmc':
        f = oflow.Flow(oflow.RealNVP((cfg['d'],), conditioner_kwargs={'n_hidden': 128, 'n_layers': 2}))
        if cfg.get('_flow_state') is not None:
            f.load_state_dict(cfg['_flow_state'])
        return f
    f = oflow.Flow(oflow.RealNVP((cfg['d'],)))
    if cfg.get('_flow_state') is not None:
        f.load_state_dict(cfg['_flow_state'])
        return f
    return _match_scale_(f) if cfg['strategy'] == 'imh' else f


def _oracle_run(cfg, x0, flow, n_steps, noise=None):
    """`n_steps` bench steps of the workload through the CPU oracle; returns (Trace, transitions per chain)."""
    from oracle import potentials as opot
    from oracle import samplers as osamp
    d, st = cfg['d'], cfg['strategy']
    if st == 'jump_mala':
        return osamp.jump_sample(x0, opot.sum_squares, flow, 'langevin', n_steps, cfg['inner'], d ** (-1 / 3),
                                 noise=noise, store=False), n_steps * (cfg['inner'] + 1)
    if st == 'jump_hmc':
        return osamp.jump_sample(x0, opot.sum_squares, flow, 'hmc', n_steps, cfg['inner'], 0.05, n_leapfrog=20,
                                 noise=noise, store=False), n_steps * (cfg['inner'] + 1)
    if st == 'imh':
        return osamp.imh_sample(x0, opot.sum_squares, flow, n_steps * cfg['inner'], noise=noise, store=False), \
            n_steps * cfg['inner']
    return osamp.neutra_hmc_sample(x0, opot.funnel(3.0), flow, n_steps, neutra_step_size(cfg.get('_flow_state')), None, 10,
                                   noise=noise), n_steps


def cpu_baseline(cfg):
    """A bounded sample of the same workload on the host cores with the CPU oracle (port of the reference: eager
    PyTorch ops + autograd, torch's generator).  Eager PyTorch on 16 MiB tensors scales badly past a few dozen
    threads (measured on the bench box: 2.2e6 chain-steps/s at 16-32 threads, 5.8e5 at 128), so the thread count is
    calibrated on a short run first; the best one is used and reported as `cores`."""
    import torch
    flow = _oracle_flow(cfg)
    n = cfg['n_per_gpu']
    n_steps = 4 if cfg['strategy'] == 'imh' else 1     # bench steps in the sample: a few seconds of CPU work per config
    x0 = initial_state(cfg, n)
    small = dict(cfg, inner=max(1, cfg['inner'] // 25))
    _oracle_run(small, x0[:1024], flow, 1)   # warm
    avail = torch.get_num_threads()
    best_t, best_dt = avail, None
    for nt in sorted({t for t in (8, 16, 32, 64, avail) if t <= avail}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        _oracle_run(small, x0, flow, 1)
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = nt, dt
    torch.set_num_threads(best_t)
    t0 = time.perf_counter()
    _tr, transitions = _oracle_run(cfg, x0, flow, n_steps)
    dt = time.perf_counter() - t0
    torch.set_num_threads(avail)
    return {'value': n * transitions / dt, 'unit': 'chain-steps/s', 'cores': best_t, 'kind': 'port',
            'sample': f'{n_steps} bench step(s) ({transitions} transitions) of {n} chains, d={cfg["d"]}, oracle/samplers.py in '
                      f'{dt:.1f} s on {best_t} threads (best of 8/16/32/64/{avail} on a short calibration run)'}


def parity_vs_oracle(cfg, dev):
    """north_star: 'match the reference CPU path on identical seeds ... first/second moments and acceptance rate'.
    ONE bench step of the first PARITY_ROWS chains: the GPU path and the CPU oracle get the same x0 rows and the
    same Philox streams (oracle/philox.py is the executable spec of the kernels' generator), so the two runs are the
    same Markov chains up to fp32 rounding (chains whose accept test lands within rounding of a tie may part)."""
    import torch
    from oracle import samplers as osamp
    n, seed = PARITY_ROWS, 123
    x0 = initial_state(cfg, n)
    s = build_sampler(cfg, 1, flow_state=cfg.get('_flow_state'))
    s.seed = seed
    out = s.sample(x0.to(dev), show_progress=False)
    tr, transitions = _oracle_run(cfg, x0, _oracle_flow(cfg), 1, noise=osamp.PhiloxNoise(seed))
    m1, m2 = tr.moments.first, tr.moments.second
    st = out.statistics
    res = {'rows': n, 'transitions': transitions, 'seed': seed,
           'first_moment_abs_diff_max': float((out.mean - m1).abs().max()),
           'second_moment_rel_diff_max': float(((out.second_moment - m2).abs() / m2.abs().clamp_min(1e-6)).max()),
           'acceptance_gpu': st.acceptance_rate, 'acceptance_cpu': tr.n_accepted / max(1, tr.n_attempted),
           'last_state_agreeing_chains': float(((out.running_samples.last_sample.cpu() - tr.last).abs().amax(1) < 1e-3)
                                               .float().mean())}
    if hasattr(st, 'n_accepted_jumps'):
        res['jump_acceptance_gpu'] = st.jump_acceptance_rate
        res['jump_acceptance_cpu'] = tr.n_accepted_jumps / max(1, tr.n_attempted_jumps)
    return res


# ---------------------------------------------------------------------------------------------------- launcher
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """Parent of `python bench.py --gpus N` (N > 1, no torchrun): N children, one per device.  Nothing here touches the
    GPU (no HIP call, no torch.cuda query), and nothing is exec'ed from a process that did."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), NFMC_BENCH_CHILD='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=600))
        except subprocess.TimeoutExpired:
            p.kill()   # this exact child
            rcs.append(-9)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    sys.exit(bad[0] if bad else 0)


def _quiet_init(backend, rank, world, dev):
    """stdout carries exactly one JSON line: RCCL prints its init banner with printf, so file descriptor 1 points at
    stderr while the communicator is built."""
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        dist.barrier()
    finally:
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def _reduce_times(dt, world, dist, dev):
    """(max over ranks, [per-rank seconds]) of one repetition."""
    import torch
    if dist is None:
        return dt, [dt]
    if dist.get_backend() == 'gloo':
        dev = 'cpu'
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    allt = torch.empty(world, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allt, t)
    per = [float(v) for v in allt.cpu()]
    return max(per), per


def rehearse(args, rank, world):
    """The launcher / rendezvous / barrier / max-over-ranks path without GPU work (CPU test of --gpus N)."""
    import torch
    import torch.distributed as dist
    _quiet_init(args.backend, rank, world, None)
    reps = []
    for _ in range(args.reps):
        dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.002 * (rank + 1))
        dist.barrier()
        reps.append(_reduce_times(time.perf_counter() - t0, world, dist, 'cpu'))
    if rank == 0:
        print(json.dumps({'metric': CONFIGS[args.config]['metric'], 'value': None, 'unit': 'chain-steps/s',
                          'n_gpus': world, 'world_size_reported_by_backend': dist.get_world_size(),
                          'backend': dist.get_backend(), 'steps': args.steps, 'warmup': args.warmup, 'rehearsal': True,
                          'per_rank_ms': [1e3 * v for v in reps[len(reps) // 2][1]]}), flush=True)
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------- measurement
def _pmc(cfg_name, kernel):
    """Per-launch counter means of the dominant kernel from the committed rocprofv3 --pmc passes of this command, the
    summary's file name, and whether it is STALE: profiled on a library whose source digest (`nfmc_build_digest()`)
    differs from the one loaded now (None: the summary predates the digest)."""
    from nfmc_amd import hip
    for name in ('r04_%s_pmc_summary.json' % cfg_name.lower(), 'r03_%s_pmc_summary.json' % cfg_name.lower(),
                 'r02_%s_pmc_summary.json' % cfg_name.lower(),
                 'r02_bench_pmc_summary.json', 'r01_bench_pmc_summary.json'):
        p = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(p):
            try:
                doc = json.load(open(p))
                pm = doc.get(kernel)
            except Exception:
                doc, pm = {}, None
            if pm:
                dig = (doc.get('_meta') or {}).get('library_digest')
                stale = (dig != hip.build_digest()) if dig else None
                return pm, 'profiles/' + name, stale
    return None, None, None


def algorithmic_flops(cfg):
    """ALGORITHMIC flops per chain-transition of the dominant kernel's work ('ALGORITHMIC flops per
    chain-step'), per config.  d = event size.
      jump_mala  Gaussian MALA step: 30*d (SURVEY's figure: proposal, two potentials + gradients, two proposal densities).
      jump_hmc   one HMC trajectory on U = sum x^2 (hmc.py:61-77,96-126) in its minimal form -- the leapfrog with ONE
                 gradient evaluation per position (the reference evaluates the same gradient twice between two position
                 updates): L + 1 gradients (d multiplies each), 2 L + 1 axpys (2d each) = (5 L + 3) d; two Hamiltonians
                 (U: 2d, kinetic: 3d each) = 10d; the momentum draw's scaling d  ->  (5 L + 14) * d, L = 20.
      imh        one independence-MH transition: a RealNVP inverse pass, L_c * 2 * (d_a H + (n_hl - 1) H^2 + 2 H d_b) + 10 d
                 (SURVEY's 'RealNVP pass', default flow at d = 64: L_c = 2, H = 4, n_hl = 2), + U(x') 2d + base density 2d."""
    d, st = cfg['d'], cfg['strategy']
    if st == 'jump_mala':
        return 30.0 * d
    if st == 'jump_hmc':
        return (5.0 * 20 + 14.0) * d
    if st == 'imh':
        da, db, H, nhl, lc = d // 2, d - d // 2, 4, 2, 2
        return lc * 2.0 * (da * H + (nhl - 1) * H * H + 2 * H * db) + 10.0 * d + 4.0 * d
    return None


def _cost_weighted(pm, secs):
    """Share of the launch that is pure issue of the kernel's own instruction mix: per-class counts (PMC) x measured
    issue time per class (CLASS_NS) / 1024 SIMDs / launch time."""
    keys = ['ADD_F32', 'MUL_F32', 'FMA_F32', 'TRANS_F32', 'INT32', 'INT64', 'CVT']
    if not (pm and secs and all(('SQ_INSTS_VALU_' + k) in pm for k in keys) and pm.get('SQ_INSTS_VALU')):
        return None
    counts = {k: pm['SQ_INSTS_VALU_' + k] for k in keys}
    counts['OTHER'] = max(0.0, pm['SQ_INSTS_VALU'] - sum(counts.values()))
    ns = sum(counts[k] * CLASS_NS[k] for k in counts) / N_SIMD
    return {'wave_insts_per_launch_by_class': counts, 'ns_per_wave_inst_by_class': CLASS_NS,
            'issue_ms_per_launch': ns * 1e-6, 'frac_of_launch': ns * 1e-9 / secs,
            'mean_ns_per_wave_inst': ns * N_SIMD / pm['SQ_INSTS_VALU']}


def roofline(cfg_name, cfg, n_local, mean_ms, launches, transitions_per_launch):
    pm, src, stale = _pmc(cfg_name, cfg['kernel'])
    d = cfg['d']
    r = {'kernel': cfg['kernel'], 'mean_launch_ms': mean_ms, 'launches': launches, 'pmc_source': src, 'pmc_stale': stale}
    secs = mean_ms * 1e-3 if mean_ms else None
    traffic = None
    if pm and 'FETCH_SIZE' in pm and 'WRITE_SIZE' in pm:
        traffic = (2 * pm['FETCH_SIZE'] + pm['WRITE_SIZE']) * 1024   # gfx950 correction: MI355X_MICROARCH.md, HBM section
    if cfg['bound'] == 'mfma':
        flops = 2.0 * C4_MACS_PER_GRADIENT * n_local * 10   # one launch = one trajectory = L gradient evaluations
        ach = flops / secs / 1e12 if secs else None
        r.update(bound='mfma', achieved=ach, peak=FP32_PEAK_TFLOPS, unit='TFLOP/s',
                 frac=(ach / FP32_PEAK_TFLOPS) if ach else None, traffic=traffic,
                 algorithmic_flops_per_launch=flops,
                 hbm_real_frac=(traffic / secs / 1e9 / HBM_PEAK_GBS) if (traffic and secs) else None,
                 note='mean over EVERY launch of the timed repetitions (sustained clocks, not a from-idle burst); one '
                      'event pair brackets a whole nfmc_neutra_hmc_steps_f32 call (K trajectory launches + one '
                      'gradient launch + K statistics folds), so the per-launch mean is <= 2 % pessimistic')
        if pm and pm.get('SQ_VALU_MFMA_BUSY_CYCLES') and pm.get('GRBM_GUI_ACTIVE'):
            r['mfma_busy_frac'] = (pm['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024) / (pm['GRBM_GUI_ACTIVE'] / 8)
        return r
    per_launch_transitions = n_local * transitions_per_launch
    alg_bytes = (8 * d + (8 if cfg['strategy'] == 'imh' else 0)) * per_launch_transitions
    fl = algorithmic_flops(cfg) * per_launch_transitions
    ach = fl / secs / 1e12 if secs else None
    insts = pm.get('SQ_INSTS_VALU') if pm else None
    scale = 1.0
    if pm and cfg['strategy'] == 'imh':
        # one nfmc_imh_parallel_f32 call covers ALL steps of a run: the PMC passes (tools/profile_bench.sh: --steps 3)
        # counted 3 * 50 transitions per call, the live call has transitions_per_launch of them
        scale = transitions_per_launch / (PMC_STEPS * cfg['inner'])
        # every per-launch EXTENSIVE counter scales with the transitions of the call (instructions, bytes, busy cycles);
        # GRBM_GUI_ACTIVE too, so that the busy ratio stays the PMC run's
        pm = {k: (v * scale if isinstance(v, (int, float)) and (k.startswith('SQ_') or k in ('FETCH_SIZE', 'WRITE_SIZE',
                                                                                            'GRBM_GUI_ACTIVE')) else v)
              for k, v in pm.items()}
        insts = pm.get('SQ_INSTS_VALU')
        if 'FETCH_SIZE' in pm and 'WRITE_SIZE' in pm:
            traffic = (2 * pm['FETCH_SIZE'] + pm['WRITE_SIZE']) * 1024
    issue = insts / secs / 1e9 if (insts and secs) else None
    r.update(bound='valu', achieved=ach, peak=FP32_PEAK_TFLOPS, unit='TFLOP/s',
             frac=(ach / FP32_PEAK_TFLOPS) if ach else None, traffic=traffic,
             algorithmic_flops_per_launch=fl, algorithmic_flops_per_chain_transition=algorithmic_flops(cfg),
             valu_issue={'achieved_G_wave_inst_per_s': issue, 'peak_G_wave_inst_per_s': VALU_PEAK_GINST,
                         'frac': (issue / VALU_PEAK_GINST) if issue else None,
                         'wave_insts_per_launch': insts,
                         'insts_per_coordinate_transition': (insts * 64 / (per_launch_transitions * d)) if insts else None},
             valu_cost_weighted=_cost_weighted(pm, secs),
             hbm_algorithmic={'bytes_per_launch': alg_bytes,
                              'achieved_GBps': alg_bytes / secs / 1e9 if secs else None,
                              'frac_of_8TBps': alg_bytes / secs / 1e9 / HBM_PEAK_GBS if secs else None},
             hbm_real_frac=(traffic / secs / 1e9 / HBM_PEAK_GBS) if (traffic and secs) else None,
             note='state stays in VGPRs for the transitions of a launch: HBM traffic << algorithmic bytes; the kernel is '
                  'bound by the fp32 vector pipe, and most of its instructions are the noise generator (integer multiplies, '
                  'bit ops, transcendentals), which the algorithmic flop count does not contain: frac = algorithmic flops '
                  '/ 157.3 TF; valu_issue.frac = wave-instructions / (1024 SIMDs x 2.4 GHz / 2 cycles)')
    if pm and pm.get('SQ_ACTIVE_INST_VALU') and pm.get('GRBM_GUI_ACTIVE'):
        r['valu_busy_frac_pmc'] = (pm['SQ_ACTIVE_INST_VALU'] / 256) / (pm['GRBM_GUI_ACTIVE'] / 8)
    return r


def measure(name, args, env, reps, min_busy_s, philox7=True, cpu_legs=True, unfitted=False):
    """One config measured by the protocol of the module docstring; returns the JSON line's dict on rank 0 (None elsewhere).
    `reps` timed repetitions of exactly K steps at least, more of the same until the repetitions add up to `min_busy_s`
    seconds of back-to-back GPU work (so that an outside sampler of GPU utilisation sees the run); value = their median."""
    import torch
    dev, world, rank, dist, shard = env['dev'], env['world'], env['rank'], env['dist'], env['shard']
    distributed = dist is not None
    cfg = dict(CONFIGS[name])
    n_local = cfg['n_per_gpu']
    n_total = n_local * world
    cfg['_flow_state'] = None
    if not unfitted:
        cfg['_flow_state'] = fitted_flow_state(name, cfg, dev)   # deterministic: every rank fits the same flow
    x_start = initial_state(cfg, n_total).to(dev)   # resident in HBM before any timed region

    def run(n_steps, x, time_kernels=False, rounds=None):
        s = build_sampler(cfg, n_steps, fit_nf=args.fit_nf, flow_state=cfg['_flow_state'])
        s.seed = 0
        s.rng_rounds = args.rng_rounds if rounds is None else rounds
        s.shard = shard
        s.time_kernels = time_kernels
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = s.sample(x, show_progress=False)
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        return time.perf_counter() - t0, out, s

    def carried(out, x_prev):
        """The state after a run as the next run's x0 (global shape: every rank writes back its own block)."""
        last = out.running_samples.last_sample.reshape(-1, cfg['d'])
        if shard is None:
            return last
        lo, hi = shard.bounds(n_total)
        x_next = x_prev.clone()
        x_next[lo:hi] = last
        return x_next

    # one-time initialisation outside the timed region whatever W is: library / code-object load, allocator warm-up,
    # weight packing -- ONE step of the same path at the same per-GPU size (every launch a profiler sees has the same
    # shape, so per-kernel averages agree with the HIP-event mean below)
    prime = build_sampler(cfg, 1, flow_state=cfg['_flow_state'])
    prime.seed = 0
    prime.sample(x_start[:n_local], show_progress=False)
    torch.cuda.synchronize(dev)
    # CPython's cyclic collector must not fire inside a timed region (a generation-2 pass is ~35 ms with torch
    # imported, several times one repetition): collect now, keep it off while timing.  BEFORE the warm-up steps: with
    # the collection between warm-up and timing the GPU sat idle for those 35 ms, dropped its clocks, and the first
    # repetitions ran 5-12 % slower than the last whatever W was (W = 2 ... 1000 measured).
    gc.collect()
    gc.disable()
    if args.warmup > 0:   # untimed; moves the state from x0 ~ N(0, I) to stationarity
        _dt, wout, _s = run(args.warmup, x_start)
        x_start = carried(wout, x_start)
    label = cfg['label']
    # R (or more, see min_busy_s) repetitions WITHOUT events give `value` (an event pair per launch costs ~6 us of stream
    # time: 3-4 % of a C3 step, measured); one more repetition of the same K steps, same brackets, WITH HIP events on the
    # launch stream around every launch of the dominant kernel gives the roofline's launch duration (`events_rep_ms`)
    reps_dt, reps_rank, out, sampler = [], [], None, None
    busy = 0.0
    while len(reps_dt) < reps or busy < min_busy_s:   # every rank sees the same max-reduced times: same decision everywhere
        dt, out, sampler = run(args.steps, x_start)
        dt_max, per_rank = _reduce_times(dt, world, dist, dev)
        reps_dt.append(dt_max)
        reps_rank.append(per_rank)
        busy += dt_max
        if len(reps_dt) >= 100000:
            break
    events_rep = (None, [])
    if not args.no_kernel_events:
        dt, eout, _s = run(args.steps, x_start, time_kernels=label)
        dt_max, _pr = _reduce_times(dt, world, dist, dev)
        events_rep = (dt_max, [a.elapsed_time(b) for (l, a, b) in (getattr(eout, 'kernel_events', None) or []) if l == label])
    # the opt-in Philox4x32-7 stream next to the default (three repetitions, same protocol): a reported side figure
    alt = None
    if philox7 and args.rng_rounds == 10 and cfg['strategy'] in ('jump_mala', 'jump_hmc') and not args.fit_nf:
        alt_reps, ev_all7 = [], []
        for i7 in range(3 if args.no_kernel_events else 4):
            timed_launches = (i7 == 3)
            dt7, out7, _s = run(args.steps, x_start, time_kernels=label if timed_launches else False, rounds=7)
            dt7_max, _pr = _reduce_times(dt7, world, dist, dev)
            if timed_launches:
                ev_all7 = [a.elapsed_time(b) for (l, a, b) in (getattr(out7, 'kernel_events', None) or []) if l == label]
            else:
                alt_reps.append((dt7_max, out7))
        alt_reps.sort(key=lambda r_: r_[0])
        dt7, out7 = alt_reps[1]
        alt = {'rounds': 7, 'value': n_total * cfg['transitions_per_step'] * args.steps / dt7,
               'ms_per_step': dt7 / args.steps * 1e3, 'rep_ms': [1e3 * r_[0] for r_ in alt_reps],
               'mean_launch_ms': (sum(ev_all7) / len(ev_all7)) if ev_all7 else None,
               'variance_rel_err_max': float(((out7.variance - 0.5).abs() / 0.5).max()),
               'mean_abs_max': float(out7.mean.abs().max()), 'mcmc_acceptance': out7.statistics.acceptance_rate,
               'note': 'Philox4x32-7 (sample(..., rng_rounds=7)): the fewest rounds Random123 reports as passing BigCrush; '
                       'opt-in, never what `value` is quoted on'}
    gc.enable()
    if rank != 0:
        return None

    order = sorted(range(len(reps_dt)), key=lambda i: reps_dt[i])
    med = order[len(order) // 2]
    dt, per_rank = reps_dt[med], reps_rank[med]     # every repetition starts from the same state with the same seed:
    all_ev = list(events_rep[1])                    # `out` (the last one) is the outcome of each of them
    # what one HIP-event pair brackets: one launch of the inner kernel (C3: 100 MALA transitions, C5: 5
    # trajectories); for C4 one nfmc_neutra_hmc_steps_f32 call = K trajectory launches; for C2 one
    # nfmc_imh_parallel_f32 call = the three kernels of K*50 transitions (reported as ONE "launch")
    launches_per_event = args.steps if cfg['strategy'] == 'neutra_hmc' else 1
    mean_ms = (sum(all_ev) / len(all_ev) / launches_per_event) if all_ev else None
    n_launches = len(all_ev) * launches_per_event
    transitions_per_launch = cfg['inner'] * (args.steps if cfg['strategy'] == 'imh' else 1)
    st = out.statistics
    value = n_total * cfg['transitions_per_step'] * args.steps / dt
    rep_ms = [1e3 * v for v in reps_dt]
    line = {
        'metric': cfg['metric'], 'value': value, 'unit': 'chain-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': cfg['workload'], 'name': name, 'n_chains_total': n_total,
                   'n_chains_per_gpu': n_local, 'n_dim': cfg['d'], 'inner_steps': cfg['inner'],
                   'transitions_per_step': cfg['transitions_per_step'], 'store_samples': False,
                   'x0': 'N(0, I), torch.manual_seed(0) on the CPU, uploaded before timing; W warm-up steps carry it '
                         'to stationarity' + (' (x 0.5 for the unfitted funnel run)'
                                              if cfg['strategy'] == 'neutra_hmc' and cfg['_flow_state'] is None else ''),
                   'fit_nf': bool(args.fit_nf),
                   'proposal_flow': ('default RealNVP, weights seed 1' + (
                       ', fitted ONCE before timing (jump configs: Flow.fit, maximum likelihood, device path, on 4096 draws of the '
                       'target; imh: Flow.variational_fit on the device, 300 epochs; neutra_hmc: Flow.variational_fit, 200 '
                       'epochs, step size 0.3): the state warmup=True leaves'
                       if cfg['_flow_state'] is not None else
                       ' (unfitted)' if cfg['strategy'] != 'imh' else ' (unfitted, proposal scale matched: _match_scale_)')),
                   'sharding': f'chains x{world}, no data-path collective; one statistics all-reduce per sample()'
                               + ('; refit-buffer all-gather every outer iteration' if args.fit_nf else '')},
        'repetitions': len(rep_ms), 'rep_ms': rep_ms[:reps], 'rep_ms_median': statistics.median(rep_ms),
        'events_rep_ms': (1e3 * events_rep[0]) if events_rep[0] is not None else None,
        'rep_ms_min': min(rep_ms), 'rep_ms_max': max(rep_ms),
        'rep_ms_p10_p90': [sorted(rep_ms)[len(rep_ms) // 10], sorted(rep_ms)[(9 * len(rep_ms)) // 10]],
        'timed_busy_s': busy,
        'world_size_reported_by_backend': dist.get_world_size() if dist is not None else 1,
        'backend': dist.get_backend() if dist is not None else None,
        'per_rank_ms': [1e3 * v for v in per_rank],
        'roofline': roofline(name, cfg, n_local, mean_ms, n_launches, transitions_per_launch),
        'parity': {'mean_abs_max': float(out.mean.abs().max()), 'variance_mean': float(out.variance.mean()),
                   'mcmc_acceptance': st.acceptance_rate},
    }
    if cfg['strategy'] != 'neutra_hmc':   # U = sum x^2: N(0, I/2)
        line['parity']['variance_rel_err_max'] = float(((out.variance - 0.5).abs() / 0.5).max())
        line['parity']['second_moment_rel_err_max'] = float(((out.second_moment - 0.5).abs() / 0.5).max())
    else:
        # NeuTra's statistics are moments of the LATENT z (the reference's quirk); the target's own
        # marginal is checked on the chains' final states mapped to x = f^-1(z): funnel x_0 ~ N(0, 3^2)
        with torch.no_grad():
            z_last = out.running_samples.last_sample.reshape(-1, cfg['d']).to(dev)
            x_last = sampler.kernel.flow.bijection.inverse(z_last)[0]
            x0c = x_last[:, 0].double()
        line['parity']['funnel_x0_marginal'] = {
            'mean': float(x0c.mean()), 'var': float(x0c.var()), 'expected_mean': 0.0, 'expected_var': 9.0,
            'chains': int(x0c.numel()), 'transitions_from_x0': args.warmup + args.steps,
            'note': 'over the chains at the last state; the chains start at N(0, I), so the variance approaches 9 from below with W'}
    if hasattr(st, 'jump_acceptance_rate'):
        line['parity']['jump_acceptance'] = st.jump_acceptance_rate
    line['config']['rng'] = 'Philox4x32-%d, chain-id keyed (oracle/philox.py)' % args.rng_rounds
    if alt is not None:
        line['philox7'] = alt
    if world == 1 and cpu_legs:
        line['parity']['vs_cpu_oracle_same_seeds'] = parity_vs_oracle(cfg, dev)
    line['cpu_baseline'] = cpu_baseline(cfg) if (world == 1 and cpu_legs is True and not args.no_cpu_baseline) else None
    return line


def _brief(line):
    """What `other_configs` keeps of a config's line."""
    r = line['roofline']
    return {'metric': line['metric'], 'value': line['value'], 'unit': line['unit'], 'ms_per_step': line['ms_per_step'],
            'repetitions': line['repetitions'], 'rep_ms': line['rep_ms'], 'timed_busy_s': line['timed_busy_s'],
            'workload': line['config']['workload'],
            'roofline': {k: r.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'mean_launch_ms',
                                               'launches', 'pmc_source', 'pmc_stale', 'mfma_busy_frac', 'valu_busy_frac_pmc',
                                               'hbm_real_frac')},
            'parity': line['parity']}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=50,
                    help='untimed steps before the timed repetitions; the default covers the ~30 ms the clocks take to ramp '
                         'under load (with 2 the first two of five repetitions were 5-12 %% slower than the last)')
    ap.add_argument('--reps', type=int, default=5, help='timed repetitions of K steps at least; value = median (SURVEY 8d)')
    ap.add_argument('--min-busy-s', type=float, default=6.0,
                    help='keep repeating the K-step repetition until the timed repetitions add up to this many seconds of '
                         'back-to-back GPU work (the headline config; value stays the median of ALL repetitions): a 6 ms '
                         'repetition x 5 is invisible to a utilisation sampler')
    ap.add_argument('--config', choices=sorted(CONFIGS), default='C3')
    ap.add_argument('--fit-nf', action='store_true', help='jump strategies: refit the flow every outer iteration, so the '
                                                          'all-gather of the refit buffer (C1) is on the measured path')
    ap.add_argument('--rng-rounds', type=int, choices=[10, 7], default=10,
                    help='Philox4x32 rounds of the noise stream: 10 = the library default (what `value` is quoted on); 7 = the '
                         'opt-in stream.  With the default, C3 / C5 also time the 7-round stream and report it as `philox7`')
    ap.add_argument('--unfitted-flow', action='store_true',
                    help='jump configs: keep the randomly initialised proposal flow (round 1-2 behaviour: ~0.2 %% of the jumps '
                         'accepted) instead of fitting it once, before any timing, as `warmup=True` would')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='the default single-GPU C3 run also measures C2, C4 and the C5 shard (3 repetitions each, >= 1 s of GPU '
                         'work each, same protocol) and the unfitted-flow variants of C3 / C4, and attaches them to the ONE '
                         'JSON line as `other_configs` / `unfitted_flow`; this switch keeps the line to the headline config')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl')
    ap.add_argument('--rehearse', action='store_true', help='launcher / rendezvous / reduction only, no GPU work')
    ap.add_argument('--share-device', action='store_true',
                    help='(rehearsal on a one-GPU box, with --backend gloo) every rank computes on cuda:0: the whole N-rank '
                         'bench path -- launcher, sharding, collectives, reductions -- with real kernels; not a measurement')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-events', action='store_true', help='(experiment) no HIP events in the timed region')
    args = ap.parse_args()
    args.reps = max(1, args.reps)

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)   # does not return
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.rehearse:
        return rehearse(args, rank, world)

    import torch
    # NFMC_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, barriers, collectives) with one rank
    distributed = world > 1 or os.environ.get('NFMC_BENCH_FORCE_DIST') == '1'
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist
        _quiet_init(args.backend, rank, world, dev)
    from nfmc_amd.dist import Shard
    shard = Shard(rank=rank, world=world) if distributed else None
    env = {'dev': dev, 'world': world, 'rank': rank, 'dist': dist, 'shard': shard}

    line = measure(args.config, args, env, args.reps, args.min_busy_s, unfitted=args.unfitted_flow)
    extras = (world == 1 and args.config == 'C3' and not args.fit_nf and not args.unfitted_flow and not args.no_other_configs
              and args.rng_rounds == 10)
    if extras:
        # driver-visible evidence for every GPU config of BASELINE.json: same protocol, 3+ repetitions and
        # >= 1 s of GPU work each, roofline and the same-seed CPU-oracle leg; no CPU throughput baseline (the headline has it)
        # A failure in one of these must not cost the headline its line: it is recorded in place of the config's entry.
        others = {}
        for other in ('C2', 'C4', 'C5'):
            try:
                others[other] = _brief(measure(other, args, env, 3, 1.0, philox7=False, cpu_legs='parity'))
            except Exception as e:   # noqa: BLE001
                gc.enable()
                others[other] = {'error': repr(e)}
        line['other_configs'] = others
        # the round 1-2 workload next to the fitted one (the default changed; keep the figures comparable)
        unf = {}
        for other in ('C3', 'C4'):
            try:
                sub = measure(other, args, env, 3, 0.5, philox7=False, cpu_legs=False, unfitted=True)
                unf[other] = {'value': sub['value'], 'ms_per_step': sub['ms_per_step'], 'rep_ms': sub['rep_ms'],
                              'mcmc_acceptance': sub['parity']['mcmc_acceptance'],
                              'jump_acceptance': sub['parity'].get('jump_acceptance'),
                              'proposal_flow': sub['config']['proposal_flow']}
            except Exception as e:   # noqa: BLE001
                gc.enable()
                unf[other] = {'error': repr(e)}
        line['unfitted_flow'] = unf
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
