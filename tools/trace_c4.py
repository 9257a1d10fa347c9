"""Timeline of one workgroup of the matrix-core NeuTra trajectory kernel at the C4 shape, from a trace build:

    tools/build_mfma_variant.sh trace "-DNFMC_TRACE"
    NFMC_LIB=$PWD/nfmc_amd/libnfmc_hip.trace.so python tools/trace_c4.py

Workgroup 0 writes (id, s_memtime) marks per wave (mfma_device.hpp: WeightPipe::mark): 1 = a staging begins (the
previous GEMM phase ended), 2 = its copies are issued and stored, 3 = the barrier released, 10 / 11 / 12 = gradient
begins / inverse sweep done / gradient done, 15 = checkpoint loads + affine backward of a layer done, 20 = trajectory end.
Prints, per wave, the segments of ONE gradient in the middle of the trajectory and the totals by segment kind."""
import os
import sys
os.environ['NFMC_KEEP_SCRATCH'] = '1'
import collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = bench.CONFIGS['C4']
n = int(os.environ.get('TRACE_N', 65536))
dev = torch.device('cuda:0')
torch.manual_seed(0)
x0 = (0.5 * torch.randn(n, cfg['d'])).to(dev)
s = bench.build_sampler(cfg, 1)
s.sample(x0, show_progress=False)           # warm
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
s = bench.build_sampler(cfg, 1)
t0.record(); s.sample(x0, show_progress=False); t1.record(); torch.cuda.synchronize()
print('sample() of one trajectory: %.3f ms' % t0.elapsed_time(t1))
tail = s._scratch[-2 * 8 * (4096 + 8192):].view(torch.int64).cpu()
marks = tail[:8 * 4096].view(8, 4096)
steps = tail[8 * 4096:].view(8, 8192)
names = {(1, 2): 'stage copy', (2, 3): 'barrier wait', (3, 1): 'GEMM phase', (3, 11): 'GEMM phase (last of inverse sweep)',
         (3, 12): 'GEMM phase + EA (last of reverse sweep)', (3, 15): '??', (11, 15): 'potential grad + ckpt loads + affine bwd',
         (1, 15): 'x', (15, 1): 'affine->stage', (10, 1): 'EA inverse', (12, 10): 'leapfrog glue (momentum round trip)',
         (12, 20): 'tail'}
for w in (0, 4, 1, 5):
    ev = [(int(v) >> 48, int(v) & ((1 << 48) - 1)) for v in marks[w].tolist() if v != 0]
    if not ev:
        print('wave', w, 'no marks (not a trace build?)'); continue
    span = ev[-1][1] - ev[0][1]
    print('wave %d: %d marks, %d ticks first to last' % (w, len(ev), span))
    tot = collections.Counter(); cnt = collections.Counter()
    for (a, ta), (b, tb) in zip(ev, ev[1:]):
        tot[(a, b)] += tb - ta; cnt[(a, b)] += 1
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        print('   %-46s %8d ticks in %4d segments  (%.1f %%)' % (names.get(k, str(k)), v, cnt[k], 100.0 * v / span))
    # one gradient in the middle: from the 5th mark 10 to the following mark 12
    tens = [i for i, e in enumerate(ev) if e[0] == 10]
    if len(tens) >= 6 and w in (0, 4):
        i0 = tens[5]
        i1 = next(i for i in range(i0, len(ev)) if ev[i][0] == 12)
        print('   gradient 5:', ' '.join('%d:%d' % (ev[i][0], ev[i + 1][1] - ev[i][1]) for i in range(i0, i1)))

# per-step marks (a -DNFMC_TRACE_STEPS build): 30 = a step's reads begin, 31 = its MFMAs are issued (epilogue follows)
for w in (0, 4):
    ev = [(int(v) >> 48, int(v) & ((1 << 48) - 1)) for v in steps[w].tolist() if v != 0]
    if len(ev) < 400:
        continue
    # steps of gradient 5: 84 steps (168 marks) per gradient
    seg = ev[5 * 168:6 * 168 + 1]
    mf = [seg[i + 1][1] - seg[i][1] for i in range(0, len(seg) - 1, 2)]       # 30 -> 31: reads + MFMAs
    ep = [seg[i + 1][1] - seg[i][1] for i in range(1, len(seg) - 1, 2)]       # 31 -> next 30: epilogue (+ phase boundary)
    print('wave %d gradient 5, per step: reads+MFMAs %s' % (w, mf))
    print('wave %d gradient 5, per step: epilogue (+boundary) %s' % (w, ep))
