"""Reproducer of the packed-fp32 co-execution hazard.  Victim: the begin convolution alone on stream A,
checked bit for bit against its own solo run.  Aggressor: ONE operator of the score plan launched 40 times on stream B.
With a library built WITH packed-fp32 instructions (tools/build_variant.sh slp <file>.hip -fslp-vectorize for every file, or
any build before round 2's Makefile change) the Winograd convolutions as aggressors corrupt the victim's v_pk_fma_f32 results
(low half of the register pair, 8-16 lanes of one pixel) in 10-60 % of the runs; the shipped build prints nothing."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, '.')
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.config import default_config
from score_based_channels_amd.scorenet import ScoreNet
from score_based_channels_amd.weights import seeded_state_dict
cfg = default_config(); sd = seeded_state_dict(cfg, 2024)
B = 48
res = {}
for mode in ('bf16x3', 'f32'):
    net = ScoreNet(cfg, conv_mode=mode).cuda().load_state_dict(sd)
    pl = net.score_plan(64, 16)
    a, b = net.bind(B, 64, 16), net.bind(B, 64, 16)
    x = torch.randn(B, 64, 16, 2, device='cuda', generator=torch.Generator('cuda').manual_seed(1)) * 3
    for bd in (a, b):
        bd.x.copy_(x); bd.labels.fill_(5)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    full = _lib.Plan(b.ops, keepalive=b); full.run(sb.cuda_stream, 1); torch.cuda.synchronize()      # b's buffers hold real data
    victim = _lib.Plan(a.ops[:1], keepalive=a)
    t = pl.ops[0].dst
    def vic(agg):
        torch.cuda.synchronize()
        if agg is not None: agg.run(sb.cuda_stream, 40)
        for _ in range(6):
            victim.run(sa.cuda_stream, 1)
        torch.cuda.synchronize()
        return a.slots[t.slot].clone()
    ref = vic(None)
    seen = set()
    for j, op in enumerate(pl.ops):
        key = (op.kind, op.src.h, op.src.c, op.dst.c, op.ksize, op.dil, op.flags & 0x20)
        if key in seen: continue
        seen.add(key)
        agg = _lib.Plan([b.ops[j]], keepalive=b)
        bad = sum(not torch.equal(ref, vic(agg)) for _ in range(10))
        agg.close()
        if bad: print(mode, 'aggressor op', j, op.name, 'kind', op.kind, (op.src.h, op.src.w, op.src.c), '->', op.dst.c, 'k', op.ksize, 'dil', op.dil, 'victim mismatches', bad, 'of 10', flush=True)
    print(mode, 'done')
