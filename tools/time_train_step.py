#!/usr/bin/env python3
"""Time one DSM optimiser step of train.TrainNet on synthetic data: python3 tools/time_train_step.py
[batch] [steps] [graph].  Used under rocprofv3 for profiles/r02_train_step_*."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from score_based_channels_amd.train import TrainNet
from score_based_channels_amd.train_score import fresh_state_dict, training_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
graph = len(sys.argv) > 3 and sys.argv[3] == 'graph'
cfg = training_config('CDL-C')
net = TrainNet(cfg, batch=B)
net.load_state_dict(fresh_state_dict(cfg, 0))
x = np.random.default_rng(0).standard_normal((B, 2, 64, 16)).astype(np.float32)
lab = torch.randint(0, 2311, (B,))
with torch.cuda.stream(torch.cuda.Stream()):
    for _ in range(3):
        net.step(x, lab, use_graph=graph)
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(steps):
        net.step(x, lab, use_graph=graph)
    torch.cuda.synchronize()
    dt = (time.time() - t) / steps
if not graph:
    from score_based_channels_amd import plan as P
    names = {getattr(P, k): k for k in ('BEGIN_CONV', 'INORM_STATS', 'CONV', 'MAXPOOL5', 'END_CONV', 'STEP_INC', 'DSM_PERTURB',
                                        'DSM_LOSS', 'GRAD_ADD', 'INORM_BWD', 'MAXPOOL5_BWD', 'UPSAMPLE_BWD', 'POOL_BWD',
                                        'CONV_WGRAD', 'PACK_WEIGHT', 'END_CONV_BWD', 'BEGIN_CONV_BWD', 'ADAM_EMA')}
    prof = net.profile_step(x, lab)
    tot = sum(v[0] for v in prof.values())
    for tag, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        print('%-8s %-15s %4d launches %8.3f ms  %5.1f %%' % ('forward' if tag < 200 else 'reverse', names.get(tag % 100, tag), n, ms, 100 * ms / tot))
    print('sum of kernel time %.2f ms' % tot)
print('batch %d, %s: %.2f ms per optimiser step = %.0f samples/s' % (B, 'graph' if graph else 'eager', dt * 1e3, B / dt))
