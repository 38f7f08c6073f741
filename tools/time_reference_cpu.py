#!/usr/bin/env python3
"""SURVEY section 8(d) "CPU baseline": the REFERENCE's own path on this host's cores -- the imported ``NCSNv2Deepest``
(``/root/reference/ncsnv2/models/ncsnv2.py``) inside the transcription of the ``test_score.py:118-171`` loop that
``tests/gen_golden.py`` holds (the script itself cannot be imported: argparse and ``.cuda()`` at module level).

Build container only: it needs ``/root/reference``, which does not travel to the GPU box (this file is listed in
``.gpurunignore``).  B = 100 channels at one SNR point, ``--steps`` Langevin steps after 2 warm-up steps at noise level 0 (the
per-step cost does not depend on the level: the network is unconditional and the level only changes three scalars),
``torch.set_num_threads`` = every core.  Writes ``profiles/<round>_reference_cpu.json``; ``bench.py`` embeds that record as
``cpu_baseline.reference_build_container`` of its cdlc line.

    python tools/time_reference_cpu.py [--steps 20] [--round r06] [--reps 7]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=100)
    ap.add_argument('--round', default='r06')
    ap.add_argument('--reps', type=int, default=7)
    args = ap.parse_args()
    import numpy as np
    import torch
    import gen_golden as G                              # puts /root/reference on sys.path, imports the reference network
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.weights import seeded_state_dict
    n_threads = os.cpu_count() or 1
    torch.set_num_threads(n_threads)
    cfg = default_config('CDL-C')
    sd = seeded_state_dict(cfg, G.WEIGHT_SEED)
    net = G.reference_net(cfg, sd)
    H, P = G.case_inputs(1, args.batch, 64, 16, 0.6)
    steps_per_channel = 2311 * 3
    G.reference_ald(net, cfg, H, P, [0.0], [0], seed=1, steps_each=2)                      # warm-up (2 steps)
    reps = []
    for _ in range(args.reps):                          # (the container's cores are shared: the fastest repetition is reported)
        t0 = time.perf_counter()
        _, _, log = G.reference_ald(net, cfg, H, P, [0.0], [0], seed=1, steps_each=args.steps)
        reps.append((time.perf_counter() - t0) / args.steps)
        assert np.isfinite(log).all()
    dt, med = min(reps), float(np.median(reps))
    rec = {'value': args.batch / (steps_per_channel * dt), 'unit': 'channels/s', 'cores': n_threads, 'kind': 'reference',
           's_per_step': dt, 's_per_step_median': med, 'value_at_median': args.batch / (steps_per_channel * med),
           's_per_step_all_repetitions': [round(r, 4) for r in reps], 'load_average_1min': os.getloadavg()[0], 'batch': args.batch, 'steps_timed': args.steps,
           'torch': torch.__version__, 'host': 'build container (%d vCPU)' % n_threads,
           'what': 'the reference itself: imported NCSNv2Deepest (ncsnv2/models/ncsnv2.py) inside the transcription of '
                   'test_score.py:118-171 (tests/gen_golden.py: reference_ald), B = %d channels, 1 SNR point, %d timed Langevin '
                   'steps after 2 warm-up steps, torch.set_num_threads(%d), fastest of %d repetitions; per-step mean x 6933 steps (SURVEY '
                   'section 8(d); the survey\'s own probe on this container: 0.313 s per step = 0.046 channels/s)'
                   % (args.batch, args.steps, n_threads, args.reps),
           'command': 'python tools/time_reference_cpu.py --steps %d --batch %d' % (args.steps, args.batch)}
    out = os.path.join(ROOT, 'profiles', '%s_reference_cpu.json' % args.round)
    with open(out, 'w') as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
