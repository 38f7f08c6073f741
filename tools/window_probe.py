#!/usr/bin/env python3
"""Where does a short timed window (the driver's 20 steps after 5 warm-up steps) lose time against the sustained rate?  Two sub-batch
streams as driver.run_concurrently runs them, but one plan.run(1) per step with an event behind each, so that the completion time of
every step of both streams is known.  GPU box only.   python tools/window_probe.py [steps=20] [lag=1]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import AldBatch, snr_to_noise
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.scorenet import ScoreNet
    from score_based_channels_amd.weights import seeded_state_dict
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    lag = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cfg = default_config('CDL-C')
    net = ScoreNet(cfg, 'cuda:0').load_state_dict(seeded_state_dict(cfg, 2024))
    nch, nt, nr = 100, 64, 16
    snr = np.arange(-10, 32.5, 2.5)
    raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=4321)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(1), nch, nt, 38), (0, 2, 1)))
    idx = np.tile(np.arange(nch), len(snr))
    ln = np.repeat(snr_to_noise(snr, nt), nch)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, device='cuda')
    alds, streams = [], []
    for part in np.array_split(np.arange(len(idx)), 2):
        a = AldBatch(net, H, Pm, idx[part], idx[part], ln[part], seed=1, traj_id=part)
        a.set_init(init[torch.from_numpy(idx[part])])
        a.synthesize_measurements()
        alds.append(a)
        streams.append(torch.cuda.Stream())
    half = torch.cuda.get_device_properties(0).multi_processor_count // 2
    for a in alds:
        a.set_persistent_cus(half)

    def window(n, use_lag):
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] for _ in alds]

        def work(k):
            with torch.cuda.stream(streams[k]):
                evs[k][0].record()
                if use_lag and k == 1:
                    alds[k]._cut_step()
                    alds[k]._lag_plan.run(streams[k].cuda_stream, 1, False)
                for s in range(n):
                    alds[k].plan.run(streams[k].cuda_stream, 1, False)
                    evs[k][s + 1].record()
        for a in alds:
            a.rewind()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        rel = [[evs[0][0].elapsed_time(e[s]) for s in range(1, n + 1)] for e in evs]
        return wall, rel
    window(5, lag)
    for rep in range(3):
        wall, rel = window(K, lag)
        d0 = np.diff([0.0] + rel[0]); d1 = np.diff([0.0] + rel[1])
        print('window of %d steps, lag %d: wall %.2f ms = %.3f ms/step; stream 0 step times %s ... last %s; stream 1 %s ... last %s; end of stream 0 / 1: %.2f / %.2f ms'
              % (K, lag, wall, wall / K, np.round(d0[:4], 2), np.round(d0[-3:], 2), np.round(d1[:4], 2), np.round(d1[-3:], 2), rel[0][-1], rel[1][-1]))


if __name__ == '__main__':
    main()
