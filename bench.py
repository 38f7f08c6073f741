#!/usr/bin/env python3
"""Throughput of the annealed-Langevin channel-estimation hot path on MI355X.

Workload (BASELINE.json configs[1]): CDL-C-like Nt64 x Nr16 channels, pilot fraction 0.6 (38 pilots), batch of
100 channels x 17 SNR points (-10..30 dB) = 1700 lock-step trajectories per GPU, full schedule of
2311 noise levels x 3 steps = 6933 Langevin steps per trajectory (test_score.py:56,72,77; train_score.py:42).

A benchmark "step" is ONE Langevin step of the whole 1700-trajectory batch: score network forward (113 convs),
data-consistency gradient, noise, update, NMSE log.  Every one of the 6933 steps of the schedule is the same
work (the noise level only changes three scalars), so
    channels/s = trajectories / (6933 * seconds_per_step)
is the full-schedule rate; `--full-schedule` walks all 6933 steps instead of K to confirm it.

One process per GPU (torch.distributed / RCCL when WORLD_SIZE > 1): trajectories are independent, each rank
runs its own 1700 (weak scaling), the only collective is the final gather of the per-step NMSE curves.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6     # same guide: v_mfma_f32_32x32x16_bf16 dense, 16x the fp32 MFMA rate ("~2.5 PF")
STEPS_PER_CHANNEL = 2311 * 3


def _cpu_worker(job):
    """One worker process of the CPU baseline: `n_steps` Langevin steps of its own `n` channels with the numpy oracle on
    ONE BLAS thread (the oracle's GEMMs are small: more OpenBLAS threads per process make it slower on this host)."""
    wid, n, n_steps = job
    from threadpoolctl import threadpool_limits
    from oracle import ald_oracle as A, ncsnv2_oracle as O
    from score_based_channels_amd import synth
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.noise import HostNoise
    from score_based_channels_amd.weights import seeded_state_dict
    cfg = default_config('CDL-C')
    sd = seeded_state_dict(cfg, 2024)
    nt, nr, npil = 64, 16, 38
    raw = synth.generate_channels('CDL-C', n, nt, nr, 0.5, 1 + wid)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    P = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng([2, wid]), n, nt, npil), (0, 2, 1)))
    noise = HostNoise(3 + wid)
    Y = A.make_measurements(P, H, 64.0, noise.measurement(0, (n, npil, nr)))
    score = lambda x, lab: O.score_forward(sd, x, lab)                      # noqa: E731
    with threadpool_limits(limits=1):
        A.ald_run(score, sd['sigmas'], cfg.model.sigma_end, P, Y, H, noise.init(H.shape), noise.step_stream(0, H.shape),
                  64.0, levels=[0], steps_each=1)                            # warm-up
        t0 = time.perf_counter()
        A.ald_run(score, sd['sigmas'], cfg.model.sigma_end, P, Y, H, noise.init(H.shape),
                  noise.step_stream(0, H.shape), 64.0, levels=[0], steps_each=n_steps)
        return time.perf_counter() - t0


def cpu_baseline(n_steps=20, per_worker=8):
    """The numpy oracle (oracle/, a port of the reference loop) timed on this host's cores: W single-threaded worker
    processes, each running `n_steps` Langevin steps of its own `per_worker` channels (trajectories are independent, so
    this is how the reference's CPU path would be spread over a multi-core host).  Called BEFORE anything touches the
    GPU: the workers are spawned processes.  Reported, never used by the GPU path."""
    import multiprocessing as mp
    W = max(1, min(64, (os.cpu_count() or 2) // 2))
    t0 = time.perf_counter()
    with mp.get_context('spawn').Pool(W) as pool:
        times = pool.map(_cpu_worker, [(w, per_worker, n_steps) for w in range(W)])
    wall = time.perf_counter() - t0
    dt = max(times) / n_steps                                               # all workers run concurrently
    n = W * per_worker
    return {'value': n / (STEPS_PER_CHANNEL * dt), 'unit': 'channels/s', 'cores': W, 'kind': 'port',
            'sample': '%d Langevin steps of %d channels with the numpy oracle on %d single-threaded worker processes '
                      '(%.2f s/step for all of them, %.0f s wall including start-up), scaled to the 6933-step '
                      'schedule' % (n_steps, n, W, dt, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--channels', type=int, default=100, help='channel realisations per GPU (test_score.py:77)')
    ap.add_argument('--snr-points', type=int, default=17)
    ap.add_argument('--graph', type=int, default=0, help='replay the step as a hipGraph (no per-kernel timing)')
    ap.add_argument('--full-schedule', action='store_true', help='time all 6933 steps instead of --steps')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--conv-mode', default='bf16x3', choices=['bf16x3', 'f32'],
                    help='convolution multiplier (scorenet.CONV_MODES)')
    ap.add_argument('--streams', type=int, default=1,
                    help='split the trajectories into this many concurrent sub-batch streams (DESIGN.md section 7)')
    args = ap.parse_args()

    # CPU baseline first: its worker processes are spawned, which must happen before this process touches the GPU
    cpu_base = None
    if int(os.environ.get('WORLD_SIZE', '1')) == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    local %= max(1, torch.cuda.device_count())           # identity on a full node; lets 2 ranks share 1 GPU in a smoke test
    torch.cuda.set_device(local)
    if world > 1:
        backend = os.environ.get('SBC_DIST_BACKEND', 'nccl')   # 'nccl' is RCCL on ROCm
        kw = {'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}
        dist.init_process_group(backend, **kw)

    from score_based_channels_amd import plan as P, synth
    from score_based_channels_amd.ald import AldBatch, snr_to_noise
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.scorenet import ScoreNet
    from score_based_channels_amd.weights import seeded_state_dict

    cfg = default_config('CDL-C')
    sd = seeded_state_dict(cfg, 2024)                     # random-init weights of the reference architecture
    net = ScoreNet(cfg, 'cuda:%d' % local, conv_mode=args.conv_mode).load_state_dict(sd)
    nt, nr, npil = 64, 16, int(np.floor(64 * 0.6))
    nch, nsnr = args.channels, args.snr_points
    raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=4321 + rank)       # per-rank channel batch
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    pil = synth.qpsk_pilots(np.random.default_rng([4321, rank]), nch, nt, npil)
    Pm = np.conj(np.transpose(pil, (0, 2, 1)))
    snr = np.arange(-10, 32.5, 2.5)[:nsnr]
    T = nch * len(snr)
    idx = np.tile(np.arange(nch), len(snr))
    ln = np.repeat(snr_to_noise(snr, nt), nch)
    traj = rank * T + np.arange(T)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, device=net.device,
                       generator=torch.Generator(net.device).manual_seed(rank))
    init = init.repeat(len(snr), 1, 1)                   # one initial estimate shared by all SNR points (:115)
    # the trajectory list may be split over several HIP streams: independent sub-batches whose kernels the GPU
    # interleaves (fills the tail of one launch with the head of another, overlaps memory-bound with MFMA-bound ones)
    parts = np.array_split(np.arange(T), args.streams)
    alds, streams = [], []
    for part in parts:
        a = AldBatch(net, H, Pm, idx[part], idx[part], ln[part], alpha_step=3e-11, beta_noise=0.01, seed=1234,
                     traj_id=traj[part])
        a.set_init(init[torch.from_numpy(part)])
        a.synthesize_measurements()
        alds.append(a)
        streams.append(torch.cuda.Stream(net.device) if args.streams > 1 else torch.cuda.current_stream(net.device))
    ald = alds[0]

    def run_all(n, graph):
        for a, st in zip(alds, streams):
            with torch.cuda.stream(st):
                a.run(n, use_graph=graph)

    K = STEPS_PER_CHANNEL - args.warmup if args.full_schedule else args.steps
    use_graph = bool(args.graph)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    run_all(args.warmup, use_graph)
    if not use_graph:
        ald.plan.profile(P.TAG_CONV_TOP)
    sync()
    t0 = time.perf_counter()
    run_all(K, use_graph)
    sync()
    dt = time.perf_counter() - t0
    kern_ms, kern_n = ald.plan.profile_read() if not use_graph else (0.0, 0)
    ald.plan.profile(-1)
    tmax = torch.tensor([dt], dtype=torch.float64, device=net.device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # the one collective of the path: gather the per-step mean NMSE curves of every rank (RCCL over xGMI)
    curves = torch.cat([a.nmse_log()[:args.warmup + K] for a in alds], dim=1).view(-1, len(snr), nch).mean(-1)
    t_g = time.perf_counter()
    if world > 1:
        gathered = [torch.empty_like(curves) for _ in range(world)]
        dist.all_gather(gathered, curves)
        curves = torch.stack(gathered).mean(0)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t_g) * 1e3
    finite = bool(torch.isfinite(curves).all().item())

    if rank == 0:
        ms_per_step = dt / K * 1e3
        value = world * T / (STEPS_PER_CHANNEL * dt / K)
        flops_fwd = P.count_conv_flops(net.score_plan(nt, nr)) * T          # conv FLOPs of one step on this GPU
        out = {
            'metric': 'channels/s full ALD inference, CDL-C Nt64xNr16', 'value': value, 'unit': 'channels/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.conv_mode == 'f32' else 'f32 (products as exact 3-term bf16 splits on the bf16 matrix cores, fp32 accumulate)',
            'data': 'synthetic (CDL-C-like cluster channels, QPSK pilots, seed-derived random-init weights)',
            'config': {'workload': 'CDL-C Nt64xNr16, batch=100 channels x 17 SNR points (-10..30 dB) = 1700 '
                                   'lock-step trajectories per GPU, full 2311x3 schedule',
                       'trajectories_per_gpu': T, 'steps_per_channel': STEPS_PER_CHANNEL, 'num_pilots': npil,
                       'step_definition': 'one Langevin step (score forward + DC gradient + update + NMSE) of all '
                                          'trajectories; channels/s = trajectories / (6933 * s_per_step)',
                       'full_schedule_timed': bool(args.full_schedule), 'conv_mode': args.conv_mode, 'graph_replay': use_graph, 'streams': args.streams,
                       'parallelism': 'independent trajectories sharded over %d GPU(s); one RCCL all_gather of NMSE '
                                      'curves at the end (%.2f ms)' % (world, gather_ms),
                       'nmse_finite': finite},
            'step_conv_tflops': flops_fwd / (ms_per_step * 1e-3) / 1e12,
        }
        if kern_n:
            per_launch = 2.0 * alds[0].T * nt * nr * 9 * 32 * 32            # 3x3 conv 32->32 at 64x16, 2*MACs
            ach = per_launch / (kern_ms / kern_n * 1e-3) / 1e12
            traffic = None                  # HBM bytes per launch from the PMC passes (profiles/), same workload only
            tfile = os.path.join(ROOT, 'profiles', 'r01_traffic.json')
            if os.path.exists(tfile) and args.conv_mode == 'bf16x3':
                with open(tfile) as f:
                    tj = json.load(f)
                if tj.get('trajectories_per_launch') == alds[0].T:
                    traffic = tj.get('hbm_bytes_per_launch')
            if args.conv_mode == 'bf16x3':
                # fp32-exact products on the bf16 matrix cores: 6 bf16 MFMAs per fp32 product block, so the roofline of
                # the instruction the kernel issues is the dense bf16 MFMA peak / 6
                peak = PEAK_BF16_MFMA_TFLOPS / 6.0
                kname = ('conv_wx3_kernel<32, 32, 1, true, 3, true, 1, 1>: the 18 3x3 32->32 convolutions at 64x16 of every step (%d '
                         'tagged launches, avg %.1f us).  achieved = direct-convolution FLOPs (2*9*32*32 per pixel) / '
                         'time; peak = dense bf16 MFMA peak %.1f / 6 (fp32 operands as three exact bf16 terms, six bf16 '
                         'MFMAs per product block); the kernel executes 16/36 of the products (Winograd F(2x2,3x3)), '
                         'i.e. %.0f TFLOP/s of bf16 MFMA' % (kern_n, kern_ms / kern_n * 1e3, PEAK_BF16_MFMA_TFLOPS,
                                                            ach * 6 * 16 / 36))
            else:
                peak = PEAK_F32_MFMA_TFLOPS
                kname = ('conv_wino_kernel<32, 32, 2, true>: the 18 3x3 32->32 convolutions at 64x16 of every step (%d '
                         'launches, avg %.1f us).  achieved = direct-convolution FLOPs (2*9*32*32 per pixel) / time; the '
                         'kernel executes 16/36 of them (fp32 Winograd F(2x2,3x3)), i.e. %.1f TFLOP/s on the MFMA pipe'
                         % (kern_n, kern_ms / kern_n * 1e3, ach * 16 / 36))
            out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                               'frac': ach / peak, 'traffic': traffic, 'kernel': kname}
        if cpu_base is not None:
            out['cpu_baseline'] = cpu_base
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
