#!/usr/bin/env python3
"""Throughput of the annealed-Langevin channel-estimation hot path on MI355X.

Default workload (BASELINE.json configs[1]): CDL-C-like Nt64 x Nr16 channels, pilot fraction 0.6 (38 pilots), batch of
100 channels x 17 SNR points (-10..30 dB) = 1700 lock-step trajectories per GPU, full schedule of
2311 noise levels x 3 steps = 6933 Langevin steps per trajectory (test_score.py:56,72,77; train_score.py:42).

A benchmark "step" is ONE Langevin step of the whole batch: score network forward (113 convs), data-consistency
gradient, noise, update, NMSE log.  Every one of the 6933 steps of the schedule is the same work (the noise level only
changes three scalars), so
    channels/s = trajectories / (6933 * seconds_per_step)
is the full-schedule rate; `--full-schedule` walks all 6933 steps instead of K to confirm it, and every default run also
times a sustained segment (`--sustained`, 1000 steps) after the K-step window and reports it next to the headline.

`python bench.py --gpus N` starts its own N ranks (one process per GPU, torch.distributed over RCCL) when it is not
already running under torch.distributed.run; the parent never touches the GPU.  Trajectories are independent: the weak
number gives every rank its own 1700 trajectories, and the same run also times the STRONG-scaling workload -- the
20 400 trajectories of the tune_hparams_score grid (12 (alpha, beta) cells x 17 SNR x 100 channels, BASELINE configs[2])
sharded over the ranks by shard.my_block -- reported as the `strong` object.  The only collective is the final gather of
the per-step NMSE curves.  `--workload big` runs BASELINE configs[4] instead (Nt256 x Nr64, 1024 trajectories, fp16
weights).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3       # v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2516.6     # v_mfma_f32_32x32x16_bf16 / _f16 dense ("~2.5 PF"), 16x the fp32 MFMA rate
PEAK_HBM_TBS = 8.0                 # HBM3E spec (6.3 TB/s achievable by a float4 copy)
STEPS_PER_CHANNEL = 2311 * 3
PROFILE_ROUND = 'r06'
REFERENCE_CPU_FILE = os.path.join(ROOT, 'profiles', PROFILE_ROUND + '_reference_cpu.json')   # written by tools/time_reference_cpu.py


def reference_cpu_record():
    """The reference's own test_score loop timed on the build container's cores (tools/time_reference_cpu.py: it needs
    /root/reference, which does not exist on the GPU box, so the number travels as a committed record)."""
    try:
        with open(REFERENCE_CPU_FILE) as f:
            rec = json.load(f)
        rec['source'] = 'profiles/' + os.path.basename(REFERENCE_CPU_FILE)
        return rec
    except (OSError, ValueError):
        return None


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


def cpu_baseline(n_steps=10, per_worker=8):
    """The numpy oracle (oracle/, a port of the reference loop) timed on this host's cores: W single-threaded worker
    processes, each running `n_steps` Langevin steps of its own `per_worker` channels (trajectories are independent, so
    this is how the reference's CPU path would be spread over a multi-core host).  Called BEFORE anything touches the
    GPU: the workers are spawned processes.  Reported, never used by the GPU path."""
    import multiprocessing as mp
    # one worker per PHYSICAL core (logical / 2).  Measured on the GPU box's host (256 logical cores), round 5: 64 workers 0.087
    # channels/s, 256 workers (every logical core) 0.051 -- hyper-threads and memory bandwidth make the full count slower, so
    # "all host cores" is read as all physical cores; SBC_CPU_BASELINE_WORKERS overrides
    W = max(1, (os.cpu_count() or 2) // 2)
    try:                                                                    # ... that the host's free memory can carry (~0.6 GB per worker)
        import psutil
        W = max(1, min(W, int(psutil.virtual_memory().available / 0.8e9)))
    except ImportError:
        pass
    W = int(os.environ.get('SBC_CPU_BASELINE_WORKERS', W))
    t0 = time.perf_counter()
    with mp.get_context('spawn').Pool(W) as pool:
        times = pool.map(_cpu_worker, [(w, per_worker, n_steps) for w in range(W)])
    wall = time.perf_counter() - t0
    dt = max(times) / n_steps                                               # all workers run concurrently
    n = W * per_worker
    return {'value': n / (STEPS_PER_CHANNEL * dt), 'unit': 'channels/s', 'cores': W, 'host_logical_cores': os.cpu_count(), 'kind': 'port',
            'sample': '%d Langevin steps of %d channels with the numpy oracle on %d single-threaded worker processes '
                      '(%.2f s/step for all of them, %.0f s wall including start-up), scaled to the 6933-step '
                      'schedule' % (n_steps, n, W, dt, wall)}


def self_launch(n_gpus):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process tree (this process
    has not imported torch or touched the GPU, and never will) and pass the children's exit code on."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % n_gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='cdlc', choices=['cdlc', 'big', 'train'],
                    help='cdlc: BASELINE configs[1] (Nt64xNr16, 1700 trajectories); big: configs[4] (Nt256xNr64, 1024 '
                         'trajectories, fp16 weights); train: one DSM optimiser step of train_score.py (SURVEY 8(f) F4; its '
                         'own metric, not the BASELINE one; single GPU)')
    ap.add_argument('--batch', type=int, default=32, help='--workload train: batch size (train_score.py:52)')
    ap.add_argument('--channels', type=int, default=None, help='channel realisations per GPU (test_score.py:77)')
    ap.add_argument('--snr-points', type=int, default=None)
    ap.add_argument('--graph', type=int, default=None,
                    help='1: replay each step as a hipGraph, 0: eager launches; default: what the CLIs default to '
                         '(driver.DEFAULT_USE_GRAPH)')
    ap.add_argument('--full-schedule', action='store_true', help='time all 6933 steps instead of --steps')
    ap.add_argument('--sustained', type=int, default=None,
                    help='steps of the sustained segment timed after the K-step window (default 1000; 0 disables)')
    ap.add_argument('--no-strong', action='store_true', help='skip the strong-scaling (tuner grid) measurement')
    ap.add_argument('--no-other-mode', action='store_true', help='skip timing the non-default launch mode')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-exact-mode', action='store_true', help='skip timing the same steps in conv_mode bf16x3')
    ap.add_argument('--force-dist', action='store_true',
                    help='with --gpus 1: still create a one-rank RCCL process group and route the barrier / max / gather through it')
    ap.add_argument('--conv-mode', default=None, choices=['f16x2', 'bf16x3', 'f32', 'f16w'],
                    help='convolution multiplier (config.CONV_MODES); default: config.DEFAULT_CONV_MODE, f16w for --workload big')
    ap.add_argument('--overlap', type=int, default=None,
                    help='1: independent low-resolution branches of the network on the plan side stream, 0: strictly '
                         'sequential launches; default: scorenet.DEFAULT_OVERLAP')
    ap.add_argument('--fold-stats', type=int, default=None,
                    help='1: full-resolution InstanceNorm++ statistics from tile moments (no statistics launches there), '
                         '0: a statistics launch per norm; default: scorenet.DEFAULT_FOLD_STATS')
    ap.add_argument('--fuse-res', type=int, default=None,
                    help='1: the two ResidualBlocks of the full-resolution level as one launch each (csrc/conv_res.hip); default: scorenet.DEFAULT_FUSE_RES')
    ap.add_argument('--fuse-pairs', type=int, default=None,
                    help='1: 32-channel RCU blocks as one launch each (csrc/conv_pair.hip), 0: two convolution launches; '
                         'default: scorenet.DEFAULT_FUSE_PAIRS')
    ap.add_argument('--streams', type=int, default=None,
                    help='split the trajectories into this many concurrent sub-batch streams; default: what the CLIs default '
                         'to (config.DEFAULT_STREAMS)')
    ap.add_argument('--pair-plan', type=int, default=0,
                    help='experiment: run the two sub-batch streams as ONE plan with two launch lanes, this many Langevin steps per record list '
                         '(ald.AldPair; with --graph 1: one captured graph with two branches); 0 = two host threads (driver.run_concurrently)')
    ap.add_argument('--skip-spec', default=None,
                    help='JSON list of [branch prefix, anchor record prefix, lane] (plan.hoist_skip_branches) for the small-batch plan; '
                         '"0" = no launch lanes; default: plan.DEFAULT_SKIP_SPEC for batches of at most scorenet.SKIP_OVERLAP_MAX_T trajectories')
    return ap.parse_args()


def bench_train(args):
    """--workload train: one step = perturbation + forward + DSM loss + backward + Adam + EMA on a batch of synthetic
    channels (train_score.py:145-173), everything resident on the device; prints one JSON line in the format of the
    main bench."""
    import torch
    from score_based_channels_amd import plan as P, synth
    from score_based_channels_amd.train import TrainNet
    from score_based_channels_amd.train_score import fresh_state_dict, training_config
    if args.gpus != 1:
        raise SystemExit('--workload train runs on one GPU (the reference trains on one device, train_score.py:31)')
    torch.cuda.set_device(0)
    B = args.batch
    cfg = training_config('CDL-C')
    net = TrainNet(cfg, batch=B, device='cuda:0', seed=1)
    net.load_state_dict(fresh_state_dict(cfg, 1))
    raw = synth.generate_channels('CDL-C', max(B, 16), 64, 16, 0.5, seed=1234)[:B]
    h = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1)))
    x = np.stack((h.real, h.imag), axis=1).astype(np.float32)
    gen = torch.Generator().manual_seed(0)
    labels = [torch.randint(0, cfg.model.num_classes, (B,), generator=gen) for _ in range(args.warmup + args.steps)]
    graph = bool(args.graph) if args.graph is not None else False
    with torch.cuda.stream(torch.cuda.Stream()):
        for k in range(args.warmup):
            net.step(x, labels[k], use_graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            net.step(x, labels[args.warmup + k], use_graph=graph)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        loss = float(net.loss_per_sample.mean().item())
        prof = net.profile_step(x, labels[0], repeats=3)
    flops = 3 * P.count_conv_flops(net.plan) * B                 # forward + input-gradient + weight-gradient convolutions
    names = {P.CONV: 'conv', P.CONV_WGRAD: 'conv_wgrad', P.GRAD_ADD: 'grad_add', P.INORM_BWD: 'inorm_bwd'}
    classes = {('forward ' if t < 200 else 'reverse ') + names.get(t % 100, 'kind %d' % (t % 100)): round(ms, 3)
               for t, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0])[:6]}
    out = {'metric': 'samples/s DSM training step (train_score.py), CDL-C Nt64xNr16', 'value': B / dt, 'unit': 'samples/s',
           'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32 (forward / input-gradient convolutions as exact 3-term bf16 splits, weight gradients on fp32 MFMA)',
           'data': 'synthetic (CDL-C-like cluster channels, fresh nn.Conv2d-style initial weights)',
           'config': {'workload': 'DSM optimiser step: batch %d of Nt64xNr16 channels, 2311 noise levels, Adam lr 1e-4 eps 1e-3, '
                                  'EMA 0.999 (train_score.py:34-67); %s launches' % (B, 'hipGraph' if graph else 'eager')},
           'final_loss': loss,
           'roofline': {'bound': 'the GPU-side chain of ~530 dependent small launches at this batch size (DESIGN.md section 10)',
                        'achieved': flops / dt / 1e12, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': flops / dt / 1e12 / 157.3, 'traffic': None,
                        'note': 'algorithmic conv FLOPs of the step (3 x forward) against the fp32 MFMA peak, the pipe the weight '
                                'gradients run on; forward and input-gradient convolutions run as exact 3-term bf16 splits on the '
                                'bf16 pipe (peak / 6 x 36/16 Winograd = 943.7 TFLOP/s algorithmic), so this fraction is an upper bound '
                                'on how busy either pipe is',
                        'ms_by_operator_class': classes},
           'cpu_baseline': {'value': None, 'unit': 'samples/s', 'cores': 0, 'kind': 'none',
                            'reference_build_container': {'value': 32 / 0.295, 'unit': 'samples/s', 'cores': 8,
                                                          'what': 'the reference step itself (NCSNv2Deepest + autograd + Adam + EMAHelper, '
                                                                  'PyTorch CPU): 0.26-0.33 s per step of 32 in the build container'},
                            'note': 'there is no CPU port of the backward pass to time on this host (the oracle restates loss and '
                                    'optimiser only), so `value` is null'}}
    print(json.dumps(out))


def main():
    args = parse_args()
    if args.workload == 'train':
        return bench_train(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))

    # CPU baseline first: its worker processes are spawned, which must happen before this process touches the GPU
    cpu_base = None
    if int(os.environ.get('WORLD_SIZE', '1')) == 1 and not args.no_cpu_baseline and args.workload == 'cdlc':
        cpu_base = cpu_baseline()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    n_dev = max(1, torch.cuda.device_count())
    explicit_gloo = os.environ.get('SBC_DIST_BACKEND') == 'gloo'
    if local >= n_dev and not explicit_gloo:
        # one process per GPU: more local ranks than devices is a launch error, not something to fold silently.  (A smoke run
        # that shares one GPU between ranks has to say so: SBC_DIST_BACKEND=gloo -- RCCL refuses two ranks on one device.)
        raise SystemExit('LOCAL_RANK %d but only %d visible device(s): launch one rank per GPU (or set SBC_DIST_BACKEND=gloo '
                         'to share a device in a smoke test)' % (local, n_dev))
    local %= n_dev
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_dist
    backend = None
    if use_dist:
        backend = os.environ.get('SBC_DIST_BACKEND', 'nccl')          # 'nccl' is RCCL on ROCm
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        import datetime
        kw = {'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}
        # a rank that never arrives must end the job with an error, not hang it
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get('SBC_DIST_TIMEOUT_S', '300'))), **kw)
        if dist.get_world_size() != world:
            raise SystemExit('process group has %d ranks, WORLD_SIZE says %d' % (dist.get_world_size(), world))

    from score_based_channels_amd import plan as P, shard, synth
    from score_based_channels_amd.ald import AldBatch, snr_to_noise
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd import _lib
    from score_based_channels_amd.config import DEFAULT_CONV_MODE, DEFAULT_STREAMS
    from score_based_channels_amd.driver import DEFAULT_USE_GRAPH, run_concurrently, stream_count
    from score_based_channels_amd.scorenet import ScoreNet
    from score_based_channels_amd.weights import seeded_state_dict

    big = args.workload == 'big'
    conv_mode = args.conv_mode or ('f16w' if big else DEFAULT_CONV_MODE)
    n_streams = DEFAULT_STREAMS if args.streams is None else max(1, args.streams)
    nt, nr = (256, 64) if big else (64, 16)
    npil = int(np.floor(nt * 0.6))
    nch = args.channels or (64 if big else 100)
    nsnr = args.snr_points or (16 if big else 17)
    profile = 'ULA' if big else 'CDL-C'
    cfg = default_config('CDL-C', image_size=(nr, nt)) if big else default_config('CDL-C')
    sd = seeded_state_dict(cfg, 2024)                     # random-init weights of the reference architecture
    net = ScoreNet(cfg, 'cuda:%d' % local, conv_mode=conv_mode,
                   overlap=None if args.overlap is None else bool(args.overlap),
                   fold_stats=None if args.fold_stats is None else bool(args.fold_stats),
                   fuse_pairs=None if args.fuse_pairs is None else bool(args.fuse_pairs),
                   fuse_res=None if args.fuse_res is None else bool(args.fuse_res),
                   skip_overlap=None if args.skip_spec is None else False if args.skip_spec == '0' else tuple(tuple(e) for e in json.loads(args.skip_spec))).load_state_dict(sd)
    use_graph = DEFAULT_USE_GRAPH if args.graph is None else bool(args.graph)
    snr = np.arange(-10, 32.5, 2.5)[:nsnr]

    cdev = net.device if backend != 'gloo' else torch.device('cpu')      # where collective payloads live

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_batches(H, Pm, h_idx, p_idx, ln, a0, be, traj, init, n_streams=1):
        """The trajectory list as `n_streams` AldBatch objects (independent sub-batches on their own HIP streams)."""
        alds, streams = [], []
        n_streams = stream_count(net, len(h_idx), nt, nr, n_streams)      # (small chunks: one batch with launch lanes, as the CLIs run them)
        for part in np.array_split(np.arange(len(h_idx)), n_streams):
            a = AldBatch(net, H, Pm, h_idx[part], p_idx[part], ln[part], alpha_step=a0[part], beta_noise=be[part],
                         seed=1234, traj_id=traj[part], lanes=None if n_streams == 1 else False)
            a.set_init(init[torch.from_numpy(h_idx[part])])
            a.synthesize_measurements()
            alds.append(a)
            streams.append(torch.cuda.Stream(net.device) if n_streams > 1 else torch.cuda.current_stream(net.device))
        return alds, streams

    pairs = {}

    def run_all(alds, streams, n, graph):
        if args.pair_plan and len(alds) == 2 and n > 0:
            # experiment: both sub-batches in ONE plan with two launch lanes -- one host thread, or one hipGraph with two branches
            from score_based_channels_amd.ald import AldPair
            key = id(alds[0])
            if key not in pairs:
                pairs[key] = AldPair(alds[0], alds[1], args.pair_plan)
                pairs[key].set_persistent_cus(torch.cuda.get_device_properties(net.device).multi_processor_count // 2)
            pairs[key].run(n, use_graph=graph)
            return
        run_concurrently(alds, streams, n, graph)           # one host thread per stream when there are several

    logged = [0]                 # rows of the NMSE log the most recent timed() call wrote

    last_rank_times = [None]     # (wall seconds of this rank before the closing barrier, host CPU seconds) of the last timed() call

    def timed(alds, streams, n, graph, warm):
        """`warm` untimed steps, then exactly `n` steps between barrier + synchronize pairs; max over ranks (seconds)."""
        for a in alds:
            a.rewind()
        logged[0] = warm + n
        run_all(alds, streams, warm, graph)
        sync()
        t0, c0 = time.perf_counter(), time.process_time()
        run_all(alds, streams, n, graph)
        torch.cuda.synchronize()
        last_rank_times[0] = (time.perf_counter() - t0, time.process_time() - c0)
        sync()
        return max_over_ranks(time.perf_counter() - t0)

    def per_rank(n):
        """This rank's own time for the `n` steps of the last timed() call (before the closing barrier) and the CPU time its host
        threads spent issuing them, gathered over the ranks: the spread says whether a slow step is one straggler or all of them."""
        t = torch.tensor(last_rank_times[0], dtype=torch.float64, device=cdev)
        allt = [torch.empty_like(t) for _ in range(world)] if use_dist else [t]
        if use_dist:
            dist.all_gather(allt, t)
        a = torch.stack(allt).cpu().numpy() / n * 1e3
        return {'ms_per_step_by_rank': [round(float(v), 4) for v in a[:, 0]], 'ms_per_step_min': float(a[:, 0].min()),
                'ms_per_step_max': float(a[:, 0].max()),
                'host_cpu_ms_per_step_by_rank': [round(float(v), 4) for v in a[:, 1]],
                'host_cpu_cores': os.cpu_count(),
                'note': 'wall time of each rank for the timed steps before the closing barrier; host CPU time = process_time of the '
                        'rank (its launch threads: eager launches cost host time, %d host thread(s) per rank)' % n_streams}

    # ---------------------------------------------------------------- weak workload: this rank's own trajectories
    raw = synth.generate_channels(profile, nch, nt, nr, 0.5, seed=4321 + rank)        # per-rank channel batch
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng([4321, rank]), nch, nt, npil), (0, 2, 1)))
    T = nch * len(snr)
    idx = np.tile(np.arange(nch), len(snr))
    ln = np.repeat(snr_to_noise(snr, nt), nch)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, device=net.device,
                       generator=torch.Generator(net.device).manual_seed(rank))   # one init shared by all SNR points (:115)
    alds, streams = make_batches(H, Pm, idx, idx, ln, np.full(T, 3e-11), np.full(T, 0.01), rank * T + np.arange(T), init,
                                 n_streams)

    K = STEPS_PER_CHANNEL - args.warmup if args.full_schedule else args.steps
    # headline: the launch mode and stream count the CLIs default to; nothing but the K steps inside the timed region
    dt = timed(alds, streams, K, use_graph, args.warmup)
    rank_times = per_rank(K)
    other = None
    if not args.no_other_mode and not args.full_schedule:
        other = timed(alds, streams, K, not use_graph, args.warmup)
    n_sus = (0 if args.full_schedule else 1000 if not big else 40) if args.sustained is None else args.sustained
    n_sus = min(n_sus, STEPS_PER_CHANNEL - args.warmup)
    dt_sus = timed(alds, streams, n_sus, use_graph, args.warmup) if n_sus > 0 else None

    # the one collective of the path: gather the per-step mean NMSE curves of every rank (RCCL over xGMI)
    curves = torch.cat([a.nmse_log()[:logged[0]] for a in alds], dim=1).view(-1, len(snr), nch).mean(-1)
    t_g = time.perf_counter()
    if use_dist:
        curves = curves.to(cdev)
        gathered = [torch.empty_like(curves) for _ in range(world)]
        dist.all_gather(gathered, curves)
        curves = torch.stack(gathered).mean(0)
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t_g) * 1e3
    finite = bool(torch.isfinite(curves).all().item())
    for a in alds:
        a.close()
    del alds

    # ---------------------------------------------------------------- per-kernel roofline: AFTER the timed region
    # One lock-step batch of all T trajectories on ONE stream with eager launches (a kernel then has the chip to itself, which
    # is what a per-kernel roofline means and what `rocprofv3 --kernel-trace --stats` of a one-stream run reports); each tagged
    # kernel class is bracketed by hipEvents on the launch stream (sbc_plan_profile) in its own short segment.
    klass = {}
    if rank == 0:
        p_alds, p_streams = make_batches(H, Pm, idx, idx, ln, np.full(T, 3e-11), np.full(T, 0.01), rank * T + np.arange(T), init, 1)
        pa = p_alds[0]
        run_all(p_alds, p_streams, 2, False)
        chain_tags = tuple(P.TAG_CHAIN + k for k in range(len(P.CHAIN_KERNELS)))
        for tag in (P.TAG_CONV_TOP, P.TAG_PAIR_TOP, P.TAG_POOL_TOP, P.TAG_RES_TOP, P.TAG_CONV_MID, P.TAG_DIRECT_MID) + chain_tags + (P.TAG_DOWN, P.TAG_DOWN + 1):
            ops = [op for op in net.score_plan(nt, nr, T).ops if op.tag == tag]
            if not ops:
                continue
            pa.rewind()
            pa.plan.profile(tag)
            run_all(p_alds, p_streams, 6, False)
            torch.cuda.synchronize()
            ms, n = pa.plan.profile_read()
            pa.plan.profile(-1)
            if n:
                klass[tag] = {'ms_total': ms, 'launches': n, 'launches_per_step': len(ops), 'us_per_launch': ms / n * 1e3,
                              # (CONV_DOWN: 3x3 + 1x1 at full resolution, as the reference runs them: 10 / 9 of a 3x3 convolution)
                              'flops_per_step': float(sum((2 if op.kind in (P.CONV_PAIR, P.RES_BLOCK) else P.chain_conv_count(op) if op.kind == P.CHAIN
                                                           else 10.0 / 9.0 if op.kind == P.CONV_DOWN else 1)
                                                          * 2.0 * T * op.src.h * op.src.w * 9 * op.src.c * op.dst.c for op in ops)),
                              # CHAIN records: the fraction of the 9 taps their column units execute, FLOP-weighted over the class
                              'live_taps': (sum(P.chain_live_tap_fraction(op) * P.chain_conv_count(op) for op in ops)
                                            / sum(P.chain_conv_count(op) for op in ops)) if ops[0].kind == P.CHAIN else 1.0,
                              'bytes_per_step': float(sum(4.0 * T * (2 * op.src.h * op.src.w * op.src.c + op.dst.h * op.dst.w * op.dst.c) if op.kind == P.CONV_DOWN
                                                          else 4.0 * T * op.src.h * op.src.w * (op.src.c + op.dst.c * (1 + (op.res1 is not None) + (op.res2 is not None)))
                                                          for op in ops))}
        pa.rewind()
        t0 = time.perf_counter()
        run_all(p_alds, p_streams, 10, False)
        torch.cuda.synchronize()
        one_stream_ms = (time.perf_counter() - t0) / 10 * 1e3
        for a in p_alds:
            a.close()
        del p_alds, pa
    range_flag = _lib.range_flag() if conv_mode == 'f16x2' else 0

    # ---------------------------------------------------------------- the same steps in the exact mode (bf16x3), for comparison
    exact = None
    if conv_mode == 'f16x2' and not args.no_exact_mode and not args.full_schedule and not big:
        net_f16x2 = net
        net = net_f16x2.fallback_net()                    # same checkpoint, conv_mode bf16x3 (what a flagged batch is re-run with)
        e_alds, e_streams = make_batches(H, Pm, idx, idx, ln, np.full(T, 3e-11), np.full(T, 0.01), rank * T + np.arange(T), init,
                                         n_streams)
        Ke = max(5, min(K, 20))
        dte = timed(e_alds, e_streams, Ke, use_graph, 2)
        exact = {'conv_mode': 'bf16x3', 'ms_per_step': dte / Ke * 1e3, 'value': world * T / (STEPS_PER_CHANNEL * dte / Ke),
                 'steps': Ke, 'unit': 'channels/s',
                 'what': 'the same workload with every product as an exact three-term bf16 split (six bf16 MFMAs per product block, '
                         'ELU with relative accuracy everywhere): fp32\'s range and precision unconditionally -- the mode a batch that '
                         'raises the f16x2 range flag is re-run in'}
        for a in e_alds:
            a.close()
        del e_alds
        net = net_f16x2

    # ---------------------------------------------------------------- strong workload: the tuner grid, sharded
    strong = None
    if not args.no_strong and not big and not args.full_schedule:
        cells = [(a, b) for a in (3e-11, 6e-11, 1e-10, 3e-10) for b in (0.1, 0.01, 0.001)]     # tune_hparams_score.py:20-23
        nC, nS, nB = len(cells), 17, 100
        Ttot = nC * nS * nB
        raw = synth.generate_channels('CDL-C', nB, nt, nr, 0.5, seed=4321)
        Hs = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
        Ps = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(4321), nC * nB, nt, npil), (0, 2, 1)))
        t_all = np.arange(Ttot)                                   # t = (cell * S + snr) * B + channel
        cell_of, snr_of, ch_of = t_all // (nS * nB), (t_all // nB) % nS, t_all % nB
        lo, hi = shard.my_block(Ttot, rank, world)
        sel = t_all[lo:hi]
        a0 = np.asarray([c[0] for c in cells])[cell_of[sel]]
        be = np.asarray([c[1] for c in cells])[cell_of[sel]]
        lns = snr_to_noise(np.arange(-10, 32.5, 2.5), nt)[snr_of[sel]]
        init_s = torch.randn(nB, nt, nr, dtype=torch.complex64, device=net.device,
                             generator=torch.Generator(net.device).manual_seed(99))
        s_alds, s_streams = make_batches(Hs, Ps, ch_of[sel], (cell_of * nB + ch_of)[sel], lns, a0, be, sel, init_s, n_streams)
        Ks = max(5, min(K, 20))
        dts = timed(s_alds, s_streams, Ks, use_graph, 2)
        strong = {'scaling': 'strong', 'value': Ttot / (STEPS_PER_CHANNEL * dts / Ks), 'unit': 'channels/s',
                  'ms_per_step': dts / Ks * 1e3, 'steps': Ks, 'trajectories_total': Ttot,
                  'trajectories_per_gpu': int(hi - lo),
                  'workload': 'tune_hparams_score grid (BASELINE configs[2]): 12 (alpha, beta) cells x 17 SNR points x 100 '
                              'channels = 20400 trajectories in total, contiguous blocks per rank (shard.my_block)'}
        for a in s_alds:
            a.close()
        del s_alds

    # ---------------------------------------------------------------- what ONE rank sees when test_score's own 1700 trajectories are
    # strong-sharded over 2 / 4 / 8 GPUs (shard.my_block of the flattened SNR x channel list: 850 / 425 / 213 trajectories): the
    # only strong-scaling evidence a one-GPU box can give for BASELINE configs[1] -- an upper bound on the 2/4/8-GPU rate of that
    # workload (the final gather is a few ms once per run), not a measured scaling curve
    strong_small = None
    if world == 1 and not args.no_strong and not big and not args.full_schedule:
        strong_small = {'what': 'ms per Langevin step of rank 0\'s block when the %d trajectories of the default test_score run '
                                '(BASELINE configs[1]) are sharded over W ranks (shard.my_block), timed on this ONE GPU with the CLI '
                                'defaults (blocks of at most scorenet.SKIP_OVERLAP_MAX_T trajectories: one batch, skip branches on a launch lane); '
                                'projection.channels_per_s_if_all_ranks_alike = %d / (6933 x s_per_step) is a PROJECTION, not a '
                                'multi-GPU measurement' % (T, T), 'by_world_size': {}}
        for w in (2, 4, 8):
            lo, hi = shard.my_block(T, 0, w)
            sel = np.arange(lo, hi)
            q_alds, q_streams = make_batches(H, Pm, idx[sel], idx[sel], ln[sel], np.full(len(sel), 3e-11), np.full(len(sel), 0.01),
                                             sel, init, n_streams)
            Kq = max(5, min(K, 30))
            dtq = timed(q_alds, q_streams, Kq, use_graph, 3)
            lanes = bool(q_alds[0].uses_lanes)
            for a in q_alds:
                a.close()
            del q_alds
            strong_small['by_world_size'][str(w)] = {
                'trajectories_per_gpu': int(hi - lo), 'ms_per_step': dtq / Kq * 1e3, 'steps': Kq, 'streams': len(q_streams), 'launch_lanes': lanes,
                'channels_per_s_this_gpu': (hi - lo) / (STEPS_PER_CHANNEL * dtq / Kq),
                'projection': {'channels_per_s_if_all_ranks_alike': T / (STEPS_PER_CHANNEL * dtq / Kq),
                               'efficiency_vs_one_gpu': (dt / K) / (w * dtq / Kq)}}

    # ---------------------------------------------------------------- N > 1: BASELINE configs[1] itself STRONG-sharded over the ranks of this
    # job -- the 1700 trajectories of one default test_score run, shard.my_block per rank, barrier + max over ranks like the headline: a
    # MEASUREMENT of what `torchrun ... test_score` does with its trajectories on N GPUs (the weak line above gives every rank its own 1700)
    strong_cfg2 = None
    if world > 1 and not args.no_strong and not big and not args.full_schedule:
        lo, hi = shard.my_block(T, rank, world)
        sel = np.arange(lo, hi)
        q_alds, q_streams = make_batches(H, Pm, idx[sel], idx[sel], ln[sel], np.full(len(sel), 3e-11), np.full(len(sel), 0.01), sel, init, n_streams)
        Kq = max(5, min(K, 30))
        dtq = timed(q_alds, q_streams, Kq, use_graph, 3)
        strong_cfg2 = {'scaling': 'strong', 'value': T / (STEPS_PER_CHANNEL * dtq / Kq), 'unit': 'channels/s', 'ms_per_step': dtq / Kq * 1e3, 'steps': Kq,
                       'trajectories_total': T, 'trajectories_per_gpu': int(hi - lo), 'streams': len(q_streams), 'launch_lanes': bool(q_alds[0].uses_lanes),
                       'workload': 'BASELINE configs[1]: the %d trajectories of ONE default test_score run (100 channels x 17 SNR points), contiguous blocks '
                                   'per rank (shard.my_block); time = max over ranks between barriers' % T}
        for a in q_alds:
            a.close()
        del q_alds

    if rank == 0:
        ms_per_step = dt / K * 1e3
        value = world * T / (STEPS_PER_CHANNEL * dt / K)
        flops_fwd = P.count_conv_flops(net.score_plan(nt, nr)) * T          # conv FLOPs of one step on this GPU
        dtype = {'f32': 'f32 (v_mfma_f32_32x32x2_f32)', 'bf16x3': 'bf16x3 (every fp32 operand as an exact 3-term bf16 split, 6 bf16 MFMAs per product block, fp32 accumulate: fp32-exact)',
                 'f16x2': 'f16x2 (every fp32 operand as two fp16 terms of a power-of-two-scaled value = 22-bit significand, hh + hl + lh on the fp16 '
                          'matrix cores, fp32 accumulate; fp32-class by the gate tests/test_gpu_parity.py::test_f16x2_is_fp32_class and held to every '
                          'reference golden at the north_star tolerance; unconditional fp32-exact mode: exact_mode.value)',
                 'f16w': 'f16 weights x f16-rounded activations on the fp16 matrix cores, fp32 accumulate, fp32 tensors in HBM'}
        out = {
            'metric': 'channels/s full ALD inference, %s Nt%dxNr%d' % ('CDL-C' if not big else 'ULA', nt, nr),
            'value': value, 'unit': 'channels/s',
            'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype[conv_mode],
            'data': 'synthetic (%s cluster channels, QPSK pilots, seed-derived random-init weights)'
                    % ('CDL-C-like' if not big else 'ULA plane-wave'),
            'config': {'workload': ('CDL-C Nt64xNr16, batch=100 channels x 17 SNR points (-10..30 dB) = 1700 lock-step '
                                    'trajectories per GPU, full 2311x3 schedule (BASELINE configs[1])') if not big else
                                   ('synthetic ULA Nt256xNr64, 64 channels x 16 SNR points = 1024 lock-step trajectories per '
                                    'GPU, 153 pilots, fp16 score-net weights, full 2311x3 schedule (BASELINE configs[4])'),
                       'trajectories_per_gpu': T, 'steps_per_channel': STEPS_PER_CHANNEL, 'num_pilots': npil,
                       'step_definition': 'one Langevin step (score forward + DC gradient + update + NMSE) of all '
                                          'trajectories; channels/s = trajectories / (6933 * s_per_step)',
                       'full_schedule_timed': bool(args.full_schedule), 'conv_mode': conv_mode,
                       'graph_replay': use_graph, 'launch_mode': 'hipGraph replay' if use_graph else 'eager launches',
                       'launch_mode_is_cli_default': use_graph == DEFAULT_USE_GRAPH, 'streams': n_streams,
                       'streams_is_cli_default': n_streams == DEFAULT_STREAMS, 'fuse_pairs': bool(net.fuse_pairs), 'fuse_res': bool(net.fuse_res),
                       'world_size_seen_by_backend': dist.get_world_size() if use_dist else 1, 'dist_backend': backend,
                       'parallelism': 'independent trajectories sharded over %d rank(s) on %d device(s); one all_gather of '
                                      'NMSE curves at the end (%.2f ms, backend %s)'
                                      % (world, min(world, n_dev), gather_ms,
                                         {'nccl': 'nccl = RCCL', 'gloo': 'gloo (more ranks than GPUs)', None: 'none: one rank'}[backend]),
                       'nmse_finite': finite},
            'step_conv_tflops': flops_fwd / (ms_per_step * 1e-3) / 1e12,
        }
        if dt_sus is not None:
            out['sustained_ms_per_step'] = dt_sus / n_sus * 1e3
            out['sustained_value'] = world * T / (STEPS_PER_CHANNEL * dt_sus / n_sus)
            out['sustained_steps'] = n_sus
            out['config']['sustained_channels_per_s'] = round(out['sustained_value'], 3)
        if other is not None:
            out['other_launch_mode'] = {'mode': 'eager launches' if use_graph else 'hipGraph replay',
                                        'ms_per_step': other / K * 1e3,
                                        'value': world * T / (STEPS_PER_CHANNEL * other / K)}
        if strong is not None:
            out['strong'] = strong
        if strong_small is not None:
            out['strong_small'] = strong_small
            for w, e in strong_small['by_world_size'].items():          # plain numbers in `config`: what the driver's record keeps
                out['config']['rank_share_ms_per_step_T%d' % e['trajectories_per_gpu']] = round(e['ms_per_step'], 4)
        if strong_cfg2 is not None:
            out['strong_cfg2'] = strong_cfg2
            out['config']['strong_cfg2_channels_per_s'] = round(strong_cfg2['value'], 3)
            out['config']['strong_cfg2_ms_per_step'] = round(strong_cfg2['ms_per_step'], 4)
        if exact is not None:
            out['exact_mode'] = exact
            out['config']['exact_mode_bf16x3_channels_per_s'] = round(exact['value'], 3)
        out['per_rank'] = rank_times
        out['config']['one_stream_ms_per_step'] = one_stream_ms
        if conv_mode == 'f16x2':
            out['config']['f16x2_range_flag'] = int(range_flag)          # 0: every staged activation stayed inside the fp16 range
        if klass:
            # executed matrix-core FLOPs per algorithmic FLOP of each kernel class (conv_mode decides the terms per product):
            #   Winograd F(2x2,3x3) kernels execute 16/36 of the direct products; the fused pair is a direct convolution whose
            #   first stage also computes the 2 halo rows of its 8-row tile: (10 + 8) / (2 * 8)
            terms = {'bf16x3': 6.0, 'f16x2': 3.0, 'f16w': 1.0}.get(conv_mode)
            pair_roll = nr == 16 and T * (nt // 8) >= 1024 and conv_mode in ('f16x2', 'f16w')
            # (conv_mode f16w, round 6: one matrix instruction per product -- the direct kernel takes the layers Winograd used to, csrc/conv_mfma.hip)
            names = {P.TAG_CONV_TOP: 'conv_x3_kernel<32, 32, 3, 2, 1, 4, 1, true, 1>' if conv_mode == 'f16w' else
                                     'conv_wx3_kernel<32, 32, 1, true, %s, true, 1, 1, %d>' % ({'bf16x3': '3', 'f16x2': '4'}.get(conv_mode, '3'),
                                                                                              {'bf16x3': 0, 'f16x2': 2}.get(conv_mode, 0)),
                     # (16-pixel rows, at least 1024 tiles in the launch: the three-role pipeline over row rings, csrc/conv_pair.hip)
                     P.TAG_PAIR_TOP: ('conv_pair_roll_kernel<%d>' % (2 if conv_mode == 'f16x2' else 1) if pair_roll else
                                      'conv_pair_kernel<%d, %d, %d, %d, 32>' % (nr, 4 if nr == 64 else 8, 2 if conv_mode == 'f16x2' else 1, 8 if nr == 64 else 4)),
                     P.TAG_POOL_TOP: 'conv_pool_kernel<%d, 8, %d, 4, 32>' % (nr, 2 if conv_mode == 'f16x2' else 1),
                     P.TAG_RES_TOP: 'conv_res_kernel',
                     # (conv_mode f16x2 with 8-pixel rows, round 6: the direct persistent kernel also takes the layers with a norm prologue and a
                     # tile-moment output -- its NM instantiation, csrc/conv_dp.hip)
                     P.TAG_CONV_MID: 'conv_x3_kernel<64, 64, 3, 2, 2, 4, 1, true, 1> (conv_wx3_kernel<64, 64, 1, true, 2, false, 2, 1, 1> for the producers of tile moments)' if conv_mode == 'f16w' else
                                     'conv_dp_kernel<64, 8, 8, 1, false, 4, true>' if conv_mode == 'f16x2' and nr == 16 and not os.environ.get('SBC_NO_CONV_DP_NORM') else
                                     'conv_wx3_kernel<64, 64, 1, true, 2, false, 2, 1, %d>' % {'bf16x3': 0, 'f16x2': 2}.get(conv_mode, 0),
                     # (conv_mode f16x2 with 8-pixel rows: the direct persistent kernel, csrc/conv_dp.hip; else the Winograd kernel)
                     P.TAG_DIRECT_MID: ('conv_dp_kernel<64, 8, 8, 1, false, 4, false>' if conv_mode == 'f16x2' and nr == 16 else
                                        'conv_x3_kernel<64, 64, 3, 2, 2, 4, 1, true, 1>' if conv_mode == 'f16w' else
                                        'conv_wx3_kernel<64, 64, 1, true, 2, false, 2, 1, %d>' % {'bf16x3': 0, 'f16x2': 2}.get(conv_mode, 0))}
            direct_mid = names[P.TAG_DIRECT_MID].startswith('conv_dp')
            direct_norm_mid = names[P.TAG_CONV_MID].startswith('conv_dp')
            names[P.TAG_DOWN], names[P.TAG_DOWN + 1] = 'conv_down_kernel<32, 64, 16>', 'conv_down_kernel<64, 64, 8>'
            for k, (cc, cw) in enumerate(P.CHAIN_KERNELS):
                # (fourth parameter: the group divisor launch_chain picks by batch size -- full groups at this size unless the batch is small)
                gd = 1
                if cw == 2:
                    gd = (4 if (T + 1) // 2 <= 128 else 2 if 2 * ((T + 7) // 8) <= 256 else 1) if cc == 128 else (2 if 2 * ((T + 3) // 4) <= 512 else 1)
                names[P.TAG_CHAIN + k] = 'conv_chain_kernel<%d, %d, %d, %d>' % (cc, cw, 8 if (cc, cw) in ((128, 2), (64, 8)) else 4, gd)
            what = {P.TAG_CONV_TOP: 'the unfused 3x3 32->32 convolutions at %dx%d (%s)' % (nt, nr, 'direct, one matrix instruction per product' if conv_mode == 'f16w' else 'Winograd F(2x2,3x3)'),
                    P.TAG_PAIR_TOP: 'the fused RCU blocks at %dx%d: two direct 3x3 32->32 convolutions per launch, intermediate in LDS%s'
                                    % (nt, nr, '; a workgroup walks a contiguous run of 8-row tiles and keeps the rows adjacent tiles share in LDS rings '
                                               '(no halo recomputation: round 5)' if pair_roll else ''),
                    P.TAG_POOL_TOP: 'the fused CRP stages at %dx%d: 5x5 max pool + direct 3x3 32->32 convolution + running sum per launch, pooled tensor in LDS' % (nt, nr),
                    P.TAG_RES_TOP: 'the fused ResidualBlocks at %dx%d: norm, ELU, direct 3x3 32->32 convolution, InstanceNorm++ statistics of the whole '
                                   'intermediate sample, norm, ELU, second convolution, + x per launch; one workgroup per sample' % (nt, nr),
                    P.TAG_CONV_MID: 'the undilated 3x3 64->64 convolutions of the %dx%d level with a norm prologue, a resized operand or '
                                    'a tile-moment output (%s)' % (nt // 2, nr // 2, 'direct, filter fragments in registers' if direct_norm_mid else 'Winograd F(2x2,3x3)'),
                    P.TAG_DIRECT_MID: 'the other undilated 3x3 64->64 convolutions of the %dx%d level (%s)'
                                      % (nt // 2, nr // 2, 'direct, filter fragments in registers' if direct_mid else 'Winograd F(2x2,3x3)')}
            for k, (cc, cw) in enumerate(P.CHAIN_KERNELS):
                what[P.TAG_CHAIN + k] = ('the chains of RCU / CRP blocks and ResidualBlocks at %dx%d with %d channels: one launch per run of blocks, %d samples '
                                         'per workgroup, the running tensor in registers, operands in LDS, direct 3x3 convolutions whose column units '
                                         'skip the taps that only read padding (csrc/conv_chain.hip)'
                                         % (nt * cw // nr, cw, cc, {2: 8 if cc == 128 else 4, 4: 2, 8: 1}[cw]))
            for k in (0, 1):
                what[P.TAG_DOWN + k] = ('the pooled 3x3 convolution + pooled 1x1 shortcut of the downsampling ResidualBlock at %dx%d: 4x4 and 2x2 stride-2 direct '
                                        'convolutions with the pooled filters, both inputs read once (csrc/conv_down.hip)' % (nt >> k, nr >> k))
            entries = {}
            for tag, kc in klass.items():
                t_launch = kc['us_per_launch'] * 1e-6
                fl = kc['flops_per_step'] / kc['launches_per_step']           # algorithmic FLOPs of an average launch of the class
                by = kc['bytes_per_step'] / kc['launches_per_step']
                ratio = (((nt + 1.0) / nt if pair_roll else 18.0 / 16.0) if tag == P.TAG_PAIR_TOP else
                         1.0 if tag in (P.TAG_POOL_TOP, P.TAG_RES_TOP) or (tag == P.TAG_DIRECT_MID and direct_mid) or (tag == P.TAG_CONV_MID and direct_norm_mid)
                         else 0.5 if tag >= P.TAG_DOWN else kc['live_taps'] if tag >= P.TAG_CHAIN else 1.0 if conv_mode == 'f16w' else 16.0 / 36.0) * (terms or 1.0)
                ach = fl / t_launch / 1e12
                e = {'kernel': names[tag], 'what': what[tag], 'launches_per_step': kc['launches_per_step'],
                     'us_per_launch': kc['us_per_launch'], 'share_of_one_stream_step': kc['us_per_launch'] * kc['launches_per_step'] / 1e3 / one_stream_ms,
                     'algorithmic_gflop_per_launch': fl / 1e9, 'algorithmic_tflops': ach,
                     'algorithmic_hbm_bytes_per_launch': by, 'algorithmic_TBps': by / t_launch / 1e12}
                if conv_mode == 'f32':
                    e.update(executed_mfma_tflops=ach * 16.0 / 36.0, mfma_peak_tflops=PEAK_F32_MFMA_TFLOPS,
                             mfma_busy=ach * 16.0 / 36.0 / PEAK_F32_MFMA_TFLOPS)
                else:
                    e.update(executed_mfma_tflops=ach * ratio, mfma_peak_tflops=PEAK_BF16_MFMA_TFLOPS,
                             mfma_busy=ach * ratio / PEAK_BF16_MFMA_TFLOPS)
                entries[tag] = e
            # Top-level entry: the DOMINANT KERNEL -- the tagged class with the largest share of a one-stream step (round 5; rounds
            # 3-4 aggregated the four full-resolution 32 -> 32 classes here and called the busy fraction `frac`).  SURVEY section 8(d):
            # achieved = ALGORITHMIC (direct-convolution) FLOPs of one launch / its average duration; peak = the dense fp16 MFMA
            # peak; frac = achieved / peak.  How busy the matrix pipe is (executed instructions) is `mfma_busy`, next to it.
            dom = max(entries, key=lambda t: entries[t]['share_of_one_stream_step'])
            cls = [dom]
            t_cls = entries[dom]['us_per_launch'] * 1e-6
            fl_cls = klass[dom]['flops_per_step'] / klass[dom]['launches_per_step']
            by_cls = klass[dom]['bytes_per_step'] / klass[dom]['launches_per_step']
            traffic, tsrc = None, None
            for rnd in (PROFILE_ROUND,):       # (a byte count of another round's kernels is not this build's: no fallback)
                tfile = os.path.join(ROOT, 'profiles', '%s_traffic_%s.json' % (rnd, args.workload))
                if not os.path.exists(tfile):
                    continue
                with open(tfile) as f:
                    tj = json.load(f)
                if (tj.get('trajectories_per_launch') == T and tj.get('conv_mode') == conv_mode
                        and entries[dom]['kernel'] in tj.get('kernels', {})):
                    traffic = tj['kernels'][entries[dom]['kernel']]['hbm_bytes_per_launch']
                    tsrc = 'profiles/' + os.path.basename(tfile)
                    break
            hbm_bound = conv_mode == 'f16w'
            mpeak = PEAK_F32_MFMA_TFLOPS if conv_mode == 'f32' else PEAK_BF16_MFMA_TFLOPS
            note = ('dominant kernel: %s -- %s; %d launches per step, avg %.1f us, %.0f %% of a one-stream step.  Timed by hipEvents on the '
                    'launch stream in a one-stream eager segment AFTER the timed region; profiles/%s_kernel_stats_%s_steps10.csv is '
                    'rocprofv3 --kernel-trace --stats of the same one-stream command.  '
                    % (entries[dom]['kernel'], what[dom], entries[dom]['launches_per_step'], entries[dom]['us_per_launch'],
                       100 * entries[dom]['share_of_one_stream_step'], PROFILE_ROUND, args.workload))
            if hbm_bound:
                rf = {'bound': 'hbm', 'achieved': by_cls / t_cls / 1e9, 'peak': PEAK_HBM_TBS * 1e3, 'unit': 'GB/s',
                      'frac': by_cls / t_cls / 1e12 / PEAK_HBM_TBS, 'traffic': traffic,
                      'kernel': note + 'achieved = algorithmic bytes of one launch (fp32 input + output + residual operands; a fused pair '
                                       'moves its input and its output only) / its average duration, against the 8 TB/s HBM3E peak '
                                       '(6.3 TB/s achievable); traffic = PMC bytes per launch'}
            else:
                rf = {'bound': 'mfma', 'achieved': fl_cls / t_cls / 1e12, 'peak': mpeak, 'unit': 'TFLOP/s',
                      'frac': fl_cls / t_cls / 1e12 / mpeak, 'traffic': traffic,
                      'mfma_busy': entries[dom]['mfma_busy'], 'executed_mfma_tflops': entries[dom]['executed_mfma_tflops'],
                      'kernel': note + 'achieved = algorithmic (direct-convolution, 2 x MACs) FLOPs of one launch / its average duration; '
                                       'peak = dense f16 MFMA peak; frac = achieved / peak (SURVEY section 8(d)).  mfma_busy = EXECUTED '
                                       'matrix-instruction FLOPs / time / peak: in f16x2 every product is 3 fp16 MFMAs (hh + hl + lh), a direct '
                                       'fused pair evaluates its first convolution on two extra rows per sample (x 130/128 at 64 rows; the '
                                       'tile-at-a-time kernels of small batches: two extra rows per 8-row tile, x 18/16), Winograd F(2x2,3x3) '
                                       'executes 16/36 of the products'}
            # the whole Langevin step the headline times, by the same definition (= channels/s x 5.6904e12 / (n_gpus x peak))
            rf['frac_step'] = flops_fwd / (ms_per_step * 1e-3) / 1e12 / mpeak
            rf['frac_note'] = ('frac, frac_step and kernels[*].frac_algorithmic: direct-convolution FLOPs / time / %.1f TFLOP/s; mfma_busy: '
                               'executed matrix-instruction FLOPs / time / the same peak (agrees with SQ_VALU_MFMA_BUSY_CYCLES)' % mpeak)
            # (outside conv_mode f16x2 both half-resolution classes run the same Winograd kernel: keep their entries apart)
            key = {t: names[t] if list(names[u] for u in entries).count(names[t]) == 1 else '%s [class %d]' % (names[t], t) for t in entries}
            rf['kernels'] = {key[t]: {k: v for k, v in entries[t].items() if k != 'kernel'} for t in entries}
            for t in entries:
                rf['kernels'][key[t]]['frac_algorithmic'] = entries[t]['algorithmic_tflops'] / mpeak
            if traffic is not None:
                rf['traffic_source'] = tsrc + ' (rocprofv3 --pmc passes of the one-stream command: FETCH_SIZE x 2 + WRITE_SIZE per launch)'
            out['roofline'] = rf
        if cpu_base is not None:
            ref_rec = reference_cpu_record()
            if ref_rec is not None:
                cpu_base['reference_build_container'] = ref_rec
            out['cpu_baseline'] = cpu_base
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
