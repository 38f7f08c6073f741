#!/usr/bin/env python3
"""Score-based channel estimation over an SNR sweep -- counterpart of
``src/score_based_channels/test_score.py`` with the sampling loop on the MI355X HIP path.

    python -m score_based_channels_amd.test_score --train CDL-C --test CDL-C --spacing 0.5 --pilot_alpha 0.6

Same arguments, input files (``./models/score/<train>/final_model.pt``, ``./data/*.mat``) and outputs
(``./results/score/train-<train>_test-<test>/results.{pt,png}``, same dictionary keys and array shapes) as the
reference (test_score.py:15-22,33-36,83,187-200).  Additions, all default-off: ``--seed`` (the reference never
seeds its RNG), ``--levels_stride/--num_levels`` (truncated schedules), ``--num_channels``, ``--synthetic`` /
``--synthetic_weights`` (stand-ins for the undistributed blobs), multi-GPU via ``torch.distributed.run``.
All 17 SNR points run as one lock-step batch (they are independent: each restarts from the same initial
estimate, test_score.py:126).

``--test`` takes one profile like the reference, or SEVERAL (``--test CDL-A CDL-B CDL-C CDL-D``, BASELINE configs[3]): the
(profile x SNR x channel) trajectories of a (spacing, pilot_alpha) combination then form ONE list that is sharded over the ranks
(``shard.my_block``), and every profile gets the result directory and ``results.pt`` its own single-profile invocation with the
same ``--seed`` writes -- bit for bit (same pilots, noise keys and normalisation constants, which always come from the TRAIN
profile's dataset, test_score.py:68-69,101).
"""
import argparse
import copy
import itertools
import os

import numpy as np

from .config import CONV_MODES, DEFAULT_CONV_MODE, DEFAULT_STREAMS


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpu', type=int, default=0)
    p.add_argument('--train', type=str, default='CDL-C')
    p.add_argument('--test', type=str, nargs='+', default=['CDL-C'],
                   help='test profile (reference: one); [added] several profiles run as ONE sharded trajectory list')
    p.add_argument('--save_channels', type=int, default=0)
    p.add_argument('--spacing', nargs='+', type=float, default=[0.5])
    p.add_argument('--pilot_alpha', nargs='+', type=float, default=[0.6])
    # additions of this build
    p.add_argument('--seed', type=int, default=None, help='seed of every noise stream (default: fresh entropy)')
    p.add_argument('--levels_stride', type=int, default=1, help='walk every k-th noise level (+ the last)')
    p.add_argument('--num_levels', type=int, default=None, help='walk only the first n selected levels')
    p.add_argument('--num_channels', type=int, default=100)
    p.add_argument('--synthetic', action='store_true', help='generate CDL-like channels instead of reading ./data')
    p.add_argument('--synthetic_weights', type=int, default=None, metavar='SEED',
                   help='seed-derived random weights instead of ./models/score/<train>/final_model.pt')
    p.add_argument('--noise', type=str, default='device', choices=['device', 'host'],
                   help="Gaussian draws: in-kernel Philox streams keyed by (seed, trajectory, step) [device, default], or "
                        "the keyed host streams of noise.HostNoise replayed from memory [host] -- the streams the "
                        "reference goldens were generated with (parity runs; needs n_steps x T x Nt x Nr x 8 bytes)")
    p.add_argument('--no_plot', action='store_true')
    p.add_argument('--conv_mode', type=str, default=DEFAULT_CONV_MODE, choices=list(CONV_MODES),
                   help='convolution multiplier (scorenet.ScoreNet): f16x2 [default] and bf16x3 are fp32-class on the fp16 / bf16 '
                        'matrix cores, f32 is fp32 MFMA, f16w rounds the parameters to fp16 (BASELINE config 5; looser tolerance)')
    p.add_argument('--no_graph', action='store_true', help='launch kernels eagerly instead of hipGraph replay')
    p.add_argument('--graph', action='store_true', help='[added] replay each Langevin step as a hipGraph (default: driver.DEFAULT_USE_GRAPH)')
    p.add_argument('--max_batch', type=int, default=4096,
                   help='[added] largest lock-step batch per device (trajectories beyond it run as further chunks; results do not '
                        'depend on it)')
    p.add_argument('--streams', type=int, default=DEFAULT_STREAMS,
                   help='[added] run each lock-step batch as this many concurrent sub-batches on their own HIP streams '
                        '(bit-identical results; +7 %% at 2 on MI355X for 1700 trajectories)')
    p.add_argument('--force_dist', action='store_true',
                   help='[added] with ONE rank: still create the torch.distributed group (backend nccl = RCCL on a GPU box) and route the seed '
                        'broadcast, the agreement points and the final gathers through it')
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    import torch
    from . import shard
    rank, world, local = shard.init_distributed(force=getattr(args, 'force_dist', False))
    # (a rank that fails tells the others at their next agreement point instead of leaving them in a collective: shard.run_guarded)
    return shard.run_guarded(world, lambda: _main(args, rank, world, local))


def _main(args, rank, world, local):
    import torch
    from . import shard
    from .checkpoint import load_checkpoint
    from .config import default_config
    from .driver import host_noise_streams, level_subset, resolve_launch_mode, run_trajectories, shared_init
    from .loaders import Channels
    from .scorenet import ScoreNet
    from .weights import seeded_state_dict

    device = 'cuda:%d' % (local if world > 1 else args.gpu)
    torch.cuda.set_device(device)
    tests = list(args.test) if isinstance(args.test, (list, tuple)) else [args.test]

    if args.synthetic_weights is not None:
        config = default_config(args.train)
        model_state = seeded_state_dict(config, args.synthetic_weights)
    else:
        contents = load_checkpoint(os.path.join('./models/score/%s' % args.train, 'final_model.pt'))
        config, model_state = contents['config'], contents['model_state']
    alpha_step, beta_noise = 3e-11, 0.01                  # all profiles, test_score.py:39-54
    config.sampling.steps_each = 3                        # :56
    diffuser = ScoreNet(config, device, conv_mode=args.conv_mode).load_state_dict(model_state).eval()

    seed = int.from_bytes(os.urandom(4), 'little') if args.seed is None else args.seed
    if world > 1 or args.force_dist:                      # every rank must use rank 0's seed
        seed = shard.broadcast_int(seed, 0, device)
    np.random.seed(seed % (2 ** 32))                      # pilots come from numpy's legacy global RNG (loaders.py:52-55)

    train_seed, val_seed = 1234, 4321
    config.data.channel = args.train
    dataset = Channels(train_seed, config, norm=config.data.norm_channels, synthetic=args.synthetic)
    # every test profile sees numpy's legacy RNG in the state its own invocation would: just behind the train dataset's pilots
    rng_after_train = np.random.get_state()

    snr_range = np.arange(-10, 32.5, 2.5)
    spacing_range = np.asarray(args.spacing)
    pilot_alpha_range = np.asarray(args.pilot_alpha)
    nt = config.data.image_size[1]
    noise_range = 10 ** (-snr_range / 10.) * nt
    num_channels = args.num_channels
    levels = level_subset(config.model.num_classes, args.levels_stride, args.num_levels)
    n_steps = len(levels) * config.sampling.steps_each
    nP, S = len(tests), len(snr_range)
    nmse_log = np.zeros((nP, len(spacing_range), len(pilot_alpha_range), S, n_steps, num_channels))
    saved_H = None                                        # --save_channels: final estimates per (profile, spacing, alpha, SNR, channel)
    result_dirs = ['./results/score/train-%s_test-%s' % (args.train, t) for t in tests]
    if rank == 0:
        for d in result_dirs:
            os.makedirs(d, exist_ok=True)

    val_configs = [None] * nP
    rng_by_profile = [rng_after_train] * nP
    run_info = {}                                         # 'f16x2_fallback': chunks (of any rank) re-run in bf16x3 (driver.py)
    for meta_idx, (spacing, pilot_alpha) in enumerate(itertools.product(spacing_range, pilot_alpha_range)):
        spacing_idx, pilot_alpha_idx = np.unravel_index(meta_idx, (len(spacing_range), len(pilot_alpha_range)))
        Hs, Ps = [], []
        for pi, test in enumerate(tests):
            val_config = copy.deepcopy(config)
            val_config.data.channel = test
            val_config.data.spacing_list = [spacing]
            val_config.data.num_pilots = int(np.floor(nt * pilot_alpha))
            np.random.set_state(rng_by_profile[pi])                         # (one profile: a no-op, the state is the current one)
            val_dataset = Channels(val_seed, val_config, norm=[dataset.mean, dataset.std], synthetic=args.synthetic)
            if rank == 0:
                print('There are %d validation channels' % len(val_dataset))
            sample = val_dataset.batch(num_channels)                        # first batch, shuffle=False (:102-108)
            Ps.append(np.conj(np.transpose(sample['P'], (0, 2, 1))))        # Hermitian pilots [B, Np, Nt] (:109-111)
            Hs.append(sample['H_herm'][:, 0] + 1j * sample['H_herm'][:, 1])  # [B, Nt, Nr] (:112-113)
            val_configs[pi] = val_config
            rng_by_profile[pi] = np.random.get_state()                      # where THIS profile's own invocation would continue
        val_P, val_H = np.concatenate(Ps, axis=0), np.concatenate(Hs, axis=0)
        nr = val_H.shape[-1]
        B = num_channels
        step_noise = meas_noise = None
        if args.noise == 'host':
            init, meas_noise, step_noise = host_noise_streams(seed, meta_idx, Hs[0].shape, S, n_steps, (B, Ps[0].shape[1], nr))
            if nP > 1:                                                      # every profile replays the streams of its own invocation
                meas_noise = np.tile(meas_noise, (nP, 1, 1))
                step_noise = np.tile(step_noise, (1, nP, 1, 1))
        else:
            init = shared_init(B, nt, nr, seed, meta_idx)                   # one init for all SNR points (:115,126)
        # trajectory t = (profile * S + snr) * B + channel; its noise key is the one of the single-profile run: meta * S * B + snr * B + channel
        t_all = np.arange(nP * S * B)
        prof_of, snr_of, ch_of = t_all // (S * B), (t_all // B) % S, t_all % B
        out = run_trajectories(diffuser, val_H, val_P, prof_of * B + ch_of, prof_of * B + ch_of, noise_range[snr_of], alpha_step,
                               beta_noise, levels, config.sampling.steps_each, seed, init, init_index=ch_of,
                               traj_id=meta_idx * S * B + snr_of * B + ch_of, use_graph=resolve_launch_mode(args),
                               max_batch=args.max_batch, rank=rank, world=world, return_final=bool(args.save_channels),
                               n_streams=args.streams, step_noise=step_noise, meas_noise=meas_noise, info=run_info)
        if args.save_channels:
            log, est = out
            if saved_H is None:
                saved_H = np.zeros((nP, len(spacing_range), len(pilot_alpha_range), S, B, nt, nr), np.complex64)
            saved_H[:, spacing_idx, pilot_alpha_idx] = est.reshape(nP, S, B, nt, nr)
        else:
            log = out
        nmse_log[:, spacing_idx, pilot_alpha_idx] = log.reshape(n_steps, nP, S, B).transpose(1, 2, 0, 3)

    avg_nmse = np.mean(nmse_log, axis=-1)                 # :174
    best_nmse = np.min(avg_nmse, axis=-1)                 # :175  (best stopping step per SNR)
    if rank == 0:
        for pi, test in enumerate(tests):
            if not args.no_plot:
                try:
                    import matplotlib
                    matplotlib.use('Agg')
                    from matplotlib import pyplot as plt
                    plt.rcParams['font.size'] = 14
                    plt.figure(figsize=(10, 10))
                    for alpha_idx, local_alpha in enumerate(pilot_alpha_range):
                        plt.plot(snr_range, 10 * np.log10(best_nmse[pi, 0, alpha_idx]), linewidth=4,
                                 label='Alpha=%.2f' % local_alpha)
                    plt.grid(); plt.legend()
                    plt.title('Score-based channel estimation')
                    plt.xlabel('SNR [dB]'); plt.ylabel('NMSE [dB]')
                    plt.tight_layout()
                    plt.savefig(os.path.join(result_dirs[pi], 'results.png'), dpi=300, bbox_inches='tight')
                    plt.close()
                except ImportError:
                    print('matplotlib not available: skipping results.png')
            results = {'nmse_log': nmse_log[pi], 'avg_nmse': avg_nmse[pi], 'best_nmse': best_nmse[pi],
                       'spacing_range': spacing_range, 'pilot_alpha_range': pilot_alpha_range, 'snr_range': snr_range,
                       'val_config': val_configs[pi], 'seed': seed, 'levels': np.asarray(levels)}
            if saved_H is not None:
                # the reference parses --save_channels but never uses it (test_score.py:19); here it stores the estimates
                # after the last Langevin step, normalised Hermitian layout [.., Nt, Nr] like val_H (:112-113)
                results['saved_H'] = saved_H[pi]
            # chunks of ANY rank whose f16x2 launches raised the range flag and were run again in bf16x3 (empty list: none);
            # with several profiles `trajectories` are positions in the flattened (profile, SNR, channel) list
            results['f16x2_fallback'] = run_info.get('f16x2_fallback', [])
            torch.save(results, os.path.join(result_dirs[pi], 'results.pt'))
            print('%sbest NMSE [dB] per SNR:' % ('' if nP == 1 else '[%s] ' % test), np.round(10 * np.log10(best_nmse[pi, 0, 0]), 2))
    if nP == 1:
        return nmse_log[0], avg_nmse[0], best_nmse[0]
    return {t: (nmse_log[i], avg_nmse[i], best_nmse[i]) for i, t in enumerate(tests)}


if __name__ == '__main__':
    main()
