#!/usr/bin/env python3
"""Hyper-parameter search of the annealed-Langevin estimator -- counterpart of
``src/score_based_channels/tune_hparams_score.py`` on the MI355X HIP path.

    python -m score_based_channels_amd.tune_hparams_score --channel CDL-C --spacing 0.5 --pilot_alpha 0.6

Same arguments and output file (``./results/score/<channel>-hyperparameters.{pt,png}`` with the keys of
tune_hparams_score.py:180-189).  The (alpha_step x beta_noise) grid cells x 17 SNR points x channels are all
independent trajectories; they run as lock-step batches sharded over the GPUs (``torch.distributed.run``), with
one gather of the NMSE logs.  "N" is not a grid axis: the best stopping step is the argmin over the logged step
axis (tune_hparams_score.py:151-152).  Like the reference, every grid cell re-creates the validation dataset
(fresh pilots from numpy's global RNG) and draws its own initial estimate (:77-97).
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
    p.add_argument('--channel', type=str, default='CDL-C')
    p.add_argument('--spacing', type=float, default=0.5)
    p.add_argument('--alpha_step_range', nargs='+', type=float, default=[3e-11, 6e-11, 1e-10, 3e-10])
    p.add_argument('--beta_noise_range', nargs='+', type=float, default=[0.1, 0.01, 0.001])
    p.add_argument('--pilot_alpha', type=float, default=0.6)
    # additions of this build
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--levels_stride', type=int, default=1)
    p.add_argument('--num_levels', type=int, default=None)
    p.add_argument('--num_channels', type=int, default=100)
    p.add_argument('--steps_each', type=int, default=None,
                   help='Langevin steps per level; default: config.sampling.steps_each of the checkpoint, else 3')
    p.add_argument('--synthetic', action='store_true')
    p.add_argument('--synthetic_weights', type=int, default=None, metavar='SEED')
    p.add_argument('--no_plot', action='store_true')
    p.add_argument('--conv_mode', type=str, default=DEFAULT_CONV_MODE, choices=list(CONV_MODES),
                   help='convolution multiplier: split-bf16 matrix cores (fp32-accurate, default), fp32 MFMA, or fp16 '
                        'weights on the fp16 matrix cores (looser tolerance)')
    p.add_argument('--noise', type=str, default='device', choices=['device', 'host'],
                   help='in-kernel Philox noise [device] or the keyed host streams of noise.HostNoise [host, parity runs]')
    p.add_argument('--no_graph', action='store_true')
    p.add_argument('--graph', action='store_true', help='[added] replay each Langevin step as a hipGraph (default: driver.DEFAULT_USE_GRAPH)')
    p.add_argument('--streams', type=int, default=DEFAULT_STREAMS,
                   help='[added] run each lock-step batch as this many concurrent sub-batches on their own HIP streams '
                        '(bit-identical results; +7 %% at 2 on MI355X for 1700 trajectories)')
    p.add_argument('--force_dist', action='store_true',
                   help='[added] with ONE rank: still create the torch.distributed group (backend nccl = RCCL on a GPU box) and route the seed '
                        'broadcast, the agreement points and the final gathers through it')
    return p.parse_args(argv)


def select_best(best_nmse, alpha_step_range, beta_noise_range):
    """Best (alpha, beta) per SNR point (tune_hparams_score.py:155-162)."""
    best_alpha_snr, best_beta_snr = [], []
    for snr_idx in range(best_nmse.shape[-1]):
        flat = best_nmse[..., snr_idx].flatten()
        a_idx, b_idx = np.unravel_index(np.argmin(flat), (len(alpha_step_range), len(beta_noise_range)))
        best_alpha_snr.append(alpha_step_range[a_idx])
        best_beta_snr.append(beta_noise_range[b_idx])
    return best_alpha_snr, best_beta_snr


def main(argv=None):
    args = parse_args(argv)
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
    if args.synthetic_weights is not None:
        config = default_config(args.channel)
        model_state = seeded_state_dict(config, args.synthetic_weights)
    else:
        contents = load_checkpoint(os.path.join('./models/score/%s' % args.channel, 'final_model.pt'))
        config, model_state = contents['config'], contents['model_state']
    if args.steps_each is not None:
        config.sampling.steps_each = args.steps_each
    elif not config.sampling.steps_each:                  # the reference raises TypeError here (:62); default to 3
        config.sampling.steps_each = 3
    diffuser = ScoreNet(config, device, conv_mode=args.conv_mode).load_state_dict(model_state).eval()

    seed = int.from_bytes(os.urandom(4), 'little') if args.seed is None else args.seed
    if world > 1 or getattr(args, 'force_dist', False):
        seed = shard.broadcast_int(seed, 0, device)
    np.random.seed(seed % (2 ** 32))

    train_seed, val_seed = 1234, 4321
    config.data.channel = args.channel
    dataset = Channels(train_seed, config, norm=config.data.norm_channels, synthetic=args.synthetic)
    snr_range = np.arange(-10, 32.5, 2.5)
    alpha_step_range = np.asarray(args.alpha_step_range)
    beta_noise_range = np.asarray(args.beta_noise_range)
    nt = config.data.image_size[1]
    noise_range = 10 ** (-snr_range / 10.) * nt
    B, S = args.num_channels, len(snr_range)
    levels = level_subset(config.model.num_classes, args.levels_stride, args.num_levels)
    steps_each = int(config.sampling.steps_each)
    n_steps = len(levels) * steps_each

    # one validation dataset / pilot draw / initial estimate per grid cell, in the reference's order
    cells = list(itertools.product(alpha_step_range, beta_noise_range))
    Hs, Ps, inits, meas_nz, step_nz = [], [], [], [], []
    for meta_idx, _ in enumerate(cells):
        val_config = copy.deepcopy(config)
        val_config.data.channel = args.channel
        val_config.data.spacing_list = [args.spacing]
        val_config.data.num_pilots = int(np.floor(nt * args.pilot_alpha))
        val_dataset = Channels(val_seed, val_config, norm=[dataset.mean, dataset.std], synthetic=args.synthetic)
        sample = val_dataset.batch(B)
        Ps.append(np.conj(np.transpose(sample['P'], (0, 2, 1))))
        Hs.append(sample['H_herm'][:, 0] + 1j * sample['H_herm'][:, 1])
        if args.noise == 'host':                          # draw order of the reference: init, then per SNR (Y noise, steps)
            i0, m0, s0 = host_noise_streams(seed, meta_idx, Hs[-1].shape, S, n_steps, (B, Ps[-1].shape[1], Hs[-1].shape[-1]))
            inits.append(i0); meas_nz.append(m0); step_nz.append(s0)
        else:
            inits.append(shared_init(B, nt, Hs[-1].shape[-1], seed, meta_idx))
    H_all, P_all, init_all = np.concatenate(Hs), np.concatenate(Ps), torch.cat(inits)
    meas_noise = np.concatenate(meas_nz, axis=0) if meas_nz else None
    step_noise = np.concatenate(step_nz, axis=1) if step_nz else None
    # trajectory t = (cell * S + snr) * B + channel
    cell_of = np.repeat(np.arange(len(cells)), S * B)
    chan_of = np.tile(np.arange(B), len(cells) * S)
    h_index = cell_of * B + chan_of
    ln = np.tile(np.repeat(noise_range, B), len(cells))
    a0 = np.asarray([c[0] for c in cells])[cell_of]
    be = np.asarray([c[1] for c in cells])[cell_of]
    run_info = {}                                         # 'f16x2_fallback': chunks this rank re-ran in bf16x3 (driver.py)
    log = run_trajectories(diffuser, H_all, P_all, h_index, h_index, ln, a0, be, levels, steps_each, seed, init_all,
                           use_graph=resolve_launch_mode(args), rank=rank, world=world, n_streams=args.streams,
                           step_noise=step_noise,
                           meas_noise=meas_noise, info=run_info)
    nmse_log = log.reshape(n_steps, len(alpha_step_range), len(beta_noise_range), S, B).transpose(1, 2, 3, 0, 4)
    nmse_log = nmse_log.astype(np.float64)

    avg_nmse = np.mean(nmse_log, axis=-1)                 # :151
    best_nmse = np.min(avg_nmse, axis=-1)                 # :152
    best_alpha_snr, best_beta_snr = select_best(best_nmse, alpha_step_range, beta_noise_range)
    if rank == 0:
        result_dir = './results/score'
        os.makedirs(result_dir, exist_ok=True)
        if not args.no_plot:
            try:
                import matplotlib
                matplotlib.use('Agg')
                from matplotlib import pyplot as plt
                plt.rcParams['font.size'] = 14
                plt.figure(figsize=(10, 10))
                for a_idx, la in enumerate(alpha_step_range):
                    for b_idx, lb in enumerate(beta_noise_range):
                        plt.plot(snr_range, 10 * np.log10(best_nmse[a_idx, b_idx]), linewidth=4,
                                 label='Alpha=%.2e, Beta=%.2e' % (la, lb))
                plt.grid(); plt.legend()
                plt.title('Score-based hyperparameter search')
                plt.xlabel('SNR [dB]'); plt.ylabel('NMSE [dB]')
                plt.tight_layout()
                plt.savefig(os.path.join(result_dir, '%s-hyperparameters.png' % args.channel), dpi=300,
                            bbox_inches='tight')
                plt.close()
            except ImportError:
                print('matplotlib not available: skipping the plot')
        torch.save({'nmse_log': nmse_log, 'avg_nmse': avg_nmse, 'best_nmse': best_nmse,
                    'best_alpha_snr': best_alpha_snr, 'best_beta_snr': best_beta_snr, 'snr_range': snr_range,
                    'alpha_step_range': alpha_step_range, 'beta_noise_range': beta_noise_range, 'config': config,
                    'args': args, 'seed': seed, 'levels': np.asarray(levels),
                    'f16x2_fallback': run_info.get('f16x2_fallback', [])},
                   os.path.join(result_dir, '%s-hyperparameters.pt' % args.channel))
        print('best alpha per SNR:', best_alpha_snr)
        print('best beta  per SNR:', best_beta_snr)
    return nmse_log, best_alpha_snr, best_beta_snr


if __name__ == '__main__':
    main()
