"""Approximate-MMSE channel estimation: the posterior mean over several annealed-Langevin chains per sample.

Counterpart of the reference ``src/score_based_channels/test_mmse.py`` (Fig. 5c of the paper) on the same HIP hot path
as ``test_score``: per SNR point every kept validation channel is sampled ``mmse_avg`` (50) times from the SAME
measurement with independent start points and annealing noise, using that SNR's tuned (step, noise, stop) triple with
early stopping (test_mmse.py:170-255); the chains' estimates are stored (``saved_H``) together with their NMSE log
(``oracle_log``), and -- added here, since it is what the stored chains are for -- their average (``mmse_H``) and its
NMSE (``mmse_nmse``).  All ``kept_samples x mmse_avg`` chains of an SNR point form one lock-step batch (5000
trajectories), sharded over ranks like ``test_score``.

Arguments of the reference script are kept (test_mmse.py:15-27); additions of this build are marked in ``--help``.
"""
import argparse
import copy
import itertools
import os

import numpy as np

from .config import CONV_MODES, DEFAULT_CONV_MODE, DEFAULT_STREAMS
import torch

from .ald import AldBatch
from .checkpoint import load_checkpoint
from .config import default_config
from .driver import level_subset, resolve_launch_mode, run_trajectories
from .loaders import Channels
from .noise import HostNoise
from .scorenet import ScoreNet
from . import shard
from .shard import init_distributed
from .weights import get_sigmas, seeded_state_dict

DEEP_MIMO = ['DeepMIMO_outdoor', 'DeepMIMO_indoor_1', 'DeepMIMO_indoor_3_nlos']


def reference_weight_path(model):
    """Checkpoint locations hard-coded in test_mmse.py:41-58."""
    if model == 'CDL-D':
        return './models_oct14/numLambdas2_lambdaMin0.1_lambdaMax0.5_sigmaT39.1/final_model.pt'
    if model == 'CDL-C':
        return './models_jan29_2022_CDL-C/numLambdas1_lambdaMin0.5_lambdaMax0.5_sigmaT27.8/final_model.pt'
    if model in ('CDL-B', 'CDL-A'):
        return './models_feb2_%s/numLambdas1_lambdaMin0.5_lambdaMax0.5_sigmaT31.2/final_model.pt' % model
    if model in DEEP_MIMO:
        return './models_feb2_DeepMIMO_%s/numLambdas1_lambdaMin0.5_lambdaMax0.5_sigmaT27.8/final_model.pt' % model[9:]
    if model == 'all':
        return './models_feb23_multi/numLambdas1_lambdaMin0.5_lambdaMax0.5_sigmaT27.8/final_model.pt'
    raise ValueError('unknown model %r' % model)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    p.add_argument('--gpu', type=int, default=1)
    p.add_argument('--model', type=str, default='CDL-C')
    p.add_argument('--channel', type=str, default='CDL-C')
    p.add_argument('--start_point', type=str, default='Noise', choices=['Noise', 'Adjoint', 'LS'])
    p.add_argument('--spacing', nargs='+', type=float, default=[0.5])
    p.add_argument('--pilot_alpha', nargs='+', type=float, default=[0.6])
    p.add_argument('--steps_each', type=int, default=3)
    p.add_argument('--normalize_grad', type=bool, default=False)       # parsed, unused (as in the reference)
    p.add_argument('--dc_boost', type=float, default=1)
    p.add_argument('--num_classes', type=int, default=2311)
    # additions of this build
    p.add_argument('--seed', type=int, default=None, help='[added] seed of every noise stream (default: fresh entropy)')
    p.add_argument('--kept_samples', type=int, default=100, help='[added] validation channels kept (test_mmse.py:103)')
    p.add_argument('--mmse_avg', type=int, default=50, help='[added] chains per channel (test_mmse.py:104)')
    p.add_argument('--levels_stride', type=int, default=1, help='[added] walk every k-th noise level (+ the last)')
    p.add_argument('--num_levels', type=int, default=None, help='[added] walk only the first n selected levels')
    p.add_argument('--hyper_file', type=str, default=None,
                   help='[added] tuned per-SNR hyper-parameters (default ./our_hyperparams_<model>.pt, test_mmse.py:123)')
    p.add_argument('--synthetic', action='store_true', help='[added] generated CDL-like channels instead of ./data')
    p.add_argument('--synthetic_weights', type=int, default=None, metavar='SEED', help='[added] seed-derived weights')
    p.add_argument('--conv_mode', type=str, default=DEFAULT_CONV_MODE, choices=list(CONV_MODES), help='[added]')
    p.add_argument('--noise', type=str, default='device', choices=['device', 'host'],
                   help='[added] in-kernel Philox noise, or the keyed host streams of noise.HostNoise (parity runs)')
    p.add_argument('--no_graph', action='store_true', help='[added] eager launches instead of hipGraph replay')
    p.add_argument('--graph', action='store_true', help='[added] replay each Langevin step as a hipGraph (default: driver.DEFAULT_USE_GRAPH)')
    p.add_argument('--streams', type=int, default=DEFAULT_STREAMS,
                   help='[added] run each lock-step batch as this many concurrent sub-batches on their own HIP streams '
                        '(bit-identical results; +7 %% at 2 on MI355X for 1700 trajectories)')
    p.add_argument('--result_dir', type=str, default=None, help='[added] default TWC_rebuttal_MMSE_aug6_seed4321')
    p.add_argument('--force_dist', action='store_true',
                   help='[added] with ONE rank: still create the torch.distributed group (backend nccl = RCCL on a GPU box) and route the seed '
                        'broadcast, the agreement points and the final gathers through it')
    return p.parse_args(argv)


def start_points(kind, P_herm, Y, n_chains, nt, nr, seed, key):
    """Initial estimates ``[n_samples * n_chains, Nt, Nr]`` (test_mmse.py:196-203).  ``P_herm`` is ``[B, Np, Nt]``
    (the conj-transposed pilots, the forward operator), ``Y`` ``[B, Np, Nr]``."""
    B = P_herm.shape[0]
    if kind == 'Noise':                                   # one CN(0,1) draw per chain
        g = torch.Generator().manual_seed((int(seed) * 1000003 + int(key)) % (2 ** 63 - 1))
        return torch.randn(B * n_chains, nt, nr, dtype=torch.complex64, generator=g)
    if kind == 'Adjoint':                                 # P^H y, the same for every chain of a sample
        x = np.matmul(np.conj(np.transpose(P_herm, (0, 2, 1))), Y)
    else:                                                 # 'LS': minimum-norm least squares per sample
        x = np.stack([np.linalg.lstsq(P_herm[b], Y[b], rcond=None)[0] for b in range(B)])
    return torch.from_numpy(np.repeat(x.astype(np.complex64), n_chains, axis=0))


def posterior_chains(diffuser, val_H, val_P, local_noise, step, noise_boost, n_run, levels, steps_each, navg,
                     start_point='Noise', seed=0, key=0, dc_boost=1.0, use_graph=None, rank=0, world=1, host_noise=None,
                     n_streams=None):
    """All chains of ONE SNR point (test_mmse.py:170-262): one measurement per kept sample shared by its ``navg`` chains
    (:176-193), start points (:196-203), ``n_run`` Langevin steps with that SNR's (step, noise) pair and ``dc_boost``
    (:216-233), early stop (:246-250).  ``val_H`` ``[kept, Nt, Nr]``, ``val_P`` ``[kept, Np, Nt]`` complex64 numpy.

    Noise: in-kernel Philox streams keyed by (``seed``, ``key``), or -- ``host_noise`` = a ``noise.HostNoise`` -- that
    object's measurement draw 0, ``init`` draw (start point 'Noise') and step stream 0, i.e. the draws the reference
    goldens were generated with.  Returns (Y ``[kept, Np, Nr]`` torch, log ``[n_run, kept, navg]``, estimates
    ``[kept, navg, Nt, Nr]``)."""
    kept, nt, nr = val_H.shape
    npil = val_P.shape[1]
    h_index = np.repeat(np.arange(kept), navg)                          # trajectory = sample * mmse_avg + chain
    base = key * kept * (navg + 1)
    # one measurement per sample (measure kernel), shared by its chains
    meas = AldBatch(diffuser, val_H, val_P, np.arange(kept), np.arange(kept), local_noise, levels=levels[:1],
                    steps_each=1, seed=seed, traj_id=base + np.arange(kept))
    mz = None if host_noise is None else torch.from_numpy(host_noise.measurement(0, (kept, npil, nr)))
    Y = meas.synthesize_measurements(mz).clone()
    meas.close()
    step_noise = None
    if host_noise is not None:
        step_noise = host_noise.step_block(0, (kept * navg, nt, nr), n_run)
    if host_noise is not None and start_point == 'Noise':
        init = torch.from_numpy(host_noise.init((kept * navg, nt, nr)))
    else:
        init = start_points(start_point, val_P, Y.cpu().numpy(), navg, nt, nr, seed, key)
    log, est = run_trajectories(
        diffuser, val_H, val_P, h_index, h_index, local_noise, step, noise_boost, levels, steps_each, seed,
        init, traj_base=base + kept, max_batch=8192, use_graph=use_graph, rank=rank, world=world, return_final=True,
        n_steps=n_run, dc_boost=float(dc_boost), init_index=np.arange(kept * navg), Y=Y, y_index=h_index,
        step_noise=step_noise, n_streams=n_streams)
    return Y, log.reshape(n_run, kept, navg), est.reshape(kept, navg, nt, nr)


def main(argv=None):
    args = parse_args(argv)
    rank, world, local = init_distributed(force=getattr(args, 'force_dist', False))
    # (a rank that fails tells the others at their next agreement point instead of leaving them in a collective: shard.run_guarded)
    return shard.run_guarded(world, lambda: _main(args, rank, world, local))


def _main(args, rank, world, local):
    if not torch.cuda.is_available():
        raise RuntimeError('test_mmse needs a HIP device (there is no CPU fallback)')
    device = 'cuda:%d' % (local if world > 1 else min(args.gpu, torch.cuda.device_count() - 1))
    torch.cuda.set_device(device)

    if args.synthetic_weights is not None:
        config = default_config(args.model)
        model_state = seeded_state_dict(config, args.synthetic_weights)
    else:
        contents = load_checkpoint(reference_weight_path(args.model))
        config, model_state = contents['config'], contents['model_state']
    config.sampling.sigma = 0.
    config.purpose = 'train'
    if int(args.num_classes) != int(config.model.num_classes):         # "more sigmas" (test_mmse.py:66-71)
        config.model.num_classes = int(args.num_classes)
        config.model.sigma_rate = (config.model.sigma_end / config.model.sigma_begin) ** (1 / (config.model.num_classes - 1))
        model_state = dict(model_state)
        model_state['sigmas'] = get_sigmas(config)
    diffuser = ScoreNet(config, device, conv_mode=args.conv_mode).load_state_dict(model_state).eval()

    seed = int.from_bytes(os.urandom(4), 'little') if args.seed is None else args.seed
    if world > 1 or getattr(args, 'force_dist', False):
        seed = shard.broadcast_int(seed, 0, device)
    np.random.seed(seed % (2 ** 32))

    train_seed, val_seed = 1234, 4321
    config.data.channel = args.model
    config.data.array = 'ULA'
    dataset = Channels(train_seed, config, norm=config.data.norm_channels, synthetic=args.synthetic)

    steps_each = int(args.steps_each)
    levels = level_subset(config.model.num_classes, args.levels_stride, args.num_levels)
    total_steps = len(levels) * steps_each
    snr_range = np.arange(-30, 17.5, 2.5)
    spacing_range = np.asarray(args.spacing)
    pilot_alpha_range = np.asarray(args.pilot_alpha)
    noise_range = 10 ** (-snr_range / 10.)                              # no Nt factor here (test_mmse.py:100)
    kept, navg = int(args.kept_samples), int(args.mmse_avg)
    nt, nr = config.data.image_size[1], config.data.image_size[0]
    S = len(snr_range)

    # tuned hyper-parameters per (pilot alpha, SNR): step size, annealing-noise factor, stopping step
    hyper_file = args.hyper_file or 'our_hyperparams_%s.pt' % args.model
    if os.path.exists(hyper_file):
        hp = torch.load(hyper_file, weights_only=False)
        best_step, best_noise, best_stop = (np.asarray(hp[k]) for k in ('best_step_idx', 'best_noise_idx', 'best_stop_idx'))
    else:
        if rank == 0:
            print('%s not found: using the test_score defaults (3e-11, 0.01, full schedule) for every SNR' % hyper_file)
        shape = (len(pilot_alpha_range), S)
        best_step, best_noise = np.full(shape, 3e-11), np.full(shape, 0.01)
        best_stop = np.full(shape, total_steps - 1, dtype=np.int64)

    oracle_log = np.zeros((len(spacing_range), len(pilot_alpha_range), S, total_steps, kept, navg))
    saved_H = np.zeros((len(spacing_range), len(pilot_alpha_range), S, kept, navg, nt, nr), np.complex64)
    result_dir = args.result_dir or 'TWC_rebuttal_MMSE_aug6_seed%d' % val_seed
    if rank == 0:
        os.makedirs(result_dir, exist_ok=True)

    val_config, oracle_H = None, None
    for meta_idx, (spacing, pilot_alpha) in enumerate(itertools.product(spacing_range, pilot_alpha_range)):
        spacing_idx, pilot_alpha_idx = np.unravel_index(meta_idx, (len(spacing_range), len(pilot_alpha_range)))
        val_config = copy.deepcopy(config)
        val_config.purpose = 'val'
        val_config.data.channel = args.channel
        val_config.data.spacing_list = [spacing]
        val_config.data.num_pilots = int(np.floor(config.data.num_pilots * pilot_alpha))      # test_mmse.py:141
        norm = [0., 1.] if args.model in DEEP_MIMO else [dataset.mean, dataset.std]
        val_dataset = Channels(val_seed, val_config, norm=norm, synthetic=args.synthetic)
        if rank == 0:
            print('There are %d validation channels!' % len(val_dataset))
        sample = val_dataset.batch(kept)
        val_P = np.conj(np.transpose(sample['P'], (0, 2, 1)))           # [B, Np, Nt]
        val_H = sample['H_herm'][:, 0] + 1j * sample['H_herm'][:, 1]     # [B, Nt, Nr]
        oracle_H = val_H
        for snr_idx, local_noise in enumerate(noise_range):
            step = float(best_step[pilot_alpha_idx, snr_idx])
            noise_boost = float(best_noise[pilot_alpha_idx, snr_idx])
            n_run = min(int(best_stop[pilot_alpha_idx, snr_idx]) + 1, total_steps)     # early stop (:246-250)
            key = (meta_idx * S + snr_idx)
            host = HostNoise(seed, combo=1 + key) if args.noise == 'host' else None
            _, log, est = posterior_chains(diffuser, val_H, val_P, local_noise, step, noise_boost, n_run, levels,
                                           steps_each, navg, args.start_point, seed, key, args.dc_boost,
                                           resolve_launch_mode(args), rank, world, host, args.streams)
            oracle_log[spacing_idx, pilot_alpha_idx, snr_idx, :n_run] = log
            saved_H[spacing_idx, pilot_alpha_idx, snr_idx] = est
            if rank == 0:
                print('SNR %.1f dB: early stopping at step %d' % (snr_range[snr_idx], n_run - 1))

    mmse_H = saved_H.mean(axis=4)                                       # posterior-mean estimate per sample
    num = np.sum(np.abs(mmse_H - oracle_H[None, None, None]) ** 2, axis=(-1, -2))
    mmse_nmse = num / np.sum(np.abs(oracle_H) ** 2, axis=(-1, -2))[None, None, None]
    if rank == 0:
        torch.save({'spacing_range': spacing_range, 'pilot_alpha_range': pilot_alpha_range, 'args': args,
                    'config': config, 'snr_range': snr_range, 'val_config': val_config, 'oracle_log': oracle_log,
                    'oracle_H': oracle_H, 'saved_H': saved_H, 'mmse_H': mmse_H, 'mmse_nmse': mmse_nmse, 'seed': seed},
                   os.path.join(result_dir, 'model_%s_channel_%s.pt' % (args.model, args.channel)))
        print('MMSE-estimate NMSE [dB] per SNR:', np.round(10 * np.log10(mmse_nmse[0, 0].mean(-1)), 2))
    return oracle_log, saved_H, mmse_nmse


if __name__ == '__main__':
    main()
