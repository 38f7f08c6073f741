#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE on CPU.

Runs only in the build container, where the read-only reference checkout exists at
/root/reference; nothing here (and nothing under tests/golden/) contains reference source --
the fixtures are inputs and outputs only.  What is executed:

* network goldens: the reference's ``ncsnv2.models.ncsnv2.NCSNv2Deepest`` (imported), loaded with
  this repo's seed-derived weights (``score_based_channels_amd.weights.seeded_state_dict``);
* loop goldens: that network inside a torch transcription of the sampling loop of
  ``src/score_based_channels/test_score.py:118-171`` -- the scripts themselves cannot be imported
  (module-level argparse, ``.cuda()``, missing blobs), so the loop body is driven here with the same
  torch complex64 operations, but with every ``randn_like`` replaced by the keyed host streams of
  ``score_based_channels_amd.noise.HostNoise``;
* loader goldens: the reference's ``loaders.Channels`` (imported with a stand-in ``hdf5storage``
  module that returns a synthetic ``output_h``).

Usage:  python tests/gen_golden.py [forward plumbing trunc big f16w cross mmse loader tune train full]
"""
import os
import sys
import types
import time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, 'src'))

from score_based_channels_amd.config import Config, default_config          # noqa: E402
from score_based_channels_amd.weights import seeded_state_dict              # noqa: E402
from score_based_channels_amd.noise import HostNoise                        # noqa: E402
from score_based_channels_amd import synth                                  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
WEIGHT_SEED = 2024


def reference_net(config, sd):
    from ncsnv2.models.ncsnv2 import NCSNv2Deepest
    cfg = Config.from_mapping(config.toDict())
    cfg.device = 'cpu'
    net = NCSNv2Deepest(cfg)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net.eval()


def case_inputs(seed, B, nt, nr, pilot_alpha, profile='CDL-C'):
    """Normalised Hermitian channels ``[B,Nt,Nr]`` and conj-transposed pilots ``[B,Np,Nt]``
    (the tensors ``val_H`` / ``val_P`` of test_score.py:109-113)."""
    raw = synth.generate_channels(profile, max(B, 16), nt, nr, 0.5, seed)
    Hn = raw[:B] / np.std(raw)                                   # 'global' norm, loaders.py:47-49,69
    H = np.conj(np.transpose(Hn, (0, 2, 1))).astype(np.complex64)
    npil = int(np.floor(nt * pilot_alpha))                       # test_score.py:100
    pil = synth.qpsk_pilots(np.random.default_rng([seed, 77]), B, nt, npil)
    P = np.conj(np.transpose(pil, (0, 2, 1))).astype(np.complex64)
    return H, P


def reference_ald(net, config, H, P, snr_db, levels, seed, steps_each=3, alpha_step=3e-11,
                  beta_noise=0.01, combo=0):
    """test_score.py:115-171 with keyed noise; returns Y, final estimates and the NMSE log."""
    nt = H.shape[1]
    noise = HostNoise(seed, combo)
    val_P, val_H = torch.from_numpy(P), torch.from_numpy(H)
    init_val_H = torch.from_numpy(noise.init(H.shape))
    noise_range = 10 ** (-np.asarray(snr_db, np.float64) / 10.) * nt
    Ys, finals, logs = [], [], []
    for snr_idx, local_noise in enumerate(noise_range):
        val_Y = torch.matmul(val_P, val_H)
        val_Y = val_Y + np.sqrt(local_noise) * torch.from_numpy(
            noise.measurement(snr_idx, tuple(val_Y.shape)))
        current = init_val_H.clone()
        forward, forward_h = val_P, torch.conj(torch.transpose(val_P, -1, -2))
        draw = noise.step_stream(snr_idx, H.shape)
        log = np.zeros((len(levels) * steps_each, H.shape[0]), np.float32)
        k = 0
        for step_idx in levels:
            current_sigma = net.sigmas[step_idx].item()
            labels = (torch.ones(H.shape[0]) * step_idx).long()
            alpha = alpha_step * (current_sigma / config.model.sigma_end) ** 2
            for _ in range(steps_each):
                current_real = torch.view_as_real(current).permute(0, 3, 1, 2)
                with torch.no_grad():
                    score = net(current_real, labels)
                score = torch.view_as_complex(score.permute(0, 2, 3, 1).contiguous())
                meas_grad = torch.matmul(forward_h, torch.matmul(forward, current) - val_Y)
                grad_noise = np.sqrt(2 * alpha * beta_noise) * torch.from_numpy(draw(k))
                current = current + alpha * (score - meas_grad /
                                             (local_noise / 2. + current_sigma ** 2)) + grad_noise
                log[k] = (torch.sum(torch.square(torch.abs(current - val_H)), dim=(-1, -2)) /
                          torch.sum(torch.square(torch.abs(val_H)), dim=(-1, -2))).numpy()
                k += 1
        Ys.append(val_Y.numpy())
        finals.append(current.numpy())
        logs.append(log)
    return np.stack(Ys), np.stack(finals), np.stack(logs)


def gen_forward():
    cfg = default_config()
    sd = seeded_state_dict(cfg, WEIGHT_SEED)
    net = reference_net(cfg, sd)
    keys = [(k, list(v.shape)) for k, v in net.state_dict().items()]
    import json
    with open(os.path.join(GOLD, 'state_dict_keys.json'), 'w') as f:
        json.dump(keys, f)
    H, _ = case_inputs(11, 4, 64, 16, 0.6)
    rng = np.random.default_rng(5)
    x = np.stack((H.real, H.imag), 1).astype(np.float32)
    x = x + (0.5 * rng.standard_normal(x.shape)).astype(np.float32)
    levels = [0, 1155, 2310]
    outs = []
    with torch.no_grad():
        for lv in levels:
            outs.append(net(torch.from_numpy(x), torch.full((4,), lv, dtype=torch.long)).numpy())
    # intermediate feature maps of sample 0 (forward hooks on the reference modules)
    stages = {}
    hooks = []
    for name in ['begin_conv', 'res1', 'res2', 'res3', 'res31', 'res4', 'res5', 'refine1', 'refine2',
                 'refine31', 'refine3', 'refine4', 'refine5']:
        mod = getattr(net, name)
        tgt = mod[-1] if isinstance(mod, torch.nn.ModuleList) else mod
        hooks.append(tgt.register_forward_hook(
            lambda m, i, o, name=name: stages.__setitem__(name, o.detach().numpy().copy())))
    with torch.no_grad():
        net(torch.from_numpy(x[:1]), torch.full((1,), 1155, dtype=torch.long))
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(GOLD, 'forward_64x16.npz'), x=x, levels=np.array(levels),
                        out=np.stack(outs), weight_seed=WEIGHT_SEED,
                        **{'stage_' + k: v.astype(np.float32) for k, v in stages.items()})
    print('forward_64x16: out absmax', [float(np.abs(o).max()) for o in outs])


def _save_ald(name, cfg, B, snr_db, levels, seed, nt=64, nr=16, weight_seed=WEIGHT_SEED, **kw):
    sd = seeded_state_dict(cfg, weight_seed)
    net = reference_net(cfg, sd)
    H, P = case_inputs(seed, B, nt, nr, 0.6)
    t = time.time()
    Y, X, log = reference_ald(net, cfg, H, P, snr_db, levels, seed, **kw)
    print('%s: %d steps x %d snr, B=%d in %.1f s; final NMSE %s' % (
        name, log.shape[1], len(snr_db), B, time.time() - t, log[:, -1].mean(-1)))
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), H=H, P=P, Y=Y, X_final=X, nmse_log=log,
                        snr_db=np.asarray(snr_db, np.float64), levels=np.asarray(levels), seed=seed,
                        weight_seed=weight_seed, steps_each=kw.get('steps_each', 3),
                        alpha_step=kw.get('alpha_step', 3e-11), beta_noise=kw.get('beta_noise', 0.01))


def gen_plumbing():
    cfg = default_config()
    _save_ald('ald_plumbing_level0', cfg, 4, [0.0], [0], seed=101)          # config 1: N = 3 steps
    _save_ald('ald_plumbing_3levels', cfg, 4, [0.0], [0, 1, 2], seed=102)


def gen_trunc():
    cfg = default_config()
    levels = list(range(0, 2311, 77)) + [2310]
    _save_ald('ald_trunc', cfg, 8, [-10.0, 10.0, 30.0], levels, seed=103)
    # a non-default (alpha, beta) cell of the tune grid (tune_hparams_score.py:20-23)
    _save_ald('ald_trunc_cell', cfg, 4, [5.0], levels, seed=104, alpha_step=3e-10, beta_noise=0.1)


def gen_full():
    cfg = default_config()
    _save_ald('ald_full', cfg, 4, [10.0], list(range(2311)), seed=105)


def gen_big():
    cfg = default_config(image_size=(64, 256))
    sd = seeded_state_dict(cfg, WEIGHT_SEED)
    net = reference_net(cfg, sd)
    H, P = case_inputs(21, 1, 256, 64, 0.6, profile='ULA')
    x = np.stack((H.real, H.imag), 1).astype(np.float32)
    with torch.no_grad():
        out = net(torch.from_numpy(x), torch.full((1,), 1155, dtype=torch.long)).numpy()
    Y, X, log = reference_ald(net, cfg, H, P, [10.0], [0, 1000], 106)
    np.savez_compressed(os.path.join(GOLD, 'big_256x64.npz'), x=x, out=out, level=1155, H=H, P=P,
                        Y=Y, X_final=X, nmse_log=log, snr_db=np.array([10.0]),
                        levels=np.array([0, 1000]), seed=106, weight_seed=WEIGHT_SEED)
    print('big_256x64: out absmax %g nmse %s' % (np.abs(out).max(), log[0, :, 0]))


def gen_f16w():
    """G6: BASELINE config 5, "fp16 score-net weights".  The reference cannot run ``.half()``
    (``MSFBlock`` allocates ``sums`` in fp32, layers.py:179), so the stated oracle is the fp32 reference with every
    parameter rounded to fp16 (``weights.fp16_state_dict``; the sigma buffer stays fp32)."""
    from score_based_channels_amd.weights import fp16_state_dict
    # 64x16: forward at three levels + a truncated schedule
    cfg = default_config()
    sd16 = fp16_state_dict(seeded_state_dict(cfg, WEIGHT_SEED))
    net = reference_net(cfg, sd16)
    g = np.load(os.path.join(GOLD, 'forward_64x16.npz'))
    x, levels = g['x'], [int(v) for v in g['levels']]
    with torch.no_grad():
        outs = [net(torch.from_numpy(x), torch.full((4,), lv, dtype=torch.long)).numpy() for lv in levels]
    H, P = case_inputs(107, 4, 64, 16, 0.6)
    lv = list(range(0, 2311, 77)) + [2310]
    Y, X, log = reference_ald(net, cfg, H, P, [0.0, 20.0], lv, 107)
    np.savez_compressed(os.path.join(GOLD, 'f16w_64x16.npz'), x=x, levels=np.array(levels), out=np.stack(outs), H=H,
                        P=P, Y=Y, X_final=X, nmse_log=log, snr_db=np.array([0.0, 20.0]), ald_levels=np.array(lv),
                        seed=107, weight_seed=WEIGHT_SEED)
    print('f16w_64x16: out absmax', [float(np.abs(o).max()) for o in outs], 'final nmse', log[:, -1].mean(-1))
    # 256x64: the inputs of big_256x64 with fp16-rounded parameters
    cfg = default_config(image_size=(64, 256))
    sd16 = fp16_state_dict(seeded_state_dict(cfg, WEIGHT_SEED))
    net = reference_net(cfg, sd16)
    H, P = case_inputs(21, 1, 256, 64, 0.6, profile='ULA')
    x = np.stack((H.real, H.imag), 1).astype(np.float32)
    with torch.no_grad():
        out = net(torch.from_numpy(x), torch.full((1,), 1155, dtype=torch.long)).numpy()
    Y, X, log = reference_ald(net, cfg, H, P, [10.0], [0, 1000], 106)
    np.savez_compressed(os.path.join(GOLD, 'f16w_256x64.npz'), x=x, out=out, level=1155, H=H, P=P, Y=Y, X_final=X,
                        nmse_log=log, snr_db=np.array([10.0]), levels=np.array([0, 1000]), seed=106,
                        weight_seed=WEIGHT_SEED)
    print('f16w_256x64: out absmax %g nmse %s' % (np.abs(out).max(), log[0, :, 0]))


def _reference_datasets(train_profile, test_profile, legacy_seed, num_pilots_val, spacing=0.5):
    """The data path of the reference scripts (test_score.py:62-113 / tune_hparams_score.py:48-97), executed with the
    reference's own ``loaders.Channels`` and ``DataLoader``: training-profile dataset for the normalisation constants,
    then a factory for validation datasets (the tuner re-creates one per grid cell, drawing fresh pilots from numpy's
    global RNG).  The ``.mat`` reader is replaced by the build's synthetic generator -- the same arrays
    ``Channels(..., synthetic=True)`` of this repo produces for those file names."""
    import re
    from torch.utils.data import DataLoader
    fake = types.ModuleType('hdf5storage')

    def loadmat(fn):
        m = re.match(r'(.+)_Nt(\d+)_Nr(\d+)_ULA([0-9.]+)_seed(\d+)\.mat', os.path.basename(fn))
        prof, nt, nr, sp, seed = m.group(1), int(m.group(2)), int(m.group(3)), float(m.group(4)), int(m.group(5))
        return {'output_h': synth.generate_output_h(prof, 200, nt, nr, sp, seed, n_sym=1)}
    fake.loadmat = loadmat
    sys.modules['hdf5storage'] = fake
    from score_based_channels.loaders import Channels
    import copy
    cfg = default_config(train_profile)
    np.random.seed(legacy_seed % (2 ** 32))
    cfg.data.channel = train_profile
    dataset = Channels(1234, cfg, norm=cfg.data.norm_channels)
    def make_val():
        val_config = copy.deepcopy(cfg)
        val_config.data.channel = test_profile
        val_config.data.spacing_list = [spacing]
        val_config.data.num_pilots = num_pilots_val
        return Channels(4321, val_config, norm=[dataset.mean, dataset.std])
    # created lazily: numpy's global RNG is consumed in script order (dataset -> its batch -> next dataset)
    return cfg, dataset, make_val, DataLoader


def _first_batch(val_dataset, DataLoader, n):
    """test_score.py:102-113: first batch, Hermitian pilots, complex normalised Hermitian channels."""
    sample = next(iter(DataLoader(val_dataset, batch_size=n, shuffle=False, num_workers=0, drop_last=True)))
    val_P = torch.conj(torch.transpose(sample['P'], -1, -2))
    val_H = sample['H_herm'][:, 0] + 1j * sample['H_herm'][:, 1]
    return val_H.numpy().astype(np.complex64), val_P.resolve_conj().numpy().astype(np.complex64)


def gen_cross():
    """BASELINE config 4 end to end: ``--train CDL-C --test CDL-D`` (normalisation constants from the TRAIN profile's
    seed-1234 dataset, test_score.py:68-69,101), 4 channels x 17 SNR points x the first 2 noise levels, seed 3.
    The build's CLI must reproduce ``nmse_log`` from the same command line with ``--noise host``."""
    seed, B = 3, 4
    cfg, dataset, make_val, DataLoader = _reference_datasets('CDL-C', 'CDL-D', seed, 38)
    H, P = _first_batch(make_val(), DataLoader, B)
    net = reference_net(cfg, seeded_state_dict(cfg, WEIGHT_SEED))
    snr = np.arange(-10, 32.5, 2.5)
    Y, X, log = reference_ald(net, cfg, H, P, snr, [0, 1], seed)
    np.savez_compressed(os.path.join(GOLD, 'cli_cross_cdlc_cdld.npz'), H=H, P=P, nmse_log=log, X_final=X, snr_db=snr,
                        train_std=np.float64(dataset.std), seed=seed, weight_seed=WEIGHT_SEED,
                        argv=np.array('--train CDL-C --test CDL-D --synthetic --synthetic_weights 2024 --num_levels 2 '
                                      '--num_channels 4 --seed 3 --noise host --no_plot'))
    print('cli_cross: std(train)=%g |H| rms %g nmse[0,-1]=%s' % (dataset.std, np.sqrt(np.mean(np.abs(H) ** 2)), log[0, -1]))


def gen_tunecli():
    """BASELINE config 3 end to end at reduced size: a 2 x 2 (alpha, beta) grid, 3 channels, first noise level; every cell
    re-creates the validation dataset and draws its own noise streams (tune_hparams_score.py:71-97)."""
    seed, B = 4, 3
    alphas, betas = [3e-11, 3e-10], [0.1, 0.01]
    cfg, dataset, make_val, DataLoader = _reference_datasets('CDL-C', 'CDL-C', seed, 38)
    net = reference_net(cfg, seeded_state_dict(cfg, WEIGHT_SEED))
    snr = np.arange(-10, 32.5, 2.5)
    logs = np.zeros((2, 2, len(snr), 3, B), np.float32)
    for meta_idx, (a, b) in enumerate([(a, b) for a in alphas for b in betas]):
        # the reference batches the whole validation set (batch_size=len(val_dataset), tune_hparams_score.py:84-85) and
        # its log has room for exactly 100 channels (:63); here the first B items play that role
        H, P = _first_batch(make_val(), DataLoader, B)
        # HostNoise(seed, combo=meta_idx): reference_ald keys its streams by seed only, so fold the cell into the call
        _, _, log = reference_ald(net, cfg, H, P, snr, [0], seed, alpha_step=a, beta_noise=b, combo=meta_idx)
        logs[meta_idx // 2, meta_idx % 2] = log
    np.savez_compressed(os.path.join(GOLD, 'cli_tune_grid.npz'), nmse_log=logs, snr_db=snr, alpha_step_range=alphas,
                        beta_noise_range=betas, seed=seed, weight_seed=WEIGHT_SEED)
    print('cli_tune: nmse[...,0,-1,0] =', logs[..., 0, -1, 0])


def reference_mmse(net, config, H, P, snr_db, levels, seed, best_step, best_noise, best_stop, mmse_avg, dc_boost,
                   start_point, steps_each=3):
    """test_mmse.py:166-277 around the imported network, with keyed noise.  Two stale lines of the script are written as
    they are meant: the per-SNR collections are re-created for every SNR point (the script appends to tensors from the
    second point on, :185-193), and the NMSE vector is reshaped to (kept_samples, mmse_avg) directly (the script's
    ``.view(len(snr_range), -1)`` only works when 19 divides the batch, :238-244)."""
    kept = H.shape[0]
    noise_range = 10 ** (-np.asarray(snr_db, np.float64) / 10.)          # no Nt factor here (:100)
    val_P, val_H = torch.from_numpy(P), torch.from_numpy(H)
    total_steps = len(levels) * steps_each
    oracle_log = np.zeros((len(noise_range), total_steps, kept, mmse_avg))
    saved_H = np.zeros((len(noise_range), kept, mmse_avg) + H.shape[1:], np.complex64)
    Ys = []
    for snr_idx, local_noise in enumerate(noise_range):
        noise = HostNoise(seed, combo=1 + snr_idx)
        step_size = 1. * best_step[snr_idx]                              # fixed_step_size * best_step (:171)
        noise_boost, target_stop = best_noise[snr_idx], best_stop[snr_idx]
        local_Y = torch.matmul(val_P, val_H)
        local_Y = local_Y + np.sqrt(local_noise) * torch.from_numpy(noise.measurement(0, tuple(local_Y.shape)))
        global_Y = torch.cat([torch.tile(local_Y[i][None, ...].clone(), (mmse_avg, 1, 1)) for i in range(kept)])
        global_P = torch.cat([torch.tile(val_P[i][None, ...].clone(), (mmse_avg, 1, 1)) for i in range(kept)])
        global_H = torch.cat([torch.tile(val_H[i][None, ...].clone(), (mmse_avg, 1, 1)) for i in range(kept)])
        if start_point == 'Noise':
            current = torch.from_numpy(noise.init(tuple(global_H.shape)))
        elif start_point == 'Adjoint':
            current = torch.matmul(torch.conj(torch.transpose(global_P, -1, -2)), global_Y)
        elif start_point == 'LS':
            # test_mmse.py:200-202 as it is meant (the script calls .cuda() on lstsq's result tuple): the minimum-norm
            # least-squares solution of P x = Y per chain, LAPACK gelsd
            current = torch.linalg.lstsq(global_P, global_Y, driver='gelsd').solution
        y, forward, forward_h, oracle = global_Y, global_P, torch.conj(torch.transpose(global_P, -1, -2)), global_H
        draw = noise.step_stream(0, tuple(global_H.shape))
        trailing_idx, mark_break = 0, False
        with torch.no_grad():
            for step_idx in levels:
                current_sigma = net.sigmas[step_idx].item()
                labels = (torch.ones(global_H.shape[0]) * step_idx).long()
                for inner_idx in range(steps_each):
                    current_real = torch.view_as_real(current).permute(0, 3, 1, 2)
                    score = net(current_real, labels)
                    score = torch.view_as_complex(score.permute(0, 2, 3, 1).contiguous())
                    alpha = step_size * (current_sigma / config.model.sigma_end) ** 2
                    meas_term = torch.matmul(forward, current) - y
                    meas_grad = torch.matmul(forward_h, meas_term)
                    grad_noise = np.sqrt(2 * alpha * noise_boost) * torch.from_numpy(draw(trailing_idx))
                    current = current + alpha * (score - dc_boost * meas_grad /
                                                 (local_noise / 2. + current_sigma ** 2)) + grad_noise
                    oracle_log[snr_idx, trailing_idx] = np.reshape(
                        (torch.sum(torch.square(torch.abs(current - oracle)), dim=(-1, -2)) /
                         torch.sum(torch.square(torch.abs(oracle)), dim=(-1, -2))).numpy(), (kept, mmse_avg))
                    if trailing_idx == target_stop:
                        mark_break = True
                        break
                    trailing_idx = trailing_idx + 1
                if mark_break:
                    saved_H[snr_idx] = torch.reshape(current, (kept, mmse_avg) + H.shape[1:]).numpy()
                    break
        Ys.append(local_Y.numpy())
    return np.stack(Ys), oracle_log, saved_H


def gen_mmse():
    """F2: posterior-mean sampling (test_mmse.py): 2 kept samples x 3 chains, two SNR points with their own (step, noise,
    stop) triple, dc_boost = 2, start points Noise and Adjoint."""
    cfg = default_config()
    net = reference_net(cfg, seeded_state_dict(cfg, WEIGHT_SEED))
    H, P = case_inputs(108, 2, 64, 16, 0.6)
    snr, levels = [-10.0, 10.0], [0, 1000, 2000]
    best_step, best_noise, best_stop = [3e-11, 6e-11], [0.01, 0.1], [3, 4]
    out = {}
    for sp in ('Noise', 'Adjoint'):
        Y, log, saved = reference_mmse(net, cfg, H, P, snr, levels, 108, best_step, best_noise, best_stop, 3, 2.0, sp)
        out.update({'Y': Y, 'oracle_log_' + sp: log, 'saved_H_' + sp: saved})
        print('mmse %s: log at stop' % sp, log[0, 3].ravel()[:3], log[1, 4].ravel()[:3])
    np.savez_compressed(os.path.join(GOLD, 'mmse.npz'), H=H, P=P, snr_db=np.array(snr), levels=np.array(levels), seed=108,
                        best_step=np.array(best_step), best_noise=np.array(best_noise), best_stop=np.array(best_stop),
                        mmse_avg=3, dc_boost=2.0, weight_seed=WEIGHT_SEED, **out)


def gen_mmse_ls():
    """F2, the third start point (test_mmse.py:200-202, ``--start_point LS``): same case as ``gen_mmse``, own file so that the
    pinned ``mmse.npz`` stays byte for byte what it was."""
    cfg = default_config()
    net = reference_net(cfg, seeded_state_dict(cfg, WEIGHT_SEED))
    H, P = case_inputs(108, 2, 64, 16, 0.6)
    snr, levels = [-10.0, 10.0], [0, 1000, 2000]
    best_step, best_noise, best_stop = [3e-11, 6e-11], [0.01, 0.1], [3, 4]
    Y, log, saved = reference_mmse(net, cfg, H, P, snr, levels, 108, best_step, best_noise, best_stop, 3, 2.0, 'LS')
    print('mmse LS: log at stop', log[0, 3].ravel()[:3], log[1, 4].ravel()[:3])
    np.savez_compressed(os.path.join(GOLD, 'mmse_ls.npz'), H=H, P=P, snr_db=np.array(snr), levels=np.array(levels), seed=108,
                        best_step=np.array(best_step), best_noise=np.array(best_noise), best_stop=np.array(best_stop),
                        mmse_avg=3, dc_boost=2.0, weight_seed=WEIGHT_SEED, Y=Y, oracle_log_LS=log, saved_H_LS=saved)


def gen_cross_b():
    """BASELINE config 4, a second foreign profile: ``--train CDL-C --test CDL-B`` (as ``gen_cross``; seed 6, 3 noise levels)."""
    seed, B = 6, 4
    cfg, dataset, make_val, DataLoader = _reference_datasets('CDL-C', 'CDL-B', seed, 38)
    H, P = _first_batch(make_val(), DataLoader, B)
    net = reference_net(cfg, seeded_state_dict(cfg, WEIGHT_SEED))
    snr = np.arange(-10, 32.5, 2.5)
    Y, X, log = reference_ald(net, cfg, H, P, snr, [0, 1, 2], seed)
    np.savez_compressed(os.path.join(GOLD, 'cli_cross_cdlc_cdlb.npz'), H=H, P=P, nmse_log=log, X_final=X, snr_db=snr,
                        train_std=np.float64(dataset.std), seed=seed, weight_seed=WEIGHT_SEED,
                        argv=np.array('--train CDL-C --test CDL-B --synthetic --synthetic_weights 2024 --num_levels 3 '
                                      '--num_channels 4 --seed 6 --noise host --no_plot'))
    print('cli_cross_b: std(train)=%g |H| rms %g nmse[0,-1]=%s' % (dataset.std, np.sqrt(np.mean(np.abs(H) ** 2)), log[0, -1]))


def gen_cross_a():
    """BASELINE config 4, the third foreign profile: ``--train CDL-C --test CDL-A`` (as ``gen_cross``; seed 9, 2 noise levels, and
    SIX channels x 17 SNR points = 102 trajectories, which the GPU test also runs in chunks of 40: chunk boundaries inside the
    reference's result)."""
    seed, B = 9, 6
    cfg, dataset, make_val, DataLoader = _reference_datasets('CDL-C', 'CDL-A', seed, 38)
    H, P = _first_batch(make_val(), DataLoader, B)
    net = reference_net(cfg, seeded_state_dict(cfg, WEIGHT_SEED))
    snr = np.arange(-10, 32.5, 2.5)
    Y, X, log = reference_ald(net, cfg, H, P, snr, [0, 1], seed)
    np.savez_compressed(os.path.join(GOLD, 'cli_cross_cdlc_cdla.npz'), H=H, P=P, nmse_log=log, X_final=X, snr_db=snr,
                        train_std=np.float64(dataset.std), seed=seed, weight_seed=WEIGHT_SEED,
                        argv=np.array('--train CDL-C --test CDL-A --synthetic --synthetic_weights 2024 --num_levels 2 '
                                      '--num_channels 6 --seed 9 --noise host --no_plot'))
    print('cli_cross_a: std(train)=%g |H| rms %g nmse[0,-1]=%s' % (dataset.std, np.sqrt(np.mean(np.abs(H) ** 2)), log[0, -1]))


def gen_tunecli2():
    """BASELINE config 3 end to end over several noise levels: the 2 x 2 grid of ``gen_tunecli`` walked over the first TWO
    levels x 3 steps (the per-level scalars alpha / noise scale / dc divisor change between levels)."""
    seed, B = 9, 3
    alphas, betas = [3e-11, 1e-10], [0.01, 0.001]
    cfg, dataset, make_val, DataLoader = _reference_datasets('CDL-C', 'CDL-C', seed, 38)
    net = reference_net(cfg, seeded_state_dict(cfg, WEIGHT_SEED))
    snr = np.arange(-10, 32.5, 2.5)
    logs = np.zeros((2, 2, len(snr), 6, B), np.float32)
    for meta_idx, (a, b) in enumerate([(a, b) for a in alphas for b in betas]):
        H, P = _first_batch(make_val(), DataLoader, B)
        _, _, log = reference_ald(net, cfg, H, P, snr, [0, 1], seed, alpha_step=a, beta_noise=b, combo=meta_idx)
        logs[meta_idx // 2, meta_idx % 2] = log
    np.savez_compressed(os.path.join(GOLD, 'cli_tune_grid_2levels.npz'), nmse_log=logs, snr_db=snr, alpha_step_range=alphas,
                        beta_noise_range=betas, seed=seed, weight_seed=WEIGHT_SEED, num_levels=2)
    print('cli_tune2: nmse[...,0,-1,0] =', logs[..., 0, -1, 0])


def gen_loader():
    out_h = synth.generate_output_h('CDL-C', 12, 64, 16, 0.5, 4321, n_sym=2)
    fake = types.ModuleType('hdf5storage')
    fake.loadmat = lambda fn: {'output_h': out_h}
    sys.modules['hdf5storage'] = fake
    from score_based_channels.loaders import Channels
    cfg = default_config()
    cfg.data.num_pilots = 38
    np.random.seed(999)
    ds = Channels(4321, cfg, norm='global')
    items = [ds[i] for i in (0, 5)]
    np.savez_compressed(os.path.join(GOLD, 'loader.npz'), output_h=out_h, legacy_seed=999,
                        mean=np.float64(ds.mean), std=np.float64(ds.std),
                        pilots=ds.pilots.astype(np.complex64),
                        H0=items[0]['H'], H_herm0=items[0]['H_herm'], P0=items[0]['P'],
                        H5=items[1]['H'], H_herm5=items[1]['H_herm'], P5=items[1]['P'],
                        filename=ds.filenames[0])
    ds2 = Channels(4321, cfg, norm=[0.25, 2.0])
    np.savez_compressed(os.path.join(GOLD, 'loader_listnorm.npz'), H_herm3=ds2[3]['H_herm'],
                        mean=0.25, std=2.0)
    print('loader: std %g file %s' % (ds.std, ds.filenames[0]))


def gen_tune():
    """tune_hparams_score.py:151-162 on a small synthetic log, run with the same numpy calls."""
    rng = np.random.default_rng(8)
    alpha_range, beta_range = np.asarray([3e-11, 6e-11, 1e-10]), np.asarray([0.1, 0.01])
    nmse_log = rng.random((3, 2, 5, 12, 7)) ** 2
    avg_nmse = np.mean(nmse_log, axis=-1)
    best_nmse = np.min(avg_nmse, axis=-1)
    ba, bb = [], []
    for snr_idx in range(5):
        local_nmse = best_nmse[..., snr_idx].flatten()
        best_idx = np.argmin(local_nmse)
        ai, bi = np.unravel_index(best_idx, (len(alpha_range), len(beta_range)))
        ba.append(alpha_range[ai])
        bb.append(beta_range[bi])
    np.savez_compressed(os.path.join(GOLD, 'tune_post.npz'), nmse_log=nmse_log, avg_nmse=avg_nmse,
                        best_nmse=best_nmse, best_alpha_snr=np.array(ba), best_beta_snr=np.array(bb),
                        alpha_step_range=alpha_range, beta_noise_range=beta_range)


def _train_batch(seed, B, config):
    """Normalised Hermitian channels as the real view train_score.py:151-153 feeds the loss, labels and N(0,1) draws."""
    H, _ = case_inputs(seed, B, 64, 16, 0.6)
    x = np.stack([H.real, H.imag], axis=1).astype(np.float32)            # [B, 2, Nt, Nr]
    rng = np.random.default_rng([seed, 5])
    labels = rng.integers(0, config.model.num_classes, size=B).astype(np.int64)
    z = rng.standard_normal(x.shape).astype(np.float32)
    return x, labels, z


def _reference_dsm(net, sigmas, x, labels, z):
    """ncsnv2/losses/dsm.py:6-32 called as train_score.py:151-153 does, with ``torch.randn_like`` returning ``z``."""
    from ncsnv2.losses import dsm
    real = torch.randn_like
    torch.randn_like = lambda t: torch.from_numpy(z)
    try:
        return dsm.anneal_dsm_score_estimation(net, torch.from_numpy(x), sigmas, torch.from_numpy(labels), 2.)
    finally:
        torch.randn_like = real


def gen_train():
    """SURVEY 8(f) F4.  (1) loss and every parameter gradient of one DSM batch through the reference network (autograd);
    (2) three optimiser steps of train_score.py:145-173 (Adam lr 1e-4, betas (0.9, 0.999), eps 1e-3; EMA 0.999) on three
    different batches.  5.9 M values per tensor set do not fit a fixture: gradients / updates are stored as per-tensor
    digests (conftest.tensor_digest), small tensors in full."""
    from conftest import tensor_digest
    from ncsnv2.models import get_sigmas
    from ncsnv2.models.ema import EMAHelper
    cfg = default_config()
    sd = seeded_state_dict(cfg, WEIGHT_SEED)
    net = reference_net(cfg, sd).train()
    sigmas = get_sigmas(net.config)
    B = 4
    x, labels, z = _train_batch(31, B, cfg)
    labels[:2] = [0, cfg.model.num_classes - 1]                          # both ends of the noise schedule
    t0 = time.time()
    loss = _reference_dsm(net, sigmas, x, labels, z)
    loss.backward()
    out = dict(x=x, labels=labels, z=z, loss=np.float64(loss.item()), weight_seed=WEIGHT_SEED)
    names = []
    for name, p in net.named_parameters():
        g = p.grad.numpy()
        names.append(name)
        out['gd_' + name] = tensor_digest(name, g)
        if g.size <= 4096:
            out['g_' + name] = g.copy()
    # per-sample losses (the mean of :32 hides which sample is off)
    with torch.no_grad():
        used = sigmas[torch.from_numpy(labels)].view(B, 1, 1, 1)
        noise = torch.from_numpy(z) * used
        scores = net(torch.from_numpy(x) + noise, torch.from_numpy(labels))
        per = 0.5 * ((scores + noise / used ** 2).reshape(B, -1) ** 2).sum(-1) * used.squeeze() ** 2
    out['loss_per_sample'] = per.numpy()
    print('train gradients: loss %.6g, %d tensors, %.1f s' % (loss.item(), len(names), time.time() - t0))

    # (2) the optimiser loop
    net = reference_net(cfg, sd).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), amsgrad=False, eps=1e-3)
    ema = EMAHelper(mu=0.999)
    ema.register(net)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    K = 3
    xs, ls, zs, losses = [], [], [], []
    for k in range(K):
        xk, lk, zk = _train_batch(40 + k, B, cfg)
        lo = _reference_dsm(net, sigmas, xk, lk, zk)
        opt.zero_grad()
        lo.backward()
        opt.step()
        ema.update(net)
        xs.append(xk); ls.append(lk); zs.append(zk); losses.append(lo.item())
    out.update(steps_x=np.stack(xs), steps_labels=np.stack(ls), steps_z=np.stack(zs), steps_loss=np.array(losses))
    for name, p in net.named_parameters():
        out['ud_' + name] = tensor_digest(name, (p.detach() - before[name]).numpy())
        out['ed_' + name] = tensor_digest(name, (ema.shadow[name] - before[name]).numpy())
    # validation-style loss of the EMA copy on the first batch (train_score.py:172-185)
    val = ema.ema_copy(net)
    with torch.no_grad():
        out['ema_loss'] = np.float64(_reference_dsm(val, sigmas, x, labels, z).item())
    print('train loop: losses', losses, 'ema loss', out['ema_loss'])
    np.savez_compressed(os.path.join(GOLD, 'train_dsm.npz'), **out)


def gen_trained_long():
    """Round 6: the same goldens on the 4000-step checkpoint (tests/trained_weights.py: LONG_STEPS; gpurun_out/r6f/trained_model_state_4000.npz)."""
    gen_trained(steps=4000, path=os.environ.get('SBC_TRAINED_NPZ', os.path.join(ROOT, 'gpurun_out', 'r6f', 'trained_model_state_4000.npz')))


def gen_trained(steps=300, path=None):
    """VERDICT r4 missing item 5: goldens on weights that have been TRAINED.  The checkpoint is the one ``tests/trained_weights.py``
    makes on the GPU (300 optimiser steps of this package's trainer; bit-reproducible, so the GPU test re-creates it instead of
    shipping 24 MB); its ``model_state`` comes home as ``gpurun_out/r5b/trained_model_state.npz`` (or $SBC_TRAINED_NPZ).  The
    REFERENCE network then runs on it: forward at three levels (the inputs of forward_64x16.npz) and the 93-step truncated
    schedule of ald_trunc.npz (8 channels x 3 SNR points)."""
    import trained_weights as TW
    path = path or os.environ.get('SBC_TRAINED_NPZ', os.path.join(ROOT, 'gpurun_out', 'r5b', 'trained_model_state.npz'))
    with np.load(path) as f:
        sd = {k: f[k] for k in f.files}
    cfg = default_config()
    net = reference_net(cfg, sd)
    g = np.load(os.path.join(GOLD, 'forward_64x16.npz'))
    x, levels = g['x'], [int(v) for v in g['levels']]
    with torch.no_grad():
        outs = [net(torch.from_numpy(x), torch.full((4,), lv, dtype=torch.long)).numpy() for lv in levels]
    H, P = case_inputs(109, 8, 64, 16, 0.6)
    lv = list(range(0, 2311, 77)) + [2310]
    snr = [-10.0, 10.0, 30.0]
    t = time.time()
    Y, X, log = reference_ald(net, cfg, H, P, snr, lv, 109)
    keys, dig, crc = TW.state_digest(sd)
    init = seeded_state_dict(cfg, 1)                     # what training started from (train_score.fresh_state_dict: betas zeroed)
    moved = max(float(np.max(np.abs(sd[k] - init[k]))) for k in sd if k.endswith('conv.weight') or k.endswith('_conv.weight'))
    np.savez_compressed(os.path.join(GOLD, 'trained_%dsteps.npz' % steps), x=x, levels=np.array(levels), out=np.stack(outs), H=H, P=P, Y=Y,
                        X_final=X, nmse_log=log, snr_db=np.array(snr), ald_levels=np.array(lv), seed=109,
                        weight_keys=np.array(keys), weight_digest=dig, weight_crc=crc)
    print('trained_%dsteps: out absmax %s, final nmse %s, largest weight move %.3g, ALD in %.0f s'
          % (steps, [float(np.abs(o).max()) for o in outs], log[:, -1].mean(-1), moved, time.time() - t))


if __name__ == '__main__':
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    todo = sys.argv[1:] or ['forward', 'plumbing', 'trunc', 'big', 'loader', 'tune']
    for name in todo:
        globals()['gen_' + name]()
