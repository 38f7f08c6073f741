"""Pin the CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/gen_golden.py).  CPU only; no HIP code is exercised here."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, rel_err
from oracle import ald_oracle, ncsnv2_oracle
from score_based_channels_amd.noise import HostNoise
from score_based_channels_amd.weights import state_dict_spec


def test_state_dict_grammar_matches_reference():
    with open(os.path.join(GOLDEN, 'state_dict_keys.json')) as f:
        ref = [(k, tuple(s)) for k, s in json.load(f)]
    assert [(k, tuple(s)) for k, s in state_dict_spec()] == ref


def test_conv_flops_match_survey():
    assert ncsnv2_oracle.conv_flops_per_sample(32, 64, 16) == 820772864
    assert ncsnv2_oracle.conv_flops_per_sample(32, 256, 64) == 13132365824


def test_forward_matches_reference(weights64):
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    for i, lv in enumerate(g['levels']):
        out = ncsnv2_oracle.score_forward(sd, g['x'], np.full((4,), lv))
        assert rel_err(out, g['out'][i]) < 2e-5, lv


def test_forward_stages_match_reference(weights64):
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    _, st = ncsnv2_oracle.score_forward(sd, g['x'][:1], np.array([1155]), return_stages=True)
    names = {'begin_conv': 'begin'}
    for k in [k for k in g if k.startswith('stage_')]:
        name = k[len('stage_'):]
        assert rel_err(st[names.get(name, name)], g[k]) < 2e-5, name


def _oracle_ald(sd, cfg, g, snr_idx):
    noise = HostNoise(int(g['seed']))
    H, P = g['H'], g['P']
    local_noise = ald_oracle.snr_to_noise(g['snr_db'], H.shape[1])[snr_idx]
    Y = ald_oracle.make_measurements(P, H, local_noise, noise.measurement(snr_idx, g['Y'][snr_idx].shape))
    X, log = ald_oracle.ald_run(
        lambda x, lab: ncsnv2_oracle.score_forward(sd, x, lab), sd['sigmas'], cfg.model.sigma_end,
        P, Y, H, noise.init(H.shape), noise.step_stream(snr_idx, H.shape), local_noise,
        alpha_step=float(g['alpha_step']), beta_noise=float(g['beta_noise']),
        steps_each=int(g['steps_each']), levels=[int(v) for v in g['levels']])
    return Y, X, log


@pytest.mark.parametrize('name', ['ald_plumbing_level0.npz', 'ald_plumbing_3levels.npz'])
def test_ald_plumbing_matches_reference(weights64, name):
    cfg, sd = weights64
    g = load_golden(name)
    Y, X, log = _oracle_ald(sd, cfg, g, 0)
    assert rel_err(Y, g['Y'][0]) < 1e-6
    assert np.max(np.abs(log / g['nmse_log'][0] - 1)) < 1e-5
    assert rel_err(X, g['X_final'][0]) < 1e-5


def test_ald_truncated_schedule_matches_reference(weights64):
    """Every 77th noise level + the last one (31 levels x 3 steps), non-default (alpha, beta) cell."""
    cfg, sd = weights64
    g = load_golden('ald_trunc_cell.npz')
    _, X, log = _oracle_ald(sd, cfg, g, 0)
    assert np.max(np.abs(log / g['nmse_log'][0] - 1)) < 1e-5
    assert rel_err(X, g['X_final'][0]) < 1e-5


def test_big_array_forward_matches_reference():
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.weights import seeded_state_dict
    cfg = default_config(image_size=(64, 256))
    sd = seeded_state_dict(cfg, 2024)
    g = load_golden('big_256x64.npz')
    out = ncsnv2_oracle.score_forward(sd, g['x'], np.array([int(g['level'])]))
    assert rel_err(out, g['out']) < 2e-5


def test_fp16_weight_goldens_match_oracle_with_rounded_parameters(weights64):
    """G6 / BASELINE config 5: the reference with fp16-rounded parameters == the oracle fed ``fp16_state_dict`` weights
    (fp32 arithmetic on both sides, so the fp32 tolerance applies); this is what conv_mode f16w is held against."""
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.weights import fp16_state_dict, seeded_state_dict
    _, sd = weights64
    sd16 = fp16_state_dict(sd)
    g = load_golden('f16w_64x16.npz')
    for i, lv in enumerate(g['levels']):
        out = ncsnv2_oracle.score_forward(sd16, g['x'], np.full((4,), lv))
        assert rel_err(out, g['out'][i]) < 2e-5, lv
        assert rel_err(out, load_golden('forward_64x16.npz')['out'][i]) > 1e-4      # and it is a different network
    cfg = default_config(image_size=(64, 256))
    gb = load_golden('f16w_256x64.npz')
    out = ncsnv2_oracle.score_forward(fp16_state_dict(seeded_state_dict(cfg, 2024)), gb['x'], np.array([int(gb['level'])]))
    assert rel_err(out, gb['out']) < 2e-5


@pytest.mark.parametrize('start', ['Noise', 'Adjoint', 'LS'])
def test_posterior_mean_loop_matches_reference(weights64, start):
    """F2: ``ald_oracle.mmse_run`` against the transcription of test_mmse.py:166-277 run around the reference network
    (2 samples x 3 chains sharing a measurement, per-SNR (step, noise, stop), dc_boost 2, both start points)."""
    cfg, sd = weights64
    g = load_golden('mmse_ls.npz' if start == 'LS' else 'mmse.npz')        # LS (test_mmse.py:200-202) has its own file
    H, P, navg = g['H'], g['P'], int(g['mmse_avg'])
    levels = [int(v) for v in g['levels']]
    for s, snr in enumerate(g['snr_db']):
        noise = HostNoise(int(g['seed']), combo=1 + s)
        local_noise = 10 ** (-snr / 10.)                                   # test_mmse.py:100, no Nt factor
        Y = ald_oracle.make_measurements(P, H, local_noise, noise.measurement(0, g['Y'][s].shape))
        assert rel_err(Y, g['Y'][s]) < 1e-6
        if start == 'Noise':
            init = noise.init((H.shape[0] * navg,) + H.shape[1:])
        elif start == 'LS':                                                # minimum-norm least squares per sample
            init = np.repeat(np.stack([np.linalg.lstsq(P[b], Y[b], rcond=None)[0] for b in range(H.shape[0])]).astype(np.complex64),
                             navg, axis=0)
        else:
            init = np.repeat(np.matmul(np.conj(np.transpose(P, (0, 2, 1))), Y), navg, axis=0)
        stop = int(g['best_stop'][s])
        log, est = ald_oracle.mmse_run(lambda x, lab: ncsnv2_oracle.score_forward(sd, x, lab), sd['sigmas'],
                                       cfg.model.sigma_end, P, Y, H, init,
                                       noise.step_stream(0, (H.shape[0] * navg,) + H.shape[1:]), local_noise,
                                       float(g['best_step'][s]), float(g['best_noise'][s]), stop, navg,
                                       dc_boost=float(g['dc_boost']), levels=levels)
        ref = g['oracle_log_' + start][s]
        assert np.all(log[stop + 1:] == 0) and np.all(ref[stop + 1:] == 0)
        assert np.max(np.abs(log[:stop + 1] / ref[:stop + 1] - 1)) < 1e-5
        assert rel_err(est, g['saved_H_' + start][s]) < 1e-5


def test_loader_pieces_match_reference():
    g = load_golden('loader.npz')
    ch, mean, std, pil = ald_oracle.channels_dataset(g['output_h'], 64, 38, 'global',
                                                     legacy_seed=int(g['legacy_seed']))
    assert mean == 0. and abs(std - float(g['std'])) < 1e-7
    assert np.array_equal(pil.astype(np.complex64), g['pilots'])
    for idx in (0, 5):
        it = ald_oracle.channels_item(ch, mean, std, pil, idx)
        assert np.array_equal(it['H_herm'], g['H_herm%d' % idx])
        assert np.array_equal(it['H'], g['H%d' % idx])
        assert np.array_equal(it['P'], g['P%d' % idx])


def test_tune_postprocessing_matches_reference():
    g = load_golden('tune_post.npz')
    avg, best = ald_oracle.reduce_nmse(g['nmse_log'])
    assert np.array_equal(avg, g['avg_nmse']) and np.array_equal(best, g['best_nmse'])
    ba, bb = ald_oracle.tune_select(best, g['alpha_step_range'], g['beta_noise_range'])
    assert np.array_equal(ba, g['best_alpha_snr']) and np.array_equal(bb, g['best_beta_snr'])


def test_dsm_loss_restatement_matches_reference(weights64):
    """oracle/dsm_oracle.py (ncsnv2/losses/dsm.py:6-32) around the numpy score network against the loss the reference's own
    ``anneal_dsm_score_estimation`` returned for the same samples, labels and noise (tests/golden/train_dsm.npz)."""
    from oracle import dsm_oracle as D
    g = load_golden('train_dsm.npz')
    _, sd = weights64
    pert, noise, used = D.perturb(g['x'], sd['sigmas'], g['labels'], g['z'])
    scores = ncsnv2_oracle.score_forward(sd, pert, g["labels"])
    per = D.loss_per_sample(scores, noise, used)
    assert np.max(np.abs(per / g['loss_per_sample'] - 1)) < 2e-5
    assert abs(per.astype(np.float64).mean() / g['loss'] - 1) < 2e-5


def test_adam_ema_restatement_matches_torch():
    """oracle adam_ema_step against torch.optim.Adam + the EMAHelper update rule (models/ema.py:17-22) on CPU."""
    import torch
    from oracle import dsm_oracle as D
    rng = np.random.default_rng(3)
    p0 = rng.standard_normal(500).astype(np.float32)
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), amsgrad=False, eps=1e-3)
    shadow_t = pt.data.clone()
    p, m, v, sh = p0, np.zeros_like(p0), np.zeros_like(p0), p0.copy()
    for t in range(1, 4):
        gk = (rng.standard_normal(500) * 10.0 ** rng.uniform(-4, 1, 500)).astype(np.float32)
        pt.grad = torch.from_numpy(gk.copy())
        opt.step()
        shadow_t = (1. - 0.999) * pt.data + 0.999 * shadow_t
        p, m, v, sh = D.adam_ema_step(p, gk, m, v, sh, t)
    assert np.max(np.abs(p - pt.data.numpy())) < 5e-7 and np.max(np.abs(sh - shadow_t.numpy())) < 5e-7


# Random123 (D. E. Shaw Research) known-answer vectors for philox4x32-10, examples/kat_vectors: counter, key -> output
PHILOX_KAT = [((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
               (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
              ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
               (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
              ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
               (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox4x32_known_answers():
    """The host restatement of the in-kernel generator reproduces the published Random123 vectors (the same vectors are put
    through the device code by tests/test_gpu_parity.py::test_device_philox_known_answers)."""
    for ctr, key, out in PHILOX_KAT:
        got = ald_oracle.philox4x32(np.array(ctr, np.uint32), np.array(key, np.uint32))
        assert [int(v) for v in got] == list(out), (ctr, [hex(int(v)) for v in got])
    # vectorised call = element-wise calls
    ctrs = np.array([k[0] for k in PHILOX_KAT], np.uint32)
    keys = np.array([k[1] for k in PHILOX_KAT], np.uint32)
    assert np.array_equal(ald_oracle.philox4x32(ctrs, keys), np.array([k[2] for k in PHILOX_KAT], np.uint32))


def test_device_noise_restatement_is_standard_complex_normal():
    """``device_complex_normal``: CN(0,1) moments at N = 2^21 draws (tolerances = 5 standard errors), distinct streams per
    (seed, trajectory, step), and a prefix property (element e does not depend on how many elements are drawn)."""
    n = 1 << 21
    z = ald_oracle.device_complex_normal(1234, 7, 3, n)
    assert z.dtype == np.complex64 and z.shape == (n,)
    se = 1.0 / np.sqrt(n)
    assert abs(z.real.mean()) < 5 * se * np.sqrt(0.5) and abs(z.imag.mean()) < 5 * se * np.sqrt(0.5)
    assert abs(z.real.var() - 0.5) < 5 * se * 0.5 * np.sqrt(2) and abs(z.imag.var() - 0.5) < 5 * se * 0.5 * np.sqrt(2)
    assert abs(np.mean(z.real * z.imag)) < 5 * se * 0.5
    assert abs(np.mean(np.abs(z) ** 4) - 2.0) < 5 * se * np.sqrt(20.0)          # E|z|^4 = 2, Var|z|^4 = 20 for CN(0,1)
    assert abs(np.mean(z[:-1] * np.conj(z[1:]))) < 5 * se                      # neighbours (the two halves of a block) uncorrelated
    assert np.array_equal(z[:1001], ald_oracle.device_complex_normal(1234, 7, 3, 1001))
    for other in ((1235, 7, 3), (1234, 8, 3), (1234, 7, 4), (1234, 7 + 2 ** 32, 3)):
        assert not np.array_equal(z[:64], ald_oracle.device_complex_normal(*other, 64))
