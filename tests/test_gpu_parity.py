"""GPU parity of the whole hot path (score network, Langevin loop) against the golden fixtures produced by the
reference (tests/gen_golden.py) and against the CPU oracle, through the C ABI.  ``pytest -m gpu``."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden, rel_err, rel_err_elementwise
from plan_interp import exec_op

pytestmark = pytest.mark.gpu

NMSE_RTOL = 1e-5      # BASELINE.json north_star: NMSE within 1e-5 relative of the reference


@pytest.fixture(scope='module', params=['bf16x3', 'f32', 'f16x2', 'f16x2+pairs', 'f16x2+pairs+res'])
def net64(weights64, request):
    """Every parity case runs with every fp32-class convolution multiplier (scorenet.CONV_MODES), with the 32-channel RCU
    blocks and CRP stages fused into single launches (``fuse_pairs``, csrc/conv_pair.hip), and with the two ResidualBlocks of the
    full-resolution level as one launch each (``fuse_res``, csrc/conv_res.hip; optional)."""
    import torch
    from score_based_channels_amd.scorenet import ScoreNet
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    cfg, sd = weights64
    mode, *opts = request.param.split('+')
    return ScoreNet(cfg, conv_mode=mode, fuse_pairs='pairs' in opts, fuse_res='res' in opts).cuda().load_state_dict(sd).eval()


def test_forward_every_op_matches_cpu_interpretation(net64, weights64):
    """Run the bound plan one launch at a time; after each launch compare its output with the CPU
    interpretation of the same record fed with the GPU's own inputs (errors cannot accumulate)."""
    import torch
    from score_based_channels_amd import _lib
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    B = 3
    bound = net64.bind(B, 64, 16)
    pl = bound.plan                       # (a batch this small: the plan with the skip branches on a launch lane -- list order is a valid order)
    bound.x.copy_(torch.from_numpy(g['x'][:B]).permute(0, 2, 3, 1))
    labels = np.array([0, 1155, 2310])
    bound.labels.copy_(torch.from_numpy(labels))
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bad = []
    for op, rec in zip(pl.ops, bound.ops):
        def get(t):
            return bound.slots[t.slot].view(B, t.h, t.w, t.c).cpu().numpy()
        ref = exec_op(op, sd, get, labels)
        _lib.check(_lib.lib().sbc_op_launch(C.byref(rec), stream))
        torch.cuda.synchronize()
        got = get(op.dst)
        err = rel_err(got, ref) if np.isfinite(got).all() else float('inf')
        if err > 5e-5:
            bad.append((op.name, err))
    assert not bad, bad[:10]


def test_forward_matches_reference_golden(net64):
    import torch
    g = load_golden('forward_64x16.npz')
    x = torch.from_numpy(g['x']).cuda()
    for i, lv in enumerate(g['levels']):
        out = net64(x, torch.full((4,), int(lv), dtype=torch.long, device='cuda'))
        assert tuple(out.shape) == (4, 2, 64, 16)
        assert rel_err(out.cpu().numpy(), g['out'][i]) < 2e-5, lv
        assert rel_err_elementwise(out.cpu().numpy(), g['out'][i], 0.02) < 1e-4, lv    # every element >= 2 % of the peak


def test_f16x2_is_fp32_class(weights64):
    """Acceptance gate of conv_mode f16x2 (two fp16 terms per operand on the fp16 matrix cores): its forward error against
    the reference golden is no larger than that of true fp32 matrix arithmetic (conv_mode f32, v_mfma_f32_32x32x2_f32) on the
    same inputs -- i.e. it is not a reduced-precision mode.  (Measured on MI355X: see DESIGN.md.)"""
    import torch
    from score_based_channels_amd import _lib
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    g = load_golden('forward_64x16.npz')
    x = torch.from_numpy(g['x']).cuda()
    err = {}
    for mode in ('f32', 'f16x2', 'bf16x3'):
        net = ScoreNet(cfg, conv_mode=mode).cuda().load_state_dict(sd).eval()
        err[mode] = max(rel_err(net(x, torch.full((4,), int(lv), dtype=torch.long, device='cuda')).cpu().numpy(), g['out'][i])
                        for i, lv in enumerate(g['levels']))
    print('forward error vs reference golden:', err)
    assert _lib.range_flag() == 0
    assert err['f16x2'] <= 1.05 * err['f32'] + 1e-7, err


def test_forward_accepts_reference_call_pattern(net64):
    """test_score.py:149-154: permuted view_as_real in, permute + contiguous + view_as_complex out."""
    import torch
    g = load_golden('forward_64x16.npz')
    cur = torch.from_numpy(np.ascontiguousarray(g['x'].transpose(0, 2, 3, 1))).cuda()
    current = torch.view_as_complex(cur)
    current_real = torch.view_as_real(current).permute(0, 3, 1, 2)
    labels = (torch.ones(4).cuda() * 1155).long()
    score = net64(current_real, labels)
    score = torch.view_as_complex(score.permute(0, 2, 3, 1).contiguous())
    ref = g['out'][1][:, 0] + 1j * g['out'][1][:, 1]
    assert rel_err(score.cpu().numpy(), ref) < 2e-5
    assert net64.sigmas.dtype == torch.float32 and net64.sigmas.shape == (2311,)
    assert abs(net64.sigmas[0].item() - 39.15) < 1e-5


def _run_golden_ald(net, g, use_graph=False):
    """All SNR points of a golden case as one lock-step batch, replaying the golden's keyed noise."""
    import torch
    from score_based_channels_amd.ald import AldBatch, snr_to_noise
    from score_based_channels_amd.noise import HostNoise
    H, P = g['H'], g['P']
    B, nt, nr = H.shape
    S = len(g['snr_db'])
    levels = [int(v) for v in g['levels']]
    n_steps = len(levels) * int(g['steps_each'])
    noise = HostNoise(int(g['seed']))
    ln = np.repeat(snr_to_noise(g['snr_db'], nt), B)
    idx = np.tile(np.arange(B), S)
    step_noise = np.concatenate([noise.step_block(s, H.shape, n_steps) for s in range(S)], axis=1)
    ald = AldBatch(net, H, P, idx, idx, ln, alpha_step=float(g['alpha_step']), beta_noise=float(g['beta_noise']),
                   levels=levels, steps_each=int(g['steps_each']), step_noise=torch.from_numpy(step_noise))
    ald.set_init(torch.from_numpy(np.tile(noise.init(H.shape), (S, 1, 1))))
    meas = np.concatenate([noise.measurement(s, g['Y'][s].shape) for s in range(S)], axis=0)
    Y = ald.synthesize_measurements(torch.from_numpy(meas))
    ald.run(use_graph=use_graph)
    torch.cuda.synchronize()
    log = ald.nmse_log().cpu().numpy().reshape(n_steps, S, B).transpose(1, 0, 2)
    X = ald.X.cpu().numpy().reshape(S, B, nt, nr)
    return Y.cpu().numpy().reshape(S, B, -1, nr), X, log


@pytest.mark.parametrize('name', ['ald_plumbing_level0.npz', 'ald_plumbing_3levels.npz', 'ald_trunc.npz',
                                  'ald_trunc_cell.npz'])
def test_ald_matches_reference_golden(net64, name):
    g = load_golden(name)
    Y, X, log = _run_golden_ald(net64, g)
    assert rel_err(Y, g['Y']) < 1e-6
    assert np.max(np.abs(log / g['nmse_log'] - 1)) < NMSE_RTOL
    assert rel_err(X, g['X_final']) < 1e-5


def test_ald_full_schedule_matches_reference_golden(net64):
    """All 2311 x 3 = 6933 Langevin steps (BASELINE config: full noise-level schedule)."""
    g = load_golden('ald_full.npz')
    _, X, log = _run_golden_ald(net64, g)
    assert log.shape[1] == 6933
    assert np.max(np.abs(log / g['nmse_log'] - 1)) < NMSE_RTOL
    avg = log.mean(-1)
    assert np.max(np.abs(avg.min(-1) / g['nmse_log'].mean(-1).min(-1) - 1)) < NMSE_RTOL
    assert rel_err(X, g['X_final']) < 1e-5


def test_ald_graph_replay_equals_eager(net64):
    g = load_golden('ald_trunc_cell.npz')
    _, X0, log0 = _run_golden_ald(net64, g, use_graph=False)
    _, X1, log1 = _run_golden_ald(net64, g, use_graph=True)
    assert np.array_equal(log0, log1) and np.array_equal(X0, X1)


def test_langevin_kernel_matches_oracle_step(net64):
    """Data-consistency gradient + update + NMSE alone (score supplied), vs oracle/ald_oracle.py."""
    import torch
    from oracle import ald_oracle as A
    from score_based_channels_amd import _lib, plan as P
    rng = np.random.default_rng(12)
    T, nt, nr, npil = 6, 64, 16, 38

    def cn(*s):
        return (rng.standard_normal(s) + 1j * rng.standard_normal(s)).astype(np.complex64)
    X, S, H, Y, Pm, nz = cn(T, nt, nr), cn(T, nt, nr), cn(T, nt, nr), cn(T, npil, nr), cn(T, npil, nt) / 8, cn(T, nt, nr)
    sched = np.array([[[0.3, 7.0, 0.05, 1.0]], [[1e-3, 0.2, 0.01, 2.5]]], np.float32)     # (alpha, dc_div, noise, dc_boost)
    group = np.array([0, 1, 0, 1, 1, 0], np.int32)
    d = {k: torch.from_numpy(v).cuda() for k, v in dict(X=X, S=S, H=H, Y=Y, P=Pm, nz=nz[None], sched=sched, group=group).items()}
    nm = torch.zeros(1, T, device='cuda')
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    r = torch.view_as_real
    ext = _lib.sbc_langevin(X=r(d['X']).data_ptr(), score=r(d['S']).data_ptr(), P=r(d['P']).data_ptr(),
                            Y=r(d['Y']).data_ptr(), Htrue=r(d['H']).data_ptr(), sched=d['sched'].data_ptr(),
                            group=d['group'].data_ptr(), noise=r(d['nz']).data_ptr(), nmse=nm.data_ptr(),
                            step=step.data_ptr(), n_steps=1, Nt=nt, Nr=nr, Np=npil)
    op = _lib.sbc_op(kind=P.LANGEVIN, B=T, ext=C.cast(C.pointer(ext), C.c_void_p))
    _lib.check(_lib.lib().sbc_op_launch(C.byref(op), None))
    torch.cuda.synchronize()
    for t in range(T):
        a, dv, ns, dcb = sched[group[t], 0]
        ref = A.langevin_step(X[t:t + 1], S[t:t + 1], Pm[t:t + 1], Y[t:t + 1], a, dv, ns, nz[t:t + 1], dc_boost32=dcb)
        assert rel_err(d['X'][t].cpu().numpy(), ref[0]) < 1e-5
        assert abs(nm[0, t].item() / A.nmse(ref, H[t:t + 1])[0] - 1) < 1e-5


def test_philox_noise_batch_independence(net64):
    """In-kernel noise: identical draws for a trajectory wherever it sits in a batch."""
    import torch
    from score_based_channels_amd.ald import AldBatch
    g = load_golden('ald_plumbing_level0.npz')
    H, P = g['H'], g['P']

    def run(order):
        ald = AldBatch(net64, H, P, order, order, 64.0, levels=[0], steps_each=2, seed=77, traj_id=order)
        ald.set_init(torch.zeros(len(order), 64, 16, dtype=torch.complex64))
        ald.synthesize_measurements()
        ald.run()
        torch.cuda.synchronize()
        return ald.Y.cpu().numpy(), ald.X.cpu().numpy()
    Ya, Xa = run(np.array([0, 1, 2, 3]))
    Yb, Xb = run(np.array([3, 1]))
    assert np.array_equal(Ya[3], Yb[0]) and np.array_equal(Ya[1], Yb[1])
    assert np.array_equal(Xa[3], Xb[0]) and np.array_equal(Xa[1], Xb[1])


def test_device_philox_known_answers():
    """The device's Philox4x32-10 (csrc/philox.h) returns the published Random123 known-answer vectors, and agrees with the
    host restatement (oracle/ald_oracle.py::philox4x32, pinned to the same vectors on the CPU) on 4096 random blocks."""
    from oracle import ald_oracle as A
    from score_based_channels_amd import _lib
    from test_oracle_golden import PHILOX_KAT
    rng = np.random.default_rng(3)
    ck = np.concatenate([np.array([list(c) + list(k) for c, k, _ in PHILOX_KAT], np.uint32),
                         rng.integers(0, 2 ** 32, size=(4096, 6), dtype=np.uint64).astype(np.uint32)])
    out = np.zeros((len(ck), 4), np.uint32)
    _lib.check(_lib.lib().sbc_debug_philox4x32(ck.ctypes.data, len(ck), out.ctypes.data))
    assert [tuple(int(v) for v in row) for row in out[:3]] == [o for _, _, o in PHILOX_KAT]
    assert np.array_equal(out, A.philox4x32(ck[:, :4], ck[:, 4:]))


def _device_normals(seed, traj, step, n):
    from score_based_channels_amd import _lib
    out = np.zeros((n, 2), np.float32)
    _lib.check(_lib.lib().sbc_debug_complex_normal(int(seed), int(traj), int(step), n, out.ctypes.data))
    return np.ascontiguousarray(out).view(np.complex64)[:, 0]


def test_device_noise_matches_host_restatement_and_is_standard_normal():
    """The CN(0,1) draws the Langevin / measurement kernels make (production default, ``--noise device``) against their host
    restatement -- same Philox blocks, float32 Box-Muller on both sides: differences are those of logf / sincosf between the
    GPU's math library and libm -- and their moments at N = 2^21 with 5-standard-error tolerances (the round-2 check accepted
    a variance anywhere in 0.45..0.55)."""
    from oracle import ald_oracle as A
    for seed, traj, step, n in ((1234, 7, 3, 4099), (2 ** 63 + 5, 2 ** 40 + 3, -1, 608), (0, 0, 0, 2)):
        dev, ref = _device_normals(seed, traj, step, n), A.device_complex_normal(seed, traj, step, n)
        assert np.max(np.abs(dev - ref)) < 4e-6, (seed, traj, step)
    n = 1 << 21
    z = _device_normals(99, 12345, 6000, n)
    se = 1.0 / np.sqrt(n)
    assert abs(z.real.mean()) < 5 * se * np.sqrt(0.5) and abs(z.imag.mean()) < 5 * se * np.sqrt(0.5)
    assert abs(z.real.var() - 0.5) < 5 * se * 0.5 * np.sqrt(2) and abs(z.imag.var() - 0.5) < 5 * se * 0.5 * np.sqrt(2)
    assert abs(np.mean(z.real * z.imag)) < 5 * se * 0.5
    assert abs(np.mean(np.abs(z) ** 4) - 2.0) < 5 * se * np.sqrt(20.0)
    assert abs(np.mean(z[:-1] * np.conj(z[1:]))) < 5 * se


def test_ald_with_in_kernel_noise_matches_oracle(net64, weights64):
    """The production noise path end to end: 4 channels x (3 levels x 3 steps) with in-kernel Philox noise for the measurements
    and every Langevin step, against the oracle loop fed the host-replayed device stream (keyed by seed, trajectory id, step)."""
    import torch
    from oracle import ald_oracle as A, ncsnv2_oracle as O
    from score_based_channels_amd.ald import AldBatch
    cfg, sd = weights64
    g = load_golden('ald_plumbing_3levels.npz')
    H, P = g['H'], g['P']
    B, nt, nr = H.shape
    npil = P.shape[1]
    levels = [int(v) for v in g['levels']]
    seed, traj = 2025, np.array([11, 2 ** 33 + 1, 5, 0], np.int64)
    ln = 64.0
    ald = AldBatch(net64, H, P, np.arange(B), np.arange(B), ln, levels=levels, steps_each=3, seed=seed, traj_id=traj)
    init = A.device_complex_normal(5, 0, 0, B * nt * nr).reshape(B, nt, nr)
    ald.set_init(torch.from_numpy(init))
    Y = ald.synthesize_measurements().cpu().numpy()
    ald.run()
    torch.cuda.synchronize()
    got = ald.nmse_log().cpu().numpy()
    meas = np.stack([A.device_complex_normal(seed, traj[b], -1, npil * nr).reshape(npil, nr) for b in range(B)])
    Yo = A.make_measurements(P, H, ln, meas)
    assert rel_err(Y, Yo) < 1e-6

    def step_noise(k):
        return np.stack([A.device_complex_normal(seed, traj[b], k, nt * nr).reshape(nt, nr) for b in range(B)])
    Xo, ref = A.ald_run(lambda x, lab: O.score_forward(sd, x, lab), sd['sigmas'], cfg.model.sigma_end, P, Yo, H, init,
                        step_noise, ln, levels=levels)
    assert np.max(np.abs(got / ref - 1)) < NMSE_RTOL
    assert rel_err(ald.X.cpu().numpy(), Xo) < 1e-5


def test_big_array_forward_and_ald(weights64):
    """256 x 64 antennas (BASELINE config 5 geometry, fp32): forward + 6 Langevin steps vs the reference."""
    import torch
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.scorenet import ScoreNet
    from score_based_channels_amd.weights import seeded_state_dict
    cfg = default_config(image_size=(64, 256))
    net = ScoreNet(cfg).cuda().load_state_dict(seeded_state_dict(cfg, 2024))
    g = load_golden('big_256x64.npz')
    out = net(torch.from_numpy(g['x']).cuda(), torch.full((1,), int(g['level']), dtype=torch.long))
    assert rel_err(out.cpu().numpy(), g['out']) < 2e-5
    _, X, log = _run_golden_ald(net, dict(g, steps_each=3, alpha_step=3e-11, beta_noise=0.01))
    assert np.max(np.abs(log / g['nmse_log'] - 1)) < NMSE_RTOL
    assert rel_err(X, g['X_final']) < 1e-5


# conv_mode 'f16w' (BASELINE config 5, "fp16 score-net weights").  Oracle: the fp32 reference with fp16-rounded
# parameters (goldens f16w_*.npz; the reference cannot run .half() itself, layers.py:179).  The HIP path additionally
# rounds the activations to fp16 where they enter the matrix cores (11-bit significands, fp32 accumulation), so the
# tolerance is looser than the fp32 contract.  Measured on MI355X: score error 0.92e-3 (64x16) / 1.8e-3 (256x64) of the
# largest score value, NMSE within 4.4e-6 / 1.1e-6 relative at every logged step (93-step truncated schedule / 6 steps);
# asserted with margin:
F16W_FWD_TOL = 4e-3        # score, max-abs error over max-abs value
F16W_NMSE_RTOL = 1e-4      # NMSE at every logged step, relative


def _f16w_net(cfg, sd):
    from score_based_channels_amd.scorenet import ScoreNet
    return ScoreNet(cfg, conv_mode='f16w').cuda().load_state_dict(sd).eval()


def test_f16w_matches_reference_with_fp16_weights(weights64):
    import torch
    cfg, sd = weights64
    net = _f16w_net(cfg, sd)
    g = load_golden('f16w_64x16.npz')
    x = torch.from_numpy(g['x']).cuda()
    errs = []
    for i, lv in enumerate(g['levels']):
        out = net(x, torch.full((4,), int(lv), dtype=torch.long, device='cuda'))
        errs.append(rel_err(out.cpu().numpy(), g['out'][i]))
    ga = dict(g, levels=g['ald_levels'], steps_each=3, alpha_step=3e-11, beta_noise=0.01)
    _, X, log = _run_golden_ald(net, ga)
    dev = float(np.max(np.abs(log / g['nmse_log'] - 1)))
    print('f16w 64x16: forward errors %s, NMSE deviation %.2e, X %.2e' % (errs, dev, rel_err(X, g['X_final'])))
    assert max(errs) < F16W_FWD_TOL and dev < F16W_NMSE_RTOL


def test_f16w_big_array_matches_reference_with_fp16_weights():
    """256 x 64 antennas with fp16 weights: golden G6 (forward + 6 Langevin steps)."""
    import torch
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.weights import seeded_state_dict
    cfg = default_config(image_size=(64, 256))
    net = _f16w_net(cfg, seeded_state_dict(cfg, 2024))
    g = load_golden('f16w_256x64.npz')
    out = net(torch.from_numpy(g['x']).cuda(), torch.full((1,), int(g['level']), dtype=torch.long))
    e = rel_err(out.cpu().numpy(), g['out'])
    _, X, log = _run_golden_ald(net, dict(g, steps_each=3, alpha_step=3e-11, beta_noise=0.01))
    dev = float(np.max(np.abs(log / g['nmse_log'] - 1)))
    print('f16w 256x64: forward error %.2e, NMSE deviation %.2e' % (e, dev))
    assert e < F16W_FWD_TOL and dev < F16W_NMSE_RTOL


def test_config5_full_batch_is_the_sum_of_its_trajectories():
    """BASELINE config 5 at full size: 1024 lock-step trajectories on 256 x 64 arrays (153 pilots) with fp16 weights --
    16 GB of activation slots.  Size-independent property: a trajectory's estimate and NMSE log are bit-identical to the
    same trajectory run in a batch of three; with the 256x64 goldens above this pins the full-size run."""
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import AldBatch, snr_to_noise
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.weights import seeded_state_dict
    cfg = default_config(image_size=(64, 256))
    net = _f16w_net(cfg, seeded_state_dict(cfg, 2024))
    nch, nt, nr, npil = 64, 256, 64, 153
    raw = synth.generate_channels('ULA', nch, nt, nr, 0.5, seed=31)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(32), nch, nt, npil), (0, 2, 1)))
    snr = np.arange(-10, 30, 2.5)                                        # 16 SNR points x 64 channels = 1024
    idx = np.tile(np.arange(nch), len(snr))
    ln = np.repeat(snr_to_noise(snr, nt), nch)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, generator=torch.Generator().manual_seed(6))

    def run(sel):
        ald = AldBatch(net, H, Pm, idx[sel], idx[sel], ln[sel], levels=[0, 1155], steps_each=2, seed=9, traj_id=sel)
        ald.set_init(init[torch.from_numpy(idx[sel])])
        ald.synthesize_measurements()
        ald.run()
        torch.cuda.synchronize()
        out = ald.X.cpu().numpy(), ald.nmse_log().cpu().numpy()
        ald.close()
        return out
    Xa, La = run(np.arange(1024))
    assert np.isfinite(La).all() and La.shape == (4, 1024)
    pick = np.array([0, 511, 1023])
    Xb, Lb = run(pick)
    assert np.array_equal(Xa[pick], Xb) and np.array_equal(La[:, pick], Lb)


def test_cli_test_score_drop_in_outputs(net64, tmp_path, monkeypatch):
    """``python -m score_based_channels_amd.test_score`` keeps the reference's result file (test_score.py:192-200)."""
    import torch
    from score_based_channels_amd import test_score
    monkeypatch.chdir(tmp_path)
    argv = ['--synthetic', '--synthetic_weights', '2024', '--num_levels', '2', '--num_channels', '4', '--seed', '3',
            '--no_plot']
    nmse_log, avg, best = test_score.main(argv)
    assert nmse_log.shape == (1, 1, 17, 6, 4) and nmse_log.dtype == np.float64 and np.isfinite(nmse_log).all()
    res = torch.load(tmp_path / 'results/score/train-CDL-C_test-CDL-C/results.pt', weights_only=False)
    assert {'nmse_log', 'avg_nmse', 'best_nmse', 'spacing_range', 'pilot_alpha_range', 'snr_range',
            'val_config'} <= set(res)
    assert np.array_equal(res['best_nmse'], np.min(np.mean(nmse_log, -1), -1)) and res['val_config'].data.num_pilots == 38
    again, _, _ = test_score.main(argv + ['--no_graph', '--save_channels', '1'])   # same seed => same result, graph or eager
    assert np.array_equal(again, nmse_log)
    res = torch.load(tmp_path / 'results/score/train-CDL-C_test-CDL-C/results.pt', weights_only=False)
    assert res['saved_H'].shape == (1, 1, 17, 4, 64, 16) and res['saved_H'].dtype == np.complex64
    assert np.isfinite(res['saved_H'].view(np.float32)).all()
    # low SNR must not beat high SNR after the same number of steps on the same channels (sanity of per-SNR scalars)
    assert not np.array_equal(nmse_log[0, 0, 0], nmse_log[0, 0, -1])
    split, _, _ = test_score.main(argv + ['--streams', '2'])                        # ... or as two concurrent sub-batches
    assert np.array_equal(split, nmse_log)


def test_cli_tune_hparams_drop_in_outputs(net64, tmp_path, monkeypatch):
    import torch
    from score_based_channels_amd import tune_hparams_score
    monkeypatch.chdir(tmp_path)
    nmse_log, ba, bb = tune_hparams_score.main(['--synthetic', '--synthetic_weights', '2024', '--num_levels', '1',
                                                '--num_channels', '3', '--seed', '4', '--no_plot',
                                                '--alpha_step_range', '3e-11', '3e-10', '--beta_noise_range', '0.1', '0.01'])
    assert nmse_log.shape == (2, 2, 17, 3, 3) and np.isfinite(nmse_log).all() and len(ba) == 17 and len(bb) == 17
    res = torch.load(tmp_path / 'results/score/CDL-C-hyperparameters.pt', weights_only=False)
    assert {'nmse_log', 'avg_nmse', 'best_nmse', 'best_alpha_snr', 'best_beta_snr', 'snr_range', 'alpha_step_range',
            'beta_noise_range', 'config', 'args'} <= set(res)
    assert set(ba) <= {3e-11, 3e-10} and set(bb) <= {0.1, 0.01}


def test_cli_test_mmse_posterior_mean(net64, tmp_path, monkeypatch):
    """``python -m score_based_channels_amd.test_mmse``: chains of a sample share its measurement, the
    stored NMSE log is the NMSE of the stored chains, early stopping follows the hyper-parameter file."""
    import torch
    from score_based_channels_amd import test_mmse
    monkeypatch.chdir(tmp_path)
    S = 19
    torch.save({'best_step_idx': np.full((1, S), 3e-11), 'best_noise_idx': np.full((1, S), 0.01),
                'best_stop_idx': np.full((1, S), 3, dtype=np.int64)}, tmp_path / 'our_hyperparams_CDL-C.pt')
    argv = ['--synthetic', '--synthetic_weights', '2024', '--num_levels', '2', '--kept_samples', '3', '--mmse_avg', '4',
            '--seed', '5']
    log, saved, mmse_nmse = test_mmse.main(argv)
    assert log.shape == (1, 1, S, 6, 3, 4) and saved.shape == (1, 1, S, 3, 4, 64, 16) and mmse_nmse.shape == (1, 1, S, 3)
    assert np.all(log[0, 0, :, 4:] == 0) and np.all(log[0, 0, :, :4] > 0)          # stopped after step index 3
    res = torch.load(tmp_path / 'TWC_rebuttal_MMSE_aug6_seed4321/model_CDL-C_channel_CDL-C.pt', weights_only=False)
    assert {'oracle_log', 'oracle_H', 'saved_H', 'snr_range', 'spacing_range', 'pilot_alpha_range', 'config',
            'val_config', 'args'} <= set(res)
    H = res['oracle_H']
    nm = np.sum(np.abs(saved[0, 0] - H[None, :, None]) ** 2, axis=(-1, -2)) / np.sum(np.abs(H) ** 2, axis=(-1, -2))[None, :, None]
    assert np.allclose(nm, log[0, 0, :, 3], rtol=1e-4)                                # log row at the stop == stored chains
    assert not np.allclose(saved[0, 0, 0, 0, 0], saved[0, 0, 0, 0, 1])                # chains differ (own start + noise)
    adj, _, _ = test_mmse.main(argv + ['--start_point', 'Adjoint', '--no_graph'])
    assert np.isfinite(adj).all() and not np.array_equal(adj, log)
    split, saved2, _ = test_mmse.main(argv + ['--streams', '2'])
    assert np.array_equal(split, log) and np.array_equal(saved2, saved)


@pytest.mark.parametrize('start', ['Noise', 'Adjoint', 'LS'])
def test_posterior_mean_chains_match_reference_golden(net64, start):
    """F2 parity: ``test_mmse.posterior_chains`` (chains sharing a measurement, per-SNR step / noise / stop, dc_boost,
    start points) against the golden produced by the transcription of test_mmse.py:166-277 around the reference network."""
    from score_based_channels_amd.noise import HostNoise
    from score_based_channels_amd.test_mmse import posterior_chains
    g = load_golden('mmse_ls.npz' if start == 'LS' else 'mmse.npz')      # --start_point LS: test_mmse.py:200-202
    H, P, navg = g['H'], g['P'], int(g['mmse_avg'])
    levels = [int(v) for v in g['levels']]
    for s, snr in enumerate(g['snr_db']):
        stop = int(g['best_stop'][s])
        Y, log, est = posterior_chains(net64, H, P, 10 ** (-snr / 10.), float(g['best_step'][s]), float(g['best_noise'][s]),
                                       stop + 1, levels, 3, navg, start_point=start, dc_boost=float(g['dc_boost']),
                                       use_graph=bool(s), host_noise=HostNoise(int(g['seed']), combo=1 + s))
        ref = g['oracle_log_' + start][s]
        assert rel_err(Y.cpu().numpy(), g['Y'][s]) < 1e-6
        assert log.shape == (stop + 1, 2, navg) and np.max(np.abs(log / ref[:stop + 1] - 1)) < NMSE_RTOL
        assert rel_err(est, g['saved_H_' + start][s]) < 1e-5


@pytest.mark.parametrize('foreign', ['CDL-D', 'CDL-B', 'CDL-A'])
def test_cli_cross_profile_matches_reference_pipeline(net64, tmp_path, monkeypatch, foreign):
    """BASELINE config 4 (``--train CDL-C --test CDL-D`` / ``CDL-B``) end to end: the golden ran the reference's own loader
    (normalisation constants from the TRAIN profile, test_score.py:68-69,101), DataLoader batch and sampling loop on the
    same synthetic files and keyed noise; the CLI must reproduce its NMSE log from the same command line."""
    from score_based_channels_amd import test_score
    g = load_golden('cli_cross_cdlc_%s.npz' % foreign.replace('-', '').lower())
    monkeypatch.chdir(tmp_path)
    argv = str(g['argv']).split() + ['--conv_mode', net64.conv_mode]
    nmse_log, _, _ = test_score.main(argv)
    assert nmse_log.shape == (1, 1, 17, g['nmse_log'].shape[1], g['H'].shape[0])
    assert np.max(np.abs(nmse_log[0, 0] / g['nmse_log'] - 1)) < NMSE_RTOL
    est = test_score.main(argv + ['--save_channels', '1', '--no_graph'])
    import torch
    res = torch.load(tmp_path / ('results/score/train-CDL-C_test-%s/results.pt' % foreign), weights_only=False)
    assert rel_err(res['saved_H'][0, 0], g['X_final']) < 1e-5
    # the same channels normalised by their OWN profile would differ: the train-profile constants really are used
    same, _, _ = test_score.main([a if a != 'CDL-C' else foreign for a in argv])
    assert np.max(np.abs(same[0, 0] / g['nmse_log'] - 1)) > 1e-3
    if foreign == 'CDL-A':
        # 6 channels x 17 SNR points = 102 trajectories in chunks of 40 (driver.run_trajectories: 40 + 40 + 22, each split over the
        # sub-batch streams): chunk boundaries inside the reference's result change nothing, bit for bit
        chunked, _, _ = test_score.main(argv + ['--max_batch', '40'])
        assert np.array_equal(chunked, nmse_log)


@pytest.mark.parametrize('name', ['cli_tune_grid.npz', 'cli_tune_grid_2levels.npz'])
def test_cli_tuner_matches_reference_pipeline(net64, tmp_path, monkeypatch, name):
    """BASELINE config 3 at reduced size: a 2 x 2 (alpha, beta) grid through the tuner CLI vs the reference pipeline
    (one validation dataset, pilot draw and noise stream per cell, tune_hparams_score.py:71-97), over one noise level and over
    two (the per-level scalars change between levels)."""
    from score_based_channels_amd import tune_hparams_score
    g = load_golden(name)
    n_levels = int(g['num_levels']) if 'num_levels' in g else 1
    monkeypatch.chdir(tmp_path)
    argv = (['--synthetic', '--synthetic_weights', '2024', '--num_levels', str(n_levels), '--num_channels', '3',
             '--seed', str(int(g['seed'])), '--no_plot', '--noise', 'host', '--conv_mode', net64.conv_mode,
             '--alpha_step_range'] + [repr(float(a)) for a in g['alpha_step_range']] +
            ['--beta_noise_range'] + [repr(float(b)) for b in g['beta_noise_range']])
    nmse_log, ba, bb = tune_hparams_score.main(argv)
    assert nmse_log.shape == (2, 2, 17, 3 * n_levels, 3)
    assert np.max(np.abs(nmse_log / g['nmse_log'] - 1)) < NMSE_RTOL
    split, _, _ = tune_hparams_score.main(argv + ['--streams', '2'])
    assert np.array_equal(split, nmse_log)


def test_concurrent_sub_batch_streams_do_not_change_results(net64, monkeypatch):
    """``run_trajectories(n_streams=2)``: two sub-batches on their own HIP streams, one host thread each
    (driver.run_concurrently).  Per-trajectory noise keys and per-sample normalisation make the split invisible -- and so
    must the hardware: with packed-fp32 instructions in the build, concurrently running kernels of different kinds corrupted
    each other's results here in 8 of 12 runs (csrc/Makefile)."""
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import snr_to_noise
    from score_based_channels_amd.driver import run_trajectories
    nch, nt, nr, npil = 24, 64, 16, 38
    raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=41)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(42), nch, nt, npil), (0, 2, 1)))
    snr = np.array([-10.0, 0.0, 12.5, 30.0])
    idx = np.tile(np.arange(nch), len(snr))
    ln = np.repeat(snr_to_noise(snr, nt), nch)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, generator=torch.Generator().manual_seed(7))
    modes = [(1, False)] + [(2 + k % 2, bool(k & 2)) for k in range(8)]       # repeated: the hazard this guards is sporadic
    # (a chunk this small would run as ONE batch with launch lanes whatever n_streams says, driver.stream_count: the lane plan is
    # switched off here so that the sub-batch streams really run; tests/...::test_skip_overlap_plan_equals_sequential_plan has the lanes)
    monkeypatch.setattr(net64, 'skip_overlap', False)
    out = [run_trajectories(net64, H, Pm, idx, idx, ln, 3e-11, 0.01, [0, 1155, 2310], 3, 11, init, n_streams=n,
                            use_graph=g, return_final=True) for n, g in modes]
    for (n, g), (log, est) in zip(modes[1:], out[1:]):
        assert np.array_equal(log, out[0][0]) and np.array_equal(est, out[0][1]), (n, g)


def test_lagging_second_stream_does_not_change_results(net64, monkeypatch):
    """The second of two sub-batch streams walks the schedule ~0.45 of a step behind the first: the leader runs the head of its first
    step alone, the follower the rest of its last step (``AldBatch.run_leading`` / ``run_following``, driver.run_concurrently: one
    step's records cut in two plans, two device events).  Every stream still runs exactly its own records in order: logs and
    estimates equal the one-stream run bit for bit."""
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import snr_to_noise
    from score_based_channels_amd.driver import run_trajectories
    nch, nt, nr, npil = 12, 64, 16, 38
    raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=43)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(44), nch, nt, npil), (0, 2, 1)))
    snr = np.array([-5.0, 20.0])
    idx = np.tile(np.arange(nch), len(snr))
    ln = np.repeat(snr_to_noise(snr, nt), nch)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, generator=torch.Generator().manual_seed(8))
    args = (net64, H, Pm, idx, idx, ln, 3e-11, 0.01, [0, 1155, 2310], 3, 13, init)
    one = run_trajectories(*args, n_streams=1, return_final=True)            # (one batch, skip branches on a launch lane)
    monkeypatch.setattr(net64, 'skip_overlap', False)                        # (... so that n_streams = 2 means two streams: driver.stream_count)
    monkeypatch.setenv('SBC_STREAM_LAG_MIN_STEPS', '1')
    for _ in range(3):
        two = run_trajectories(*args, n_streams=2, return_final=True)
        assert np.array_equal(two[0], one[0]) and np.array_equal(two[1], one[1])


@pytest.mark.timeout(300)
def test_failing_leader_stream_does_not_hang_its_follower(net64, monkeypatch):
    """The following stream waits on the HOST until its leader has recorded the event it is to wait for on the device
    (driver.run_concurrently).  A leader that fails -- here before its first launch -- releases it all the same: the call raises the
    leader's error instead of hanging."""
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import AldBatch, snr_to_noise
    from score_based_channels_amd.driver import run_trajectories
    nch, nt, nr, npil = 8, 64, 16, 38
    raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=45)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(46), nch, nt, npil), (0, 2, 1)))
    idx = np.arange(nch)
    ln = np.repeat(snr_to_noise(np.array([10.0]), nt), nch)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, generator=torch.Generator().manual_seed(9))

    def boom(self, *a, **k):
        raise RuntimeError('leader failed')
    monkeypatch.setattr(AldBatch, 'run_leading', boom)
    # (a chunk this small would run as ONE batch with launch lanes, driver.stream_count: switch the lane plan off to get the two streams)
    monkeypatch.setattr(net64, 'skip_overlap', False)
    with pytest.raises(RuntimeError, match='leader failed'):
        run_trajectories(net64, H, Pm, idx, idx, ln, 3e-11, 0.01, [0, 1155, 2310], 3, 13, init, n_streams=2)


@pytest.mark.parametrize('nt,nr', [(16, 64), (32, 32), (128, 8)])
def test_forward_other_geometries_match_oracle(nt, nr, weights64):
    """Array shapes the reference goldens do not cover (wide images, square images, 8-column images): every level still
    tiles, and the forward agrees with the (reference-pinned) oracle in both multiplier modes."""
    import torch
    from oracle import ncsnv2_oracle
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.scorenet import ScoreNet
    _, sd = weights64                                     # weights do not depend on the array shape
    cfg = default_config(image_size=(nr, nt))
    x = np.random.default_rng(nt * 1000 + nr).standard_normal((3, 2, nt, nr)).astype(np.float32)
    labels = np.array([0, 1155, 2310])
    ref = ncsnv2_oracle.score_forward(sd, x, labels)
    for mode, pairs in (('bf16x3', False), ('f32', False), ('f16x2', False), ('f16x2', True)):   # (the last one: the shipped default)
        net = ScoreNet(cfg, conv_mode=mode, fuse_pairs=pairs).cuda().load_state_dict(sd).eval()
        out = net(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda())
        assert rel_err(out.cpu().numpy(), ref) < 2e-5, (mode, pairs, nt, nr)
        assert net.range_fallbacks == 0


def _scaled_msf_weights(sd, factor):
    """The state dict with refine5's two MSF convolutions (weight and bias, layers.py:178-184) multiplied by ``factor``.  Everything
    behind them -- the CRP block and the three output RCU blocks at full resolution, none of which normalises (layers.py:76-83,
    126-134) -- then runs on activations ``factor`` times smaller, and the final InstanceNorm++ (ncsnv2.py:291) brings the result
    back to O(1): every error made on the small tensors shows in the score at full size."""
    out = dict(sd)
    n = 0
    for k, v in sd.items():
        if k.startswith('refine5.msf.convs.'):
            out[k] = (np.asarray(v, np.float32) * np.float32(factor)).astype(np.float32)
            n += 1
    assert n == 4
    return out


def _forward_case(weights64, log2_factor):
    from oracle import ncsnv2_oracle
    cfg, sd = weights64
    sd2 = _scaled_msf_weights(sd, 2.0 ** log2_factor)
    g = load_golden('forward_64x16.npz')
    x, labels = g['x'][:3], np.array([0, 1155, 2310])
    return cfg, sd2, x, labels, ncsnv2_oracle.score_forward(sd2, x, labels)


@pytest.mark.parametrize('log2_factor', [-3, 3, 8])
def test_f16x2_holds_the_tolerance_when_the_refine_activations_are_not_o1(weights64, log2_factor):
    """The reference computes in IEEE fp32 (test_score.py:25-26): its relative precision does not depend on how large a
    checkpoint's un-normalised activations are.  The shipped default (f16x2 + fused pairs + folded statistics) must not either:
    with refine5's activations 8x smaller or 8x / 256x larger the forward agrees with the oracle on the SAME weights at the
    tolerance of the unscaled golden -- through the per-layer activation scales of sbc_f16x2_calibrate, with the range flag clear
    and no fallback."""
    import torch
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd2, x, labels, ref = _forward_case(weights64, log2_factor)
    net = ScoreNet(cfg).cuda().load_state_dict(sd2).eval()
    assert net.conv_mode == 'f16x2' and net.fuse_pairs and net.fold_stats
    out = net(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()).cpu().numpy()
    assert net.range_fallbacks == 0
    assert rel_err(out, ref) < 2e-5, rel_err(out, ref)


@pytest.mark.parametrize('log2_factor', [-7, -10, -13])
def test_small_refine_activations_hold_the_tolerance_in_every_mode(weights64, log2_factor):
    """refine5's activations 128x ... 8192x smaller (1e-2 ... 1e-4): below 2^-4 the hardware form of ELU, exp(x) - 1, no longer has
    fp32's relative accuracy (6e-8 absolute), and with act_scale = 1 the two-term fp16 split would not either.
      * bf16x3 (the exact mode: ELU with relative accuracy everywhere, SBC_PRO_ELU_ACC) agrees with the oracle;
      * f16x2 without pair fusion does too, with NO fallback: calibrated activation scales + the accurate ELU its calibration
        requests per layer;
      * the shipped default meets its fused RCU kernels there, which evaluate ELU as exp(x) - 1 only: they raise the range flag
        (SBC_RANGE_ELU) and the call is answered by the bf16x3 network in the same process -- still within tolerance."""
    import torch
    from score_based_channels_amd import _lib
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd2, x, labels, ref = _forward_case(weights64, log2_factor)
    xt, lt = torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()
    exact = ScoreNet(cfg, conv_mode='bf16x3').cuda().load_state_dict(sd2).eval()
    assert rel_err(exact(xt, lt).cpu().numpy(), ref) < 2e-5
    unfused = ScoreNet(cfg, conv_mode='f16x2', fuse_pairs=False).cuda().load_state_dict(sd2).eval()
    out = unfused(xt, lt).cpu().numpy()
    assert unfused.range_fallbacks == 0 and rel_err(out, ref) < 2e-5, (unfused.range_fallbacks, rel_err(out, ref))
    net = ScoreNet(cfg).cuda().load_state_dict(sd2).eval()
    out = net(xt, lt).cpu().numpy()
    assert net.range_fallbacks == 1 and net.last_range_bits & _lib.RANGE_ELU
    assert rel_err(out, ref) < 2e-5


def test_f16x2_guard_reruns_in_bf16x3_in_process(weights64, monkeypatch):
    """Without the calibration (act_scale = 1 everywhere, what rounds 2-3 shipped) small refine5 activations put whole regions
    below 2^-6: the kernels raise the underflow bit, and the host answers the call from the bf16x3 network in the same process
    (module call) / re-runs the chunk (driver.run_trajectories) and says so -- never silently degraded numbers."""
    import torch
    from score_based_channels_amd import _lib, synth
    from score_based_channels_amd.ald import snr_to_noise
    from score_based_channels_amd.driver import run_trajectories
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd2, x, labels, ref = _forward_case(weights64, -10)
    monkeypatch.setenv('SBC_NO_CALIB', '1')
    net = ScoreNet(cfg).cuda().load_state_dict(sd2).eval()
    out = net(torch.from_numpy(x).cuda(), torch.from_numpy(labels).cuda()).cpu().numpy()
    assert net.range_fallbacks == 1 and net.last_range_bits & _lib.RANGE_UNDERFLOW
    assert rel_err(out, ref) < 2e-5
    # the sampling loop: the chunk is run again, the result-file entry says so, and the numbers are the bf16x3 run's
    nch, nt, nr, npil = 3, 64, 16, 38
    raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=2)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(3), nch, nt, npil), (0, 2, 1)))
    ln = snr_to_noise(np.zeros(nch), nt)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, generator=torch.Generator().manual_seed(1))
    info = {}
    log = run_trajectories(net, H, Pm, np.arange(nch), np.arange(nch), ln, 3e-11, 0.01, [0, 2310], 3, 5, init, info=info)
    assert len(info['f16x2_fallback']) == 1 and info['f16x2_fallback'][0]['rerun_in'] == 'bf16x3'
    exact = ScoreNet(cfg, conv_mode='bf16x3').cuda().load_state_dict(sd2).eval()
    log_exact = run_trajectories(exact, H, Pm, np.arange(nch), np.arange(nch), ln, 3e-11, 0.01, [0, 2310], 3, 5, init)
    assert np.array_equal(log, log_exact)


def test_full_batch_is_the_sum_of_its_trajectories(net64):
    """BASELINE config 2 size (100 channels x 17 SNR points = 1700 lock-step trajectories): a trajectory's estimate and
    NMSE log do not depend on what else is in the batch -- bit for bit, although tiles of the low-resolution levels
    hold 2 or 8 samples.  Together with the small-batch goldens this pins the full-size run."""
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import AldBatch, snr_to_noise
    nch, nt, nr, npil = 100, 64, 16, 38
    raw = synth.generate_channels('CDL-C', nch, nt, nr, 0.5, seed=11)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(12), nch, nt, npil), (0, 2, 1)))
    snr = np.arange(-10, 32.5, 2.5)
    idx = np.tile(np.arange(nch), len(snr))
    ln = np.repeat(snr_to_noise(snr, nt), nch)
    init = torch.randn(nch, nt, nr, dtype=torch.complex64, generator=torch.Generator().manual_seed(5))

    def run(sel):
        ald = AldBatch(net64, H, Pm, idx[sel], idx[sel], ln[sel], levels=[0, 1155], steps_each=3, seed=9, traj_id=sel)
        ald.set_init(init[torch.from_numpy(idx[sel])])
        ald.synthesize_measurements()
        ald.run()
        torch.cuda.synchronize()
        return ald.X.cpu().numpy(), ald.nmse_log().cpu().numpy()
    allsel = np.arange(len(idx))
    Xa, La = run(allsel)
    assert np.isfinite(La).all() and La.shape == (6, 1700)
    pick = np.array([0, 777, 1203, 1699])
    Xb, Lb = run(pick)
    assert np.array_equal(Xa[pick], Xb) and np.array_equal(La[:, pick], Lb)


def test_overlapped_branches_do_not_change_results(weights64):
    """``ScoreNet(overlap=True)``: shortcut convolutions and the second input's adapt / MSF convolutions of the low-resolution
    RefineBlocks run on the plan's side stream (SBC_OP_SIDE / SBC_OP_JOIN), with activation slots kept live until the join.
    Same kernels on the same data: the score must be bit-identical to the sequential plan, launch after launch."""
    import torch
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((96, 2, 64, 16)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, 2311, 96))
    # (the overlapped plan keeps the low-resolution levels as separate records -- side records must stay individual launches -- so the
    # sequential plan it is compared with is built without SBC_OP_CHAIN / SBC_OP_CONV_DOWN records too: same kernels on the same data)
    seq = ScoreNet(cfg, overlap=False, fuse_chain=False, fuse_down=False).cuda().load_state_dict(sd)
    ovl = ScoreNet(cfg, overlap=True).cuda().load_state_dict(sd)
    ref = seq(x, labels)
    assert sum(o.side for o in ovl.score_plan(64, 16).ops) == 24
    for rep in range(6):
        assert torch.equal(ovl(x, labels), ref), rep


def test_skip_overlap_plan_equals_sequential_plan(weights64):
    """Round 6: batches of at most scorenet.SKIP_OVERLAP_MAX_T trajectories run the decoder's skip branches (refineK.adapt_convs.0) on
    launch lanes of their own (sbc_op.lane / signal / wait, plan.hoist_skip_branches) beside the latency-bound low-resolution launches.
    Same records, same arguments, other order and streams: bit-identical to the sequential plan -- module call, eager Langevin steps and
    hipGraph replay (a captured graph is flat: the lane records in list order on the run stream, csrc/api.hip) -- also when two sub-batches
    with lanes share the chip, and repeatedly (a race would show as a flicker)."""
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import AldBatch
    from score_based_channels_amd.driver import run_concurrently
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((96, 2, 64, 16)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, 2311, 96))
    seq = ScoreNet(cfg, skip_overlap=False).cuda().load_state_dict(sd)
    ovl = ScoreNet(cfg).cuda().load_state_dict(sd)
    assert not any(o.lane for o in seq.score_plan(64, 16, 96).ops) and not any(o.lane for o in ovl.score_plan(64, 16, 4096).ops)
    lanes = [o for o in ovl.score_plan(64, 16, 96).ops if o.lane]
    assert len(lanes) == 6 and {o.name.split('.')[0] for o in lanes} == {'refine2', 'refine31', 'refine3', 'refine4', 'refine5'}
    ref = seq(x, labels)
    for rep in range(6):
        assert torch.equal(ovl(x, labels), ref), rep
    # Langevin steps: one stream, two sub-batch streams, graph replay
    T = 192
    raw = synth.generate_channels('CDL-C', 16, 64, 16, 0.5, 3)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(1), 16, 64, 38), (0, 2, 1)))
    idx = np.arange(T) % 16
    ln = np.repeat(64.0 * 10 ** (-np.arange(-10, 20, 2.5) / 10.), 16)[:T]
    init = torch.randn(16, 64, 16, dtype=torch.complex64, generator=torch.Generator().manual_seed(2))

    def run(net, parts, graph):
        alds = []
        for part in np.array_split(np.arange(T), parts):
            a = AldBatch(net, H, Pm, idx[part], idx[part], ln[part], levels=[0, 700, 2310], steps_each=3, seed=11, traj_id=part)
            a.set_init(init[torch.from_numpy(idx[part])])
            a.synthesize_measurements()
            alds.append(a)
        streams = [torch.cuda.Stream() for _ in alds] if parts > 1 else [torch.cuda.current_stream()]
        run_concurrently(alds, streams, 9, graph)
        torch.cuda.synchronize()
        X = torch.cat([a.X for a in alds]).cpu().numpy()
        L = torch.cat([a.nmse_log() for a in alds], dim=1).cpu().numpy()
        lanes_used = [a.uses_lanes for a in alds]
        for a in alds:
            a.close()
        return X, L, lanes_used
    X0, L0, u0 = run(seq, 1, False)
    assert u0 == [False]
    for parts, graph in ((1, False), (2, False), (1, True), (2, True), (1, False)):
        X1, L1, u1 = run(ovl, parts, graph)
        assert all(u1), (parts, graph)
        assert np.array_equal(X0, X1) and np.array_equal(L0, L1), (parts, graph)


def test_two_sub_batches_in_one_plan_equal_two_runs(weights64):
    """ald.AldPair: two sub-batches as ONE plan -- A on the run stream, B on a launch lane, several Langevin steps
    unrolled, B lagging behind the head of A's first step -- issued by one host thread or replayed as ONE hipGraph with two branches.
    Every record is a record of A's or B's own step plan: both equal the batches run alone, bit for bit."""
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import AldBatch, AldPair
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    net = ScoreNet(cfg).cuda().load_state_dict(sd)
    T = 48
    raw = synth.generate_channels('CDL-C', 16, 64, 16, 0.5, 5)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(2), 16, 64, 38), (0, 2, 1)))
    idx = np.arange(T) % 16
    ln = np.repeat(64.0 * 10 ** (-np.arange(-10, 20, 10) / 10.), 16)[:T]
    init = torch.randn(16, 64, 16, dtype=torch.complex64, generator=torch.Generator().manual_seed(4))

    def batches():
        out = []
        for part in np.array_split(np.arange(T), 2):
            a = AldBatch(net, H, Pm, idx[part], idx[part], ln[part], levels=[0, 700, 2310], steps_each=3, seed=21, traj_id=part, lanes=False)
            a.set_init(init[torch.from_numpy(idx[part])])
            a.synthesize_measurements()
            out.append(a)
        return out

    def result(alds):
        torch.cuda.synchronize()
        r = torch.cat([a.X for a in alds]).cpu().numpy(), torch.cat([a.nmse_log() for a in alds], dim=1).cpu().numpy()
        for a in alds:
            a.close()
        return r
    alone = batches()
    for a in alone:
        a.run(9)
    X0, L0 = result(alone)
    # (eager only: replaying such a plan as ONE hipGraph with two branches works -- profiles/r06_one_graph_two_branches.txt -- but
    # hipGraphLaunch of graphs with parallel branches crashed inside the runtime in one particular test order (csrc/api.hip: sbc_plan_run),
    # so the suite does not depend on it; `bench.py --pair-plan K --graph 1` runs it)
    for graph in (False, False):
        alds = batches()
        pair = AldPair(alds[0], alds[1], k_steps=4)                 # 9 steps = two lists of four + one of one
        pair.set_persistent_cus(128)
        pair.run(9, use_graph=graph)
        X1, L1 = result(alds)
        pair.close()
        assert np.array_equal(X0, X1) and np.array_equal(L0, L1), graph


@pytest.mark.parametrize('mode', ['bf16x3', 'f16x2'])
def test_folded_statistics_match_the_statistics_launches(weights64, mode):
    """``ScoreNet(fold_stats=True)`` (optional): the full-resolution InstanceNorm++ statistics are formed from the tile moments
    their producing launches write (SBC_EPI_MOMENTS_OUT) by statistics launches that read those moments instead of the tensors
    (SBC_PRO_NORM_MOMENTS).  Same mathematics in a different (fixed) summation order: golden-level agreement with the
    reference, and the properties the default path has -- reproducible bit for bit, independent of what else is in the batch."""
    import torch
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    g = load_golden('forward_64x16.npz')
    x = torch.from_numpy(g['x'])
    fold = ScoreNet(cfg, conv_mode=mode, fold_stats=True).cuda().load_state_dict(sd)
    base = ScoreNet(cfg, conv_mode=mode, fold_stats=False).cuda().load_state_dict(sd)
    ops = fold.score_plan(64, 16).ops
    n_fold = sum(1 for op in ops if op.kind == P.INORM_STATS and op.flags & P.PRO_NORM_MOMENTS)
    # f16x2: the last RCU block is a pair launch (no moments), and the two full-resolution ResidualBlocks are one launch each, which
    # forms the statistics of its intermediate itself
    assert n_fold == (10 if mode == 'bf16x3' else 7)
    # ... and the fourteen norms of the 16x4 and 8x2 levels have no statistics launch at all: SBC_PRO_NORM_SELF, or -- f16x2, where five of
    # those ResidualBlocks are RES blocks of SBC_OP_CHAIN records -- formed inside the chain launch
    n_res = sum(1 for op in ops if op.kind == P.CHAIN for b in op.blocks if b[0] == P.CHAIN_RES)
    # (f16x2 default plan: the normalizer's statistics are formed inside the END_CONV launch too -- one more SBC_PRO_NORM_SELF record, one
    # statistics record fewer)
    end_self = int(ops[-1].kind == P.END_CONV and bool(ops[-1].flags & P.PRO_NORM_SELF))
    assert end_self == (0 if mode == 'bf16x3' else 1)
    assert n_res == (0 if mode == 'bf16x3' else 5) and sum(1 for op in ops if op.flags & P.PRO_NORM_SELF) == 14 - 2 * n_res + end_self
    assert sum(1 for op in ops if op.kind == P.INORM_STATS) == (11 if mode == 'bf16x3' else 9 - end_self)
    for li, lev in enumerate([0, 1155, 2310]):
        labels = torch.full((x.shape[0],), lev)
        a = fold(x, labels)
        assert rel_err(a.cpu().numpy(), g['out'][li]) < 3e-6
        assert rel_err(a.cpu().numpy(), base(x, labels).cpu().numpy()) < 3e-6
        assert torch.equal(fold(x, labels), a)
        assert torch.equal(fold(x[1:3], labels[1:3]), a[1:3])              # batch independence


@pytest.mark.parametrize('nt,nr', [(16, 16), (32, 16), (16, 64), (64, 64), (128, 32)])
@pytest.mark.parametrize('mode', ['f16x2', 'f16w'])
def test_folded_statistics_at_other_array_sizes(weights64, mode, nt, nr):
    """The folded statistics (tile moments from producers, statistics computed by the consumers of images of at most 64 pixels)
    at array sizes other than the headline one: 8x8 / 4x4 / 2x2 levels (2, 8, 32 samples per 128-pixel tile), rows of 2 to 64
    pixels, 64-channel moments at 64x16 and 32x8.  Same network, statistics
    launches against folded statistics, batch of 5 (ragged last tiles at every level) and batch independence."""
    import torch
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    fold = ScoreNet(cfg, conv_mode=mode, fold_stats=True).cuda().load_state_dict(sd)
    base = ScoreNet(cfg, conv_mode=mode, fold_stats=False).cuda().load_state_dict(sd)
    ops = fold.score_plan(nt, nr).ops
    n_self = sum(1 for op in ops if op.flags & P.PRO_NORM_SELF)
    n_mom = sum(1 for op in ops if op.kind == P.INORM_STATS and op.flags & P.PRO_NORM_MOMENTS)
    assert n_self > 0 and (n_mom > 0 or nt * nr < 256)
    x = torch.randn(5, 2, nt, nr, generator=torch.Generator().manual_seed(nt * 1000 + nr))
    for lev in (0, 2310):
        labels = torch.full((5,), lev)
        a, b = fold(x, labels), base(x, labels)
        assert torch.isfinite(a).all()
        # (f16w rounds the transformed activations to fp16, 2^-11 each: a statistic one fp32 ulp off moves such roundings)
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < (3e-6 if mode == 'f16x2' else 4e-3)
        assert torch.equal(fold(x[2:4], labels[2:4]), a[2:4])


@pytest.mark.parametrize('mode', ['f16x2', 'f16w', 'bf16x3'])
def test_forward_does_not_depend_on_the_batch_size(weights64, mode):
    """A sample's score is the same bit pattern in a batch of 2400, 1200, 800 or 100: the kernel variants that large and small
    launches select (wave groups, blocks per phase, tile sizes, the pipelined pair kernel) compute identical sums, and the
    statistics folded into producers / consumers do not look at the batch.  Sub-batch streams rely on it."""
    import torch
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    net = ScoreNet(cfg, conv_mode=mode).cuda().load_state_dict(sd)
    x = torch.randn(2400, 2, 64, 16, generator=torch.Generator().manual_seed(5))
    lab = torch.full((2400,), 1155)
    a = net(x, lab)
    assert torch.isfinite(a).all()
    for n in (1200, 800, 100):
        assert torch.equal(net(x[:n], lab[:n]), a[:n]), n


def _run_cli_ranks(tmp_path, world, module, argv, port):
    """``python -m torch.distributed.run --nproc-per-node <world> -m score_based_channels_amd.<module> ...`` in its own directory
    (both ranks on this box's one GPU: gloo carries the collectives, RCCL does on a node with a GPU per rank)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = tmp_path / ('w%d' % world)
    out.mkdir(parents=True)
    env = dict(os.environ, PYTHONPATH=ROOT, SBC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(port + world), '-m', 'score_based_channels_amd.' + module] + argv
    r = subprocess.run(cmd, env=env, cwd=str(out), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def test_cli_two_ranks_equal_one_rank(tmp_path):
    """What shard.py promises: trajectories are independent and their noise is keyed by GLOBAL trajectory id, so
    ``torchrun --nproc-per-node 2 -m ...test_score`` (contiguous blocks per rank, one all_gather of the logs at the end) returns
    the one-rank NMSE log and final estimates BIT FOR BIT -- here with 5 channels x 17 SNR points = 85 trajectories (blocks of
    43 and 42), default conv_mode and streams.  The same for the tuner with a 3 x 1 grid x 17 x 3 = 153 trajectories (77 / 76)."""
    import torch
    argv = ['--synthetic', '--synthetic_weights', '2024', '--num_channels', '5', '--levels_stride', '1155', '--seed', '7',
            '--save_channels', '1', '--no_plot']
    res = {w: torch.load(_run_cli_ranks(tmp_path, w, 'test_score', argv, 29710) / 'results/score/train-CDL-C_test-CDL-C/results.pt',
                         weights_only=False) for w in (1, 2)}
    assert res[1]['nmse_log'].shape == (1, 1, 17, 9, 5) and np.isfinite(res[1]['nmse_log']).all()
    assert np.array_equal(res[1]['nmse_log'], res[2]['nmse_log'])
    assert np.array_equal(res[1]['saved_H'], res[2]['saved_H'])
    assert res[1]['f16x2_fallback'] == [] and res[2]['f16x2_fallback'] == []
    targv = ['--synthetic', '--synthetic_weights', '2024', '--num_channels', '3', '--levels_stride', '1155', '--seed', '8', '--no_plot',
             '--alpha_step_range', '3e-11', '6e-11', '1e-10', '--beta_noise_range', '0.01']
    tun = {w: torch.load(_run_cli_ranks(tmp_path / 'tune', w, 'tune_hparams_score', targv, 29720) / 'results/score/CDL-C-hyperparameters.pt',
                         weights_only=False) for w in (1, 2)}
    assert tun[1]['nmse_log'].shape == (3, 1, 17, 9, 3)
    assert np.array_equal(tun[1]['nmse_log'], tun[2]['nmse_log'])
    assert np.array_equal(tun[1]['best_alpha_snr'], tun[2]['best_alpha_snr'])


def test_cli_several_test_profiles_are_one_sharded_list(tmp_path, monkeypatch):
    """BASELINE configs[3] as worded: ``--test CDL-A CDL-B CDL-C CDL-D`` flattens (profile x SNR x channel) into ONE trajectory list
    (4 x 17 x 3 = 204 trajectories) that shard.my_block splits over the ranks.  Every profile's results.pt equals the one its own
    single-profile invocation writes (normalisation constants from the train profile, pilots, noise keys: test_score.py:15-22,91-101)
    bit for bit, and two ranks (blocks of 102 cut through profile CDL-B / CDL-C) equal one rank bit for bit."""
    import torch
    from score_based_channels_amd import test_score
    profiles = ['CDL-A', 'CDL-B', 'CDL-C', 'CDL-D']
    base = ['--synthetic', '--synthetic_weights', '2024', '--num_channels', '3', '--levels_stride', '1155', '--seed', '11',
            '--save_channels', '1', '--no_plot']
    monkeypatch.chdir(tmp_path)
    single = {}
    for p in profiles:
        test_score.main(base + ['--test', p])
        single[p] = torch.load(tmp_path / ('results/score/train-CDL-C_test-%s/results.pt' % p), weights_only=False)
    multi = test_score.main(base + ['--test'] + profiles)
    assert sorted(multi) == sorted(profiles)
    for p in profiles:
        r = torch.load(tmp_path / ('results/score/train-CDL-C_test-%s/results.pt' % p), weights_only=False)
        assert r['nmse_log'].shape == (1, 1, 17, 9, 3) and np.array_equal(r['nmse_log'], multi[p][0])
        assert np.array_equal(r['nmse_log'], single[p]['nmse_log']), p
        assert np.array_equal(r['saved_H'], single[p]['saved_H']), p
        assert r['val_config'].data.channel == p
    assert not np.array_equal(single['CDL-A']['nmse_log'], single['CDL-D']['nmse_log'])
    out = _run_cli_ranks(tmp_path / 'ranks', 2, 'test_score', base + ['--test'] + profiles, 29740)
    for p in profiles:
        r2 = torch.load(out / ('results/score/train-CDL-C_test-%s/results.pt' % p), weights_only=False)
        assert np.array_equal(r2['nmse_log'], single[p]['nmse_log']) and np.array_equal(r2['saved_H'], single[p]['saved_H']), p


def test_cli_rank_failure_ends_every_rank_quickly(tmp_path):
    """A rank that raises before the final gather must not leave the others in the all_gather until the 300 s collective timeout
    (shard.run_guarded / check_peers): tests/failing_rank_cli.py wraps the CLI so that rank 1 raises inside its run (no hook in the
    product for this); the job exits non-zero within seconds."""
    import os
    import subprocess
    import sys
    import time
    from conftest import ROOT
    env = dict(os.environ, PYTHONPATH=ROOT, SBC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29761', os.path.join(ROOT, 'tests', 'failing_rank_cli.py'), '1', '--synthetic', '--synthetic_weights', '2024',
           '--num_channels', '2', '--levels_stride', '2310', '--seed', '1', '--no_plot']
    t0 = time.time()
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert time.time() - t0 < 150, 'the healthy rank waited for the collective timeout'
    assert 'fails before the gather, on request' in r.stderr and 'PeerFailure' in r.stderr


@pytest.mark.parametrize('mode', ['f16x2', 'bf16x3'])
def test_module_call_propagates_nan_like_the_reference(weights64, mode):
    """F.instance_norm / F.elu / MaxPool2d of the reference propagate a NaN in a sample's input through that sample's whole
    score; the fused kernels' max-based ELU and pooling would swallow it, so the module call restores the reference's
    behaviour for samples whose input is not finite.  The other samples are untouched, bit for bit."""
    import torch
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    net = ScoreNet(cfg, conv_mode=mode).cuda().load_state_dict(sd)
    x = torch.randn(3, 2, 64, 16, generator=torch.Generator().manual_seed(2))
    lab = torch.full((3,), 100)
    clean = net(x, lab)
    xn = x.clone()
    xn[1, 0, 17, 5] = float('nan')
    out = net(xn, lab)
    assert torch.isnan(out[1]).all()
    assert torch.equal(out[0], clean[0]) and torch.equal(out[2], clean[2])


def test_f16x2_scales_do_not_depend_on_which_size_is_bound_first(weights64):
    """one weight buffer serves every array size; a layer that runs fused (direct form) at 64 x 16 runs as an
    unfused Winograd launch at other sizes, and the calibration used to write only the form the first-bound plan reads.  Now the pass
    always runs at the configuration's size and writes every form: binding 128 x 8 first or 64 x 16 first gives the same numbers."""
    import torch
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    g = torch.Generator().manual_seed(9)
    xa, xb = torch.randn(2, 2, 64, 16, generator=g), torch.randn(2, 2, 128, 8, generator=g)
    lab = torch.full((2,), 7)
    n1 = ScoreNet(cfg, conv_mode='f16x2').cuda().load_state_dict(sd)
    a1, b1 = n1(xa, lab), n1(xb, lab)
    n2 = ScoreNet(cfg, conv_mode='f16x2').cuda().load_state_dict(sd)
    b2, a2 = n2(xb, lab), n2(xa, lab)
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    assert n1.range_fallbacks == 0 and n2.range_fallbacks == 0
    # both packed forms of a fused layer carry the same calibrated scale (first trailer word), and it is not the initial 1.0
    key = 'refine5.output_convs.1_1_conv.weight'
    w = sd[key]
    for form, taps in (('#split', 9), ('#winograd_split', 16)):
        off = n1._woff[key + form] + taps * w.shape[0] * w.shape[1]             # two fp16 terms per weight = one float32 word each
        tr = n1._wdev[off:off + 4].cpu().numpy()
        if form == '#split':
            first = tr
        assert tr[0] == first[0] and tr[0] != 1.0, (form, tr)
    # the two layers of a CONV_DOWN record (res2.0 / res3.0: pooled 3x3 + pooled 1x1 shortcut) carry their scale in the
    # pooled forms the fused launch reads AND in the unpooled forms the unfused launches of a non-down-fusable size (128 x 8) read
    for blk in ('res2.0.', 'res3.0.'):
        for key, forms in ((blk + 'conv2.conv.weight', (('#pool', 16), ('#split', 9), ('#winograd_split', 16))),
                           (blk + 'shortcut.conv.weight', (('#pool', 4), ('#split', 1)))):
            w = sd[key]
            scales = []
            for form, taps in forms:
                off = n1._woff[key + form] + taps * w.shape[0] * w.shape[1]
                scales.append(float(n1._wdev[off].item()))
                assert float(n2._wdev[n2._woff[key + form] + taps * w.shape[0] * w.shape[1]].item()) == scales[-1], (key, form)
            assert len(set(scales)) == 1 and scales[0] != 1.0, (key, scales)


def test_persistent_grid_width_is_a_property_of_the_plan(weights64):
    """sbc_set_persistent_cus was process-global state toggled by driver.run_concurrently.  The width is now a field of
    the plan (sbc_plan_set_persistent_cus): two batches driven from two host threads with different widths give the results of a
    default run bit for bit, and nothing leaks into the process default."""
    import threading
    import torch
    from score_based_channels_amd import synth
    from score_based_channels_amd.ald import AldBatch
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    net = ScoreNet(cfg).cuda().load_state_dict(sd)
    raw = synth.generate_channels('CDL-C', 16, 64, 16, 0.5, 3)
    H = np.conj(np.transpose(raw / np.std(raw), (0, 2, 1))).astype(np.complex64)
    Pm = np.conj(np.transpose(synth.qpsk_pilots(np.random.default_rng(1), 16, 64, 38), (0, 2, 1)))
    init = torch.randn(16, 64, 16, dtype=torch.complex64, generator=torch.Generator().manual_seed(4))

    def batch():
        a = AldBatch(net, H, Pm, np.arange(16), np.arange(16), 64.0, levels=[0, 1155], steps_each=2, seed=5)
        a.set_init(init)
        a.synthesize_measurements()
        return a
    ref = batch()
    ref.run()
    torch.cuda.synchronize()
    want = ref.nmse_log().cpu().numpy()
    outs = {}

    def work(width):
        a = batch()
        a.set_persistent_cus(width)
        with torch.cuda.stream(torch.cuda.Stream()):
            a.run()
        torch.cuda.synchronize()
        outs[width] = a.nmse_log().cpu().numpy()
    ths = [threading.Thread(target=work, args=(w,)) for w in (32, 200)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert np.array_equal(outs[32], want) and np.array_equal(outs[200], want)


@pytest.fixture(scope='module', params=[300, 4000])
def trained_state(request, tmp_path_factory):
    """The repository's own TRAINED checkpoints (tests/trained_weights.py: 300 and -- round 6 -- 4000 optimiser steps of the package's
    trainer), re-created here on the GPU -- a training step is bit-reproducible -- and checked against the digests in the golden."""
    import warnings
    import trained_weights as TW
    cfg, sd = TW.train_checkpoint(tmp_path_factory.mktemp('trained%d' % request.param), request.param)
    g = load_golden('trained_%dsteps.npz' % request.param)
    keys, dig, crc = TW.state_digest(sd)
    assert keys == [str(k) for k in g['weight_keys']]
    identical = bool(np.array_equal(crc, g['weight_crc']))
    if not identical:
        # a changed training kernel moves the weights by rounding noise: the parity assertions below still hold the HIP path to the
        # reference on ITS weights within tolerance, but say so
        worst = float(np.max(np.abs(dig - g['weight_digest']) / (np.abs(g['weight_digest']) + 1e-12)))
        warnings.warn('re-trained checkpoint differs from the one the golden was made with (worst digest deviation %.2e)' % worst)
        assert worst < 1e-3
    return cfg, sd, g, identical


@pytest.mark.parametrize('mode', ['default', 'bf16x3'])
def test_trained_checkpoint_matches_reference_golden(trained_state, mode):
    """every other golden uses seed-derived random-init weights; the f16x2 calibration (per-layer activation scales
    from a fixed input) had never met other weight statistics.  Forward at three levels and the 93-step truncated schedule on a
    checkpoint that has been trained, in the default mode (f16x2 + fused plans, calibrated) with NO range fallback, at the
    north_star tolerances (forward 2e-5 norm-wise, NMSE log 1e-5 at every step); bf16x3 beside it."""
    import torch
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd, g, _ = trained_state
    net = (ScoreNet(cfg) if mode == 'default' else ScoreNet(cfg, conv_mode=mode)).cuda().load_state_dict(sd).eval()
    assert mode != 'default' or (net.conv_mode == 'f16x2' and net.fuse_pairs and net.fuse_res)
    x = torch.from_numpy(g['x']).cuda()
    for i, lv in enumerate(g['levels']):
        out = net(x, torch.full((x.shape[0],), int(lv), device='cuda')).cpu().numpy()
        assert rel_err(out, g['out'][i]) < 2e-5, (mode, int(lv), rel_err(out, g['out'][i]))
        assert rel_err_elementwise(out, g['out'][i], floor=0.02) < 1e-4
    assert net.range_fallbacks == 0
    from score_based_channels_amd import _lib
    gg = dict(g, levels=g['ald_levels'], steps_each=3, alpha_step=3e-11, beta_noise=0.01)
    _lib.range_flag(True, net.device)
    Y, X, log = _run_golden_ald(net, gg)
    assert _lib.range_flag(True, net.device) == 0          # the calibrated scales hold on trained weights: no bf16x3 re-run needed
    assert rel_err(Y, g['Y']) < 1e-6
    assert np.max(np.abs(log / g['nmse_log'] - 1)) < NMSE_RTOL, np.max(np.abs(log / g['nmse_log'] - 1))
    assert rel_err(X, g['X_final']) < 1e-5
    if mode == 'default':
        # the calibrated activation scales of this checkpoint (first trailer word of every direct f16x2 form): how far training moved
        # the window -- printed with -s, recorded in profiles/r06_trained_checkpoints.txt
        sc = []
        for k, off in net._woff.items():
            if k.endswith('#split'):
                w = sd[k[:-len('#split')]]
                sc.append(float(net._wdev[off + w.shape[2] * w.shape[3] * w.shape[0] * w.shape[1]].item()))
        lg = np.log2(np.asarray(sc))
        print('f16x2 act_scale exponents over %d layers: min %d median %d max %d' % (len(sc), lg.min(), np.median(lg), lg.max()))
