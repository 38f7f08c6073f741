"""ORACLE -- test infrastructure, NOT product code.

CPU (numpy) restatement of the annealed-Langevin channel-estimation loop of the reference,
``src/score_based_channels/test_score.py:118-171`` (the same loop is copied into
``tune_hparams_score.py:100-148``), of the measurement synthesis ``test_score.py:122-124``,
of the result post-processing ``test_score.py:174-175`` / ``tune_hparams_score.py:151-162``
and of the parts of ``loaders.Channels`` (``loaders.py:11-107``) that feed the loop.

The reference scripts cannot be imported (module-level argparse, unconditional ``.cuda()``,
missing checkpoint / data blobs), so this file follows them line by line and
is pinned by ``tests/gen_golden.py``, which runs the *reference's own* ``NCSNv2Deepest``
(imported from ``/root/reference`` in the build container) inside this loop and commits the
results under ``tests/golden/``.  The reference never seeds its RNG; here every Gaussian draw
is an explicit argument so the HIP path and the oracle consume identical noise.

Scalar semantics mirrored from PyTorch: a python/numpy float64 scalar multiplied into a
complex64 tensor is rounded to float32 first (type promotion keeps the tensor dtype).
"""
import numpy as np

F32 = np.float32
C64 = np.complex64


def snr_to_noise(snr_db, nt):
    """``noise_range = 10 ** (-snr_range / 10.) * config.data.image_size[1]`` (test_score.py:75)."""
    return 10 ** (-np.asarray(snr_db, np.float64) / 10.) * nt


def complex_normal(rng, shape):
    """Stand-in for ``torch.randn_like`` on a complex64 tensor: real and imaginary parts are
    each N(0, 1/2) (test_score.py:115,124,161).  Drawn as interleaved (re, im) float32 pairs from
    a numpy ``Generator`` (platform-stable ziggurat), scaled by float32 sqrt(0.5)."""
    z = rng.standard_normal(tuple(shape) + (2,), dtype=F32) * F32(np.sqrt(0.5))
    return np.ascontiguousarray(z).view(C64)[..., 0]


def make_measurements(P, H, local_noise, noise):
    """``val_Y = P H + sqrt(local_noise) * n`` (test_score.py:122-124).
    P ``[B, Np, Nt]`` (already conj-transposed pilots, test_score.py:111), H ``[B, Nt, Nr]``."""
    Y = np.matmul(P.astype(C64), H.astype(C64))
    return (Y + F32(np.sqrt(local_noise)) * noise.astype(C64)).astype(C64)


def step_scalars(sigma_f32, sigma_end, alpha_step, beta_noise, local_noise):
    """Per-level scalars of test_score.py:137-165, float64 like the python code that computes
    them; returns what each is rounded to when it meets a complex64 tensor."""
    current_sigma = float(sigma_f32)                               # .item() of a float32
    alpha = alpha_step * (current_sigma / sigma_end) ** 2          # :143-144
    noise_scale = np.sqrt(2 * alpha * beta_noise)                  # :160
    dc_div = local_noise / 2. + current_sigma ** 2                 # :165
    return F32(alpha), F32(dc_div), F32(noise_scale)


def langevin_step(current, score, P, Y, alpha32, dc_div32, noise_scale32, noise, dc_boost32=None):
    """One update of test_score.py:156-165.  All tensors complex64.  ``dc_boost32``: the factor of
    test_mmse.py:231-233, ``alpha * (score - dc_boost * meas_grad / (...))`` -- multiplied into the gradient first."""
    P_h = np.conj(np.transpose(P, (0, 2, 1)))
    meas_grad = np.matmul(P_h, np.matmul(P, current) - Y)          # :157-158
    if dc_boost32 is not None:
        meas_grad = F32(dc_boost32) * meas_grad
    grad_noise = noise_scale32 * noise                             # :160-161
    return (current + alpha32 * (score - meas_grad / dc_div32) + grad_noise).astype(C64)


def nmse(current, oracle):
    """test_score.py:168-170: sum |X - H|^2 / sum |H|^2 per sample, float32."""
    num = np.sum(np.square(np.abs(current - oracle)), axis=(-1, -2), dtype=F32)
    den = np.sum(np.square(np.abs(oracle)), axis=(-1, -2), dtype=F32)
    return (num / den).astype(F32)


def ald_run(score_fn, sigmas, sigma_end, P, Y, H_true, init, step_noise, local_noise,
            alpha_step=3e-11, beta_noise=0.01, steps_each=3, levels=None):
    """The loop of test_score.py:126-171 for ONE SNR point.

    score_fn(x_real ``[B,2,Nt,Nr]`` float32, labels ``[B]`` int64) -> float32 ``[B,2,Nt,Nr]``
    (``diffuser(current_real, labels)``, test_score.py:149-151);
    ``step_noise(k)`` returns the complex64 ``[B,Nt,Nr]`` CN(0,1) draw of Langevin step ``k``;
    ``levels``: noise-level indices to walk (default ``range(len(sigmas))``, the full schedule).
    Returns (final estimate, nmse_log ``[len(levels)*steps_each, B]`` float32).
    """
    levels = range(len(sigmas)) if levels is None else list(levels)
    current = init.astype(C64).copy()                              # :126
    B = current.shape[0]
    log = np.zeros((len(levels) * steps_each, B), F32)
    k = 0
    for step_idx in levels:
        a32, d32, n32 = step_scalars(sigmas[step_idx], sigma_end, alpha_step, beta_noise,
                                     local_noise)
        labels = np.full((B,), step_idx, np.int64)                 # :139-140
        for _ in range(steps_each):
            current_real = np.stack((current.real, current.imag), axis=1).astype(F32)   # :149
            score_real = score_fn(current_real, labels)                                   # :151
            score = (score_real[:, 0] + 1j * score_real[:, 1]).astype(C64)                # :153-154
            current = langevin_step(current, score, P, Y, a32, d32, n32, step_noise(k))
            log[k] = nmse(current, H_true)                          # :168-170
            k += 1
    return current, log


def mmse_run(score_fn, sigmas, sigma_end, P, Y, H_true, init, step_noise, local_noise, step_size, noise_boost,
             target_stop, mmse_avg, dc_boost=1.0, steps_each=3, levels=None):
    """Posterior sampling of ONE SNR point, ``src/score_based_channels/test_mmse.py:182-262``: every kept sample is
    repeated ``mmse_avg`` times (chains share the sample's measurement, :185-193), walked with that SNR's tuned
    ``step_size`` / ``noise_boost`` (:171-172,216-228), the data-consistency gradient scaled by ``dc_boost`` (:231-233), and
    stopped after step index ``target_stop`` (:246-250).

    P ``[kept, Np, Nt]``, Y ``[kept, Np, Nr]``, H_true ``[kept, Nt, Nr]``; ``init`` ``[kept * mmse_avg, Nt, Nr]`` (chain
    ``c`` of sample ``s`` is row ``s * mmse_avg + c``); ``step_noise(k)`` the CN(0,1) draw of step ``k`` for all chains.
    Returns (``oracle_log`` ``[total_steps, kept, mmse_avg]`` float32, zero beyond the stop; final estimates
    ``[kept, mmse_avg, Nt, Nr]``)."""
    levels = range(len(sigmas)) if levels is None else list(levels)
    kept = H_true.shape[0]
    gY, gP, gH = (np.repeat(a, mmse_avg, axis=0) for a in (Y, P, H_true))      # tile per sample (:185-193)
    current = init.astype(C64).copy()
    log = np.zeros((len(levels) * steps_each, kept, mmse_avg), F32)
    k = 0
    for step_idx in levels:
        a32, d32, n32 = step_scalars(sigmas[step_idx], sigma_end, step_size, noise_boost, local_noise)
        labels = np.full((current.shape[0],), step_idx, np.int64)
        for _ in range(steps_each):
            current_real = np.stack((current.real, current.imag), axis=1).astype(F32)
            score_real = score_fn(current_real, labels)
            score = (score_real[:, 0] + 1j * score_real[:, 1]).astype(C64)
            current = langevin_step(current, score, gP, gY, a32, d32, n32, step_noise(k), dc_boost32=dc_boost)
            log[k] = nmse(current, gH).reshape(kept, mmse_avg)                   # :238-244
            if k == target_stop:                                                 # :246-250
                return log, current.reshape((kept, mmse_avg) + H_true.shape[1:])
            k += 1
    return log, current.reshape((kept, mmse_avg) + H_true.shape[1:])


def reduce_nmse(nmse_log):
    """test_score.py:174-175: mean over channels, then min over steps."""
    avg = np.mean(nmse_log, axis=-1)
    return avg, np.min(avg, axis=-1)


def tune_select(best_nmse, alpha_step_range, beta_noise_range):
    """tune_hparams_score.py:155-162: per-SNR argmin over the flattened (alpha, beta) grid."""
    best_alpha, best_beta = [], []
    for snr_idx in range(best_nmse.shape[-1]):
        local = best_nmse[..., snr_idx].flatten()
        ai, bi = np.unravel_index(np.argmin(local), (len(alpha_step_range), len(beta_noise_range)))
        best_alpha.append(alpha_step_range[ai])
        best_beta.append(beta_noise_range[bi])
    return best_alpha, best_beta


# ----------------------------------------------------------------------------- loader pieces
def channels_dataset(output_h, image_size1, num_pilots, norm, legacy_seed=None):
    """What ``Channels.__init__`` derives from one ``.mat`` file (loaders.py:29-58).

    output_h: complex ``[N, n_sym, Nr, Nt]`` as ``hdf5storage.loadmat`` returns it.  Keeps the
    first subcarrier, applies the normalisation rule, draws QPSK pilots from numpy's *legacy
    global* RNG (two ``binomial`` calls: real then imaginary).  Returns (channels ``[N,Nr,Nt]``,
    mean, std, pilots ``[N, Nt, Np]`` complex128)."""
    channels = np.asarray(output_h, dtype=np.complex64)[:, 0]                 # :30-33
    channels = np.reshape(np.asarray([channels]), (-1, channels.shape[-2], channels.shape[-1]))
    if type(norm) == list:
        mean, std = norm[0], norm[1]
    elif norm == 'entrywise':
        mean, std = np.mean(channels, axis=0), np.std(channels, axis=0)
    elif norm == 'global':
        mean, std = 0., np.std(channels)
    if legacy_seed is not None:
        np.random.seed(legacy_seed)
    size = (channels.shape[0], image_size1, num_pilots)
    pilots = 1 / np.sqrt(2) * (2 * np.random.binomial(1, 0.5, size=size) - 1 +
                               1j * (2 * np.random.binomial(1, 0.5, size=size) - 1))   # :52-55
    return channels, mean, std, pilots


def channels_item(channels, mean, std, pilots, idx):
    """The tensors of ``Channels.__getitem__`` the loop consumes (loaders.py:67-106)."""
    H = channels[idx]
    Hn = (H - mean) / std
    H_herm_norm = np.conj(np.transpose(Hn))
    return {'H_herm': np.stack((np.real(H_herm_norm), np.imag(H_herm_norm)), 0).astype(F32),
            'H': np.stack((np.real(Hn), np.imag(Hn)), 0).astype(F32),
            'P': pilots[idx].astype(C64)}


# ----------------------------------------------------------------------------- the production noise stream, restated
# The product's default noise is drawn inside the Langevin / measurement kernels (csrc/philox.h): Philox4x32-10 (Salmon,
# Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 ships known-answer vectors, pinned in
# tests/test_oracle_golden.py) + Box-Muller.  The reference draws torch.randn_like (test_score.py:115,124,160-161) from an
# unseeded generator, so there is no reference stream to match -- this restatement lets a run with in-kernel noise be replayed
# on the host and held to the oracle loop number for number.
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(counter, key, rounds=10):
    """Philox4x32-``rounds`` of uint32 counters ``[..., 4]`` under keys ``[..., 2]`` (broadcast): uint32 ``[..., 4]``."""
    c = np.array(np.broadcast_to(np.asarray(counter, np.uint32), np.broadcast_shapes(np.shape(counter), np.shape(key)[:-1] + (4,))))
    k = np.array(np.broadcast_to(np.asarray(key, np.uint32), c.shape[:-1] + (2,)))
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0].copy(), k[..., 1].copy()
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over='ignore'):
        for _ in range(rounds):
            p0 = c0.astype(np.uint64) * _PHILOX_M0
            p1 = c2.astype(np.uint64) * _PHILOX_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = k0 + _PHILOX_W0
            k1 = k1 + _PHILOX_W1
    return np.stack((c0, c1, c2, c3), axis=-1)


def device_complex_normal(seed, traj, step, n_elem):
    """The CN(0,1) draws csrc/philox.h::complex_normal gives elements ``0..n_elem-1`` of trajectory ``traj`` at Langevin step
    ``step`` (``-1``: the measurement noise of SBC_OP_MEASURE): elements 2q, 2q+1 share the block of counter
    (q, step, traj_lo, traj_hi) under key (seed_lo, seed_hi); words (x, y) / (z, w) -> u = ((w >> 8) + 0.5) / 2^24 ->
    sqrt(-ln u1) (cos 2 pi u2, sin 2 pi u2).  float32 arithmetic like the kernel (libm vs the GPU's math library: ~1e-6)."""
    nq = (int(n_elem) + 1) // 2
    traj, seed = int(traj) & (2 ** 64 - 1), int(seed) & (2 ** 64 - 1)
    ctr = np.zeros((nq, 4), np.uint32)
    ctr[:, 0] = np.arange(nq, dtype=np.uint32)
    ctr[:, 1] = np.uint32(int(step) & 0xFFFFFFFF)
    ctr[:, 2] = np.uint32(traj & 0xFFFFFFFF)
    ctr[:, 3] = np.uint32(traj >> 32)
    r = philox4x32(ctr, np.array([seed & 0xFFFFFFFF, seed >> 32], np.uint32)).reshape(-1, 2)      # [(q, half), (a, b)]
    u = ((r >> np.uint32(8)).astype(F32) + F32(0.5)) * F32(1.0 / 16777216.0)
    rad = np.sqrt(-np.log(u[:, 0], dtype=F32), dtype=F32)
    ang = (F32(6.283185307179586) * u[:, 1]).astype(F32)
    z = (rad * np.cos(ang, dtype=F32)).astype(F32) + 1j * (rad * np.sin(ang, dtype=F32)).astype(F32)
    return z.astype(C64)[:n_elem]
