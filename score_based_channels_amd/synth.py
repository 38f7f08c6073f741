"""Synthetic narrowband CDL-like MIMO channels and QPSK pilots.

The reference draws its channels offline with MATLAB's 5G Toolbox ``nrCDLChannel``
(``matlab/genChannels.m:5-16,36-56``, parameters ``matlab/generate_data.m:9-21``) and the
resulting ``.mat`` files are not shipped.  This generator stands in for them: a cluster/ray
sum per 3GPP TR 38.901 section 7.7.1 evaluated at one subcarrier of one symbol (the loader keeps
``output_h[:, 0]`` only, ``loaders.py:32-33``), for the vertical ``[N,1,1,1,1]`` uniform
linear arrays the reference configures (``genChannels.m:13-16``).

Cluster tables are the CDL-A..D rows of TR 38.901 Table 7.7.1-1..4 as recalled by the author
(no network access to re-check every digit); they only shape the angular statistics of
*synthetic* test data and have no influence on parity, which is pinned on identical inputs.
"""
import numpy as np

# per profile: per-cluster (power dB, ZOD deg, ZOA deg), cluster zenith spreads (cZSD, cZSA), LOS K (dB) or None
_RAY_OFFSETS = np.array([0.0447, 0.1413, 0.2492, 0.3715, 0.5129, 0.6797, 0.8844, 1.1481, 1.5195, 2.1551])
_RAY_OFFSETS = np.concatenate([_RAY_OFFSETS, -_RAY_OFFSETS])

_CDL = {
    'CDL-A': dict(czsd=3.0, czsa=7.0, los=None, clusters=[
        (-13.4, 98.5, 85.5), (0.0, 89.9, 78.2), (-2.2, 89.9, 78.2), (-4.0, 89.9, 78.2),
        (-6.0, 104.2, 89.2), (-8.2, 104.2, 89.2), (-9.9, 104.2, 89.2), (-10.5, 99.4, 96.8),
        (-7.5, 100.8, 86.7), (-15.9, 98.8, 94.6), (-6.6, 100.7, 93.6), (-16.7, 100.6, 106.2),
        (-12.4, 98.3, 95.6), (-15.2, 97.1, 91.1), (-10.8, 96.8, 81.3), (-11.3, 98.7, 93.2),
        (-12.7, 102.0, 110.5), (-16.2, 98.8, 98.4), (-18.3, 98.0, 102.3), (-18.9, 95.0, 81.4),
        (-16.6, 100.5, 100.2), (-19.9, 96.4, 100.8), (-29.7, 105.6, 72.1)]),
    'CDL-B': dict(czsd=7.0, czsa=7.0, los=None, clusters=[
        (0.0, 105.8, 77.9), (-2.2, 95.5, 86.1), (-4.0, 95.5, 86.1), (-3.2, 95.5, 86.1),
        (-9.8, 103.1, 94.4), (-1.2, 104.3, 91.4), (-3.4, 104.3, 91.4), (-5.2, 104.3, 91.4),
        (-7.6, 93.8, 101.6), (-3.0, 104.2, 68.1), (-8.9, 94.9, 97.0), (-9.0, 93.1, 84.2),
        (-4.8, 92.2, 101.3), (-5.7, 106.7, 96.8), (-7.5, 93.0, 92.9), (-1.9, 92.9, 88.6),
        (-7.6, 92.9, 88.6), (-12.2, 105.2, 95.6), (-9.8, 107.8, 84.7), (-11.4, 93.7, 81.7),
        (-14.9, 94.2, 96.1), (-9.2, 92.7, 89.4), (-11.3, 92.9, 79.4)]),
    'CDL-C': dict(czsd=3.0, czsa=7.0, los=None, clusters=[
        (-4.4, 97.2, 87.6), (-1.2, 98.6, 72.1), (-3.5, 98.6, 72.1), (-5.2, 98.6, 72.1),
        (-2.5, 100.6, 70.1), (0.0, 99.2, 75.3), (-2.2, 99.2, 75.3), (-3.9, 99.2, 75.3),
        (-7.4, 105.2, 67.4), (-7.1, 95.3, 63.8), (-10.7, 106.1, 71.4), (-11.1, 93.5, 60.5),
        (-5.1, 103.7, 90.6), (-6.8, 104.2, 60.1), (-8.7, 93.0, 61.0), (-13.2, 104.2, 100.7),
        (-13.9, 94.9, 62.3), (-13.9, 93.1, 66.7), (-15.8, 92.2, 52.9), (-17.1, 106.7, 61.8),
        (-16.0, 93.0, 51.9), (-15.7, 92.9, 61.7), (-21.6, 105.2, 58.0), (-22.8, 107.8, 57.0)]),
    'CDL-D': dict(czsd=3.0, czsa=3.0, los=13.3, clusters=[
        (-13.5, 98.5, 81.5), (-18.8, 85.5, 86.9), (-21.0, 85.5, 86.9), (-22.8, 85.5, 86.9),
        (-17.9, 100.1, 104.8), (-20.1, 100.1, 104.8), (-21.9, 100.1, 104.8), (-22.9, 98.6, 94.7),
        (-27.8, 91.7, 108.1), (-23.6, 98.2, 100.0), (-24.8, 99.6, 104.0), (-30.0, 93.6, 81.2),
        (-27.7, 91.3, 78.0)]),
}
PROFILES = tuple(_CDL) + ('ULA',)


def _steer(n, spacing, zenith_deg):
    """Vertical ULA response ``exp(j 2 pi d n cos(theta_z))``, shape ``[..., n]``."""
    k = np.arange(n)
    return np.exp(2j * np.pi * spacing * np.cos(np.deg2rad(zenith_deg))[..., None] * k)


def generate_channels(profile, num, nt=64, nr=16, spacing=0.5, seed=9999):
    """Complex ``[num, Nr, Nt]`` channels (the layout of ``output_h[:, 0]``, loaders.py:33).

    ``profile``: 'CDL-A'..'CDL-D' (cluster model) or 'ULA' (a few random plane waves per sample,
    for large-array stress cases).  Deterministic in ``(profile, num, nt, nr, spacing, seed)``.
    """
    rng = np.random.default_rng([seed, nt, nr, int(round(spacing * 100)),
                                 PROFILES.index(profile)])
    H = np.zeros((num, nr, nt), np.complex128)
    if profile == 'ULA':
        for _ in range(6):
            g = (rng.standard_normal(num) + 1j * rng.standard_normal(num)) / np.sqrt(12)
            zod, zoa = rng.uniform(30, 150, num), rng.uniform(30, 150, num)
            H += g[:, None, None] * _steer(nr, spacing, zoa)[:, :, None] * _steer(nt, spacing, zod)[:, None, :]
        return H.astype(np.complex64)
    tab = _CDL[profile]
    p = np.array([c[0] for c in tab['clusters']])
    zod = np.array([c[1] for c in tab['clusters']])
    zoa = np.array([c[2] for c in tab['clusters']])
    lin = 10 ** (p / 10)
    if tab['los'] is not None:           # first row is the LOS ray; rescale so K-factor holds
        k_lin = 10 ** (tab['los'] / 10)
        nlos = lin / lin.sum() / (1 + k_lin)
        los_amp = np.sqrt(k_lin / (1 + k_lin))
    else:
        nlos, los_amp = lin / lin.sum(), 0.0
    ray_zod = zod[:, None] + tab['czsd'] * _RAY_OFFSETS[None, :]      # [clusters, 20]
    ray_zoa = zoa[:, None] + tab['czsa'] * _RAY_OFFSETS[None, :]
    for i in range(num):
        perm = np.stack([rng.permutation(20) for _ in range(len(p))])            # random ray coupling
        phase = np.exp(2j * np.pi * rng.random((len(p), 20)))
        a_rx = _steer(nr, spacing, np.take_along_axis(ray_zoa, perm, 1))         # [c, 20, nr]
        a_tx = _steer(nt, spacing, ray_zod)                                       # [c, 20, nt]
        amp = np.sqrt(nlos / 20)[:, None] * phase
        H[i] = np.einsum('cm,cmr,cmt->rt', amp, a_rx, a_tx)
        if los_amp:
            ph = np.exp(2j * np.pi * rng.random())
            H[i] += los_amp * ph * np.outer(_steer(nr, spacing, zoa[:1])[0], _steer(nt, spacing, zod[:1])[0])
    return H.astype(np.complex64)


def generate_output_h(profile, num, nt=64, nr=16, spacing=0.5, seed=9999, n_sym=10):
    """Array shaped like the ``output_h`` variable of the reference ``.mat`` files:
    ``[num, n_sym, Nr, Nt]`` (``genChannels.m:27``); the loader only reads symbol 0, the other
    symbols are filled with slowly phase-rotated copies."""
    h0 = generate_channels(profile, num, nt, nr, spacing, seed)
    rot = np.exp(2j * np.pi * 0.01 * np.arange(n_sym))[None, :, None, None]
    return (h0[:, None] * rot).astype(np.complex64)


def qpsk_pilots(rng, num, nt, num_pilots):
    """QPSK pilots ``(+-1 +-j)/sqrt(2)`` of shape ``[num, Nt, Np]`` (loaders.py:52-55), drawn
    from a numpy ``Generator`` (real signs first, then imaginary signs)."""
    re = 2 * rng.integers(0, 2, size=(num, nt, num_pilots)) - 1
    im = 2 * rng.integers(0, 2, size=(num, nt, num_pilots)) - 1
    return ((re + 1j * im) / np.sqrt(2)).astype(np.complex64)
