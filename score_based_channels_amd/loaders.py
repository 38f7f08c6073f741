"""``Channels`` dataset: counterpart of ``src/score_based_channels/loaders.py:8-107``.

Same constructor, attributes and per-item dictionary as the reference so the inference scripts can use it
unchanged: file name pattern ``./data/<profile>_Nt64_Nr16_ULA<spacing>_seed<seed>.mat`` (loaders.py:23-24),
variable ``output_h`` complex ``[N, n_sym, Nr, Nt]`` of which only symbol 0 is kept (:29-33), normalisation
rules (:40-49), QPSK pilots drawn from numpy's legacy global RNG with two ``binomial`` calls, real part first
(:52-55), and two ``normal`` draws per ``__getitem__`` (:78-79) so the global stream stays aligned with the
reference.  The per-item 64x64 ``eigvals`` (:83-85) is not on the estimation path and only computed on request.

Reading the data: the reference uses ``hdf5storage`` (MATLAB v7.3 = HDF5).  Here ``h5py`` is used when it is
installed, otherwise the package's own dependency-free reader (``mat73.py``: super block, old-style groups, compound
``{real, imag}``, contiguous / chunked + deflate layouts -- what ``save -v7.3`` writes); a sibling ``.npz`` / ``.npy``
file with the same stem (key ``output_h``) is accepted as an equivalent; ``synthetic=True`` generates CDL-like channels with ``synth.generate_output_h`` instead (the
reference's data blobs are not distributed).
"""
import os

import numpy as np

from . import synth


def read_output_h(filename):
    """Return ``output_h`` as a complex ndarray ``[N, n_sym, Nr, Nt]`` from ``.mat`` (v7.3), ``.npz`` or ``.npy``."""
    stem = os.path.splitext(filename)[0]
    for alt in (stem + '.npz', stem + '.npy'):
        if os.path.exists(alt):
            arr = np.load(alt)
            return np.asarray(arr['output_h'] if hasattr(arr, 'files') else arr)
    if not os.path.exists(filename):
        raise FileNotFoundError('%s (or %s.npz / .npy) not found; the reference data blobs are not distributed -- '
                                'generate them with matlab/generate_data.m or pass synthetic=True' % (filename, stem))
    try:
        import h5py
    except ImportError:
        # no HDF5 library in this environment: the package's own reader for MATLAB's subset of the format
        from .mat73 import loadmat_variable
        return loadmat_variable(filename, 'output_h')
    with h5py.File(filename, 'r') as f:
        d = f['output_h'][()]
    if d.dtype.names:                                  # MATLAB stores complex as a compound {real, imag}
        d = d['real'] + 1j * d['imag']
    return np.transpose(d)                             # HDF5 keeps MATLAB's column-major order reversed


def _herm(a):
    return np.conj(np.transpose(a))


def _real_view(a):
    """complex [..] -> float32 [2, ..] (re, im), the layout the score network is trained on (loaders.py:72-73,88-91)."""
    return np.stack((a.real, a.imag), axis=0).astype(np.float32)


class Channels:
    """MIMO Channels (map-style dataset; usable with ``torch.utils.data.DataLoader``)."""

    def __init__(self, seed, config, norm=None, synthetic=False, data_dir='./data', num_synthetic=200,
                 compute_eig=False):
        target_spacings = config.data.spacing_list
        target_channel = config.data.channel
        nr, nt = int(config.data.image_size[0]), int(config.data.image_size[1])
        self.channels, self.filenames = [], []
        self.spacings = np.copy(target_spacings)
        self.compute_eig = compute_eig
        for spacing in target_spacings:
            filename = os.path.join(data_dir, '%s_Nt%d_Nr%d_ULA%.2f_seed%d.mat' % (target_channel, nt, nr, spacing, seed))
            self.filenames.append(filename)
            if synthetic:
                contents = synth.generate_output_h(target_channel, num_synthetic, nt, nr, spacing, seed, n_sym=1)
            else:
                contents = read_output_h(filename)
            channels = np.asarray(contents, dtype=np.complex64)
            self.channels.append(channels[:, 0])                       # first subcarrier of each symbol
        self.channels = np.asarray(self.channels)
        self.channels = np.reshape(self.channels, (-1, self.channels.shape[-2], self.channels.shape[-1]))

        if type(norm) == list:
            self.mean, self.std = norm[0], norm[1]
        elif norm == 'entrywise':
            self.mean, self.std = np.mean(self.channels, axis=0), np.std(self.channels, axis=0)
        elif norm == 'global':
            self.mean, self.std = 0., np.std(self.channels)
        else:
            raise ValueError("norm must be [mean, std], 'entrywise' or 'global' (got %r)" % (norm,))

        size = (self.channels.shape[0], config.data.image_size[1], config.data.num_pilots)
        self.pilots = 1 / np.sqrt(2) * (2 * np.random.binomial(1, 0.5, size=size) - 1 +
                                        1j * (2 * np.random.binomial(1, 0.5, size=size) - 1))
        self.noise_power = 1 / np.sqrt(2) * config.data.noise_std

    def __len__(self):
        return len(self.channels)

    def __getitem__(self, idx):
        if hasattr(idx, 'tolist'):
            idx = idx.tolist()
        h = self.channels[idx]                                  # [Nr, Nt] complex64, un-normalised
        hn = (h - self.mean) / self.std
        pil = self.pilots[idx]                                  # [Nt, Np]
        # received pilots with the dataset's own (usually zero) noise level; the two normal draws keep numpy's
        # global stream in step with the reference (loaders.py:78-79)
        y = h @ pil
        y = y + self.noise_power * (np.random.normal(size=y.shape) + 1j * np.random.normal(size=y.shape))
        sample = {'H': _real_view(hn), 'H_herm': _real_view(_herm(hn)),
                  'H_herm_cplx': _herm(h).astype(np.complex64),
                  'P': pil.astype(np.complex64), 'P_herm': _herm(pil).astype(np.complex64),
                  'Y': y.astype(np.complex64), 'Y_herm': _herm(y).astype(np.complex64),
                  'sigma_n': np.float32(self.noise_power), 'idx': int(idx)}
        if self.compute_eig:                                    # loaders.py:82-85 (first eigenvalue of P P^H)
            sample['eig1'] = np.real(np.linalg.eigvals(pil @ _herm(pil)))[0].astype(np.float32)
        return sample

    def batch(self, n):
        """The first ``n`` items stacked, i.e. ``next(iter(DataLoader(self, batch_size=n, shuffle=False)))``
        (test_score.py:102-108) restricted to the tensors the estimation loop reads."""
        items = [self[i] for i in range(n)]
        return {k: np.stack([it[k] for it in items]) for k in ('H_herm', 'P', 'idx')}
