"""Host-side Gaussian streams for reproducible ("external noise") annealed-Langevin runs.

The reference never seeds its RNG (``test_score.py:115,124,161`` call ``torch.randn_like`` on the
default CUDA generator), so "identical seeds" has to be defined by the build: every draw the loop
makes is keyed by ``(seed, purpose, snr index)`` on a counter-based numpy ``Philox`` generator whose
float32 ziggurat stream does not depend on platform or thread count.  The draw order inside one
stream follows the reference: one initial estimate per (spacing,
pilot_alpha) combination shared by all SNR points, then per SNR point one measurement-noise draw
followed by one draw per Langevin step.

A complex64 standard normal is two float32 N(0, 1/2) values, interleaved (re, im), exactly what
``torch.randn_like`` produces for a complex tensor.
"""
import numpy as np

_SQRT_HALF = np.float32(np.sqrt(0.5))
INIT, MEAS, STEP = 0, 1, 2


def _gen(seed, purpose, index):
    return np.random.Generator(np.random.Philox(key=[int(seed), (int(purpose) << 32) | int(index)]))


def complex_normal(rng, shape):
    z = rng.standard_normal(tuple(shape) + (2,), dtype=np.float32) * _SQRT_HALF
    return np.ascontiguousarray(z).view(np.complex64)[..., 0]


class HostNoise:
    """Keyed CN(0,1) draws for one (spacing, pilot_alpha) combination / grid cell ``combo``."""

    def __init__(self, seed, combo=0):
        self.seed = int(seed)
        self.combo = int(combo)

    def init(self, shape):
        """``init_val_H = randn_like(val_H)`` (test_score.py:115)."""
        return complex_normal(_gen(self.seed, INIT, self.combo << 16), shape)

    def measurement(self, snr_idx, shape):
        """The ``randn_like(val_Y)`` of test_score.py:124 for SNR point ``snr_idx``."""
        return complex_normal(_gen(self.seed, MEAS, (self.combo << 16) | snr_idx), shape)

    def step_stream(self, snr_idx, shape):
        """Returns ``f(k)`` giving the ``randn_like(current)`` of Langevin step ``k`` (test_score.py:161)
        for SNR point ``snr_idx``; steps must be requested in increasing order."""
        rng = _gen(self.seed, STEP, (self.combo << 16) | snr_idx)
        state = {'k': 0}

        def draw(k):
            if k != state['k']:
                raise ValueError('step noise must be drawn sequentially (got %d, expected %d)' % (k, state['k']))
            state['k'] += 1
            return complex_normal(rng, shape)
        return draw

    def step_block(self, snr_idx, shape, n_steps):
        """All ``n_steps`` step draws of one SNR point as ``[n_steps, *shape]`` complex64."""
        rng = _gen(self.seed, STEP, (self.combo << 16) | snr_idx)
        return np.stack([complex_normal(rng, shape) for _ in range(n_steps)])
