"""MI355X-native annealed-Langevin MIMO channel estimation (the ``test_score`` /
``tune_hparams_score`` hot path of utcsilab/score-based-channels) on hand-written HIP kernels.

Only the hot path lives here: the score network, the data-consistency + Langevin update and the
NMSE reduction run as gfx950 kernels behind a C ABI (``csrc/``, ``include/sbc_hip.h``); the modules in
this package are the Python host side mirroring the reference's operator interface.
"""
__version__ = '0.1.0'
