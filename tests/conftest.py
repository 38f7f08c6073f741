import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope='session')
def weights64():
    from score_based_channels_amd.config import default_config
    from score_based_channels_amd.weights import seeded_state_dict
    cfg = default_config()
    return cfg, seeded_state_dict(cfg, 2024)


def rel_err(a, b):
    ctype = np.complex128 if (np.iscomplexobj(a) or np.iscomplexobj(b)) else np.float64
    a = np.asarray(a, ctype)
    b = np.asarray(b, ctype)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


def rel_err_elementwise(a, b, floor=0.05):
    """Largest element-wise relative error over the elements of the reference that are at least ``floor`` of its largest
    magnitude (``rel_err`` is norm-wise: it does not bound the error of individual small-but-significant elements)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    sel = np.abs(b) >= floor * np.max(np.abs(b))
    return float(np.max(np.abs(a[sel] - b[sel]) / np.abs(b[sel])))


def tensor_digest(name, a, k=24):
    """Small fingerprint of a parameter-shaped array for fixtures that cannot hold 5.9 M values per case: its Euclidean
    norm, its sum, and ``k`` elements at positions derived from the tensor's name."""
    import zlib
    a = np.asarray(a, np.float64).ravel()
    idx = np.random.default_rng(zlib.crc32(name.encode())).integers(0, a.size, size=k)
    return np.concatenate([[np.sqrt(np.sum(a * a)), np.sum(a)], a[idx]])
