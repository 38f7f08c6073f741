"""Reading / writing score-model checkpoints in the reference's layout.

``train_score.py:211-216`` saves ``{'model_state', 'optim_state', 'config', 'train_loss', 'val_loss'}`` with
``torch.save``; ``config`` is a pickled ``dotmap.DotMap``.  ``dotmap`` is not a dependency here, so while loading
a stand-in module is registered that re-creates each pickled DotMap node as a plain holder of its state, which is
then converted to ``config.Config``.  Checkpoints written by ``save_checkpoint`` store the config as nested dicts.
"""
import sys
import types
from collections import OrderedDict

import numpy as np

from .config import Config


class _DotMapShim(object):
    """Receives the pickled state of a ``dotmap.DotMap`` (its ``__dict__``: ``_map`` OrderedDict + flags)."""

    def __init__(self, *args, **kwargs):
        self._map = OrderedDict()

    def __setstate__(self, state):
        self.__dict__.update(state)

    def __setitem__(self, k, v):          # some DotMap versions rebuild via item assignment
        self.__dict__.setdefault('_map', OrderedDict())[k] = v


def _to_config(node):
    if isinstance(node, _DotMapShim):
        node = node.__dict__.get('_map', {})
    if isinstance(node, dict):
        out = Config()
        for k, v in node.items():
            out[k] = _to_config(v)
        return out
    if isinstance(node, (list, tuple)):
        return type(node)(_to_config(v) for v in node)
    return node


def load_checkpoint(path, map_location='cpu'):
    """-> dict with ``model_state`` (name -> tensor) and ``config`` (``Config``); other keys passed through."""
    import torch
    had = sys.modules.get('dotmap')
    shim = types.ModuleType('dotmap')
    shim.DotMap = _DotMapShim
    _DotMapShim.__module__, _DotMapShim.__qualname__ = 'dotmap', 'DotMap'
    sys.modules['dotmap'] = shim
    try:
        contents = torch.load(path, map_location=map_location, weights_only=False)
    finally:
        if had is not None:
            sys.modules['dotmap'] = had
        else:
            del sys.modules['dotmap']
        _DotMapShim.__module__, _DotMapShim.__qualname__ = __name__, '_DotMapShim'
    if 'model_state' not in contents or 'config' not in contents:
        raise KeyError("%s is not a score-model checkpoint (needs 'model_state' and 'config')" % path)
    contents['config'] = _to_config(contents['config'])
    return contents


def save_checkpoint(path, state_dict, config, **extra):
    """Write ``final_model.pt`` with the reference's keys (config as nested dicts, loadable by ``load_checkpoint``)."""
    import torch
    sd = {k: torch.from_numpy(np.array(v)) if isinstance(v, np.ndarray) else v for k, v in state_dict.items()}
    torch.save(dict({'model_state': sd, 'config': config.toDict(), 'optim_state': None, 'train_loss': [],
                     'val_loss': []}, **extra), path)
