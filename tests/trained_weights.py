"""A checkpoint that has been TRAINED (missing item 5): the authors' ``final_model.pt`` is not distributed
(/root/reference/.MISSING_LARGE_BLOBS), so the repository makes its own -- 300 optimiser steps of the package's trainer
(``score_based_channels_amd.train_score --synthetic --max_steps 300 --seed 1``, the reference's configuration, loop and file format,
train_score.py:34-67,145-216) on the GPU.  A training step is reproducible bit for bit (tests/test_gpu_train.py), so the GPU test
re-creates the very weights the golden ``tests/golden/trained_*.npz`` was generated with (``tests/gen_golden.py`` loads them into
the imported reference) instead of shipping a 24 MB blob; the fixture holds a digest of every tensor to prove it.
"""
import os

import numpy as np

TRAIN_ARGV = ['--synthetic', '--max_steps', '300', '--seed', '1', '--val_every', '1000']
# round 6: a LONGER-trained checkpoint beside it -- 4000 optimiser steps, ~25 s of GPU -- to stress the f16x2
# calibration window with weights that have moved further from their initial statistics (golden: trained_4000steps.npz)
LONG_STEPS = 4000


def train_argv(steps=300):
    return ['--synthetic', '--max_steps', str(int(steps)), '--seed', '1', '--val_every', '1000' if steps == 300 else '1000000']


def train_checkpoint(out_dir, steps=300):
    """Run the trainer CLI into ``out_dir`` (GPU only) and return the ``model_state`` as numpy arrays."""
    import torch
    from score_based_channels_amd import train_score
    from score_based_channels_amd.checkpoint import load_checkpoint
    train_score.main(train_argv(steps) + ['--out_dir', str(out_dir)])
    torch.cuda.synchronize()
    ck = load_checkpoint(os.path.join(str(out_dir), 'final_model.pt'))
    return ck['config'], {k: np.asarray(v.detach().cpu().numpy() if hasattr(v, 'detach') else v) for k, v in ck['model_state'].items()}


def state_digest(sd):
    """Per-tensor fingerprints (conftest.tensor_digest) as one ``[n_tensors, 26]`` array in sorted key order, plus a checksum of
    every byte."""
    import zlib
    from conftest import tensor_digest
    keys = sorted(sd)
    dig = np.stack([tensor_digest(k, sd[k]) for k in keys])
    crc = np.array([zlib.crc32(np.ascontiguousarray(sd[k]).tobytes()) for k in keys], np.int64)
    return keys, dig, crc
