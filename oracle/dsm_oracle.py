"""ORACLE -- test infrastructure, NOT product code.

CPU (numpy) restatement of the denoising-score-matching loss of the reference, ``ncsnv2/losses/dsm.py:6-32``, as
``train_score.py:151-153`` calls it (``labels=None`` there: drawn with ``torch.randint``; every random draw is an explicit
argument here so that the HIP path and the oracle consume identical noise), and of the optimiser step around it
(``torch.optim.Adam`` as configured by ``ncsnv2/losses/__init__.py:3-7`` and ``train_score.py:43-49``; ``EMAHelper.update``,
``ncsnv2/models/ema.py:17-22``).  Pinned by ``tests/golden/train_dsm.npz``, which ``tests/gen_golden.py train`` produces
with the reference's own loss function, network, autograd, optimiser and EMA helper.
"""
import numpy as np

F32 = np.float32


def perturb(samples, sigmas, labels, z):
    """dsm.py:14-17: ``used_sigmas = sigmas[labels]``; ``noise = randn_like(samples) * used_sigmas``;
    ``perturbed = samples + noise``.  Returns (perturbed, noise, used_sigmas[B])."""
    used = np.asarray(sigmas, F32)[np.asarray(labels)]
    us = used.reshape((-1,) + (1,) * (samples.ndim - 1))
    noise = (np.asarray(z, F32) * us).astype(F32)
    return (np.asarray(samples, F32) + noise).astype(F32), noise, used


def loss_per_sample(scores, noise, used_sigmas, anneal_power=2.):
    """dsm.py:19-30: ``target = -1 / sigma^2 * noise``; ``1/2 * sum((scores - target)^2) * sigma^p`` per sample
    (the return value of the reference is the mean of these, :32)."""
    B = scores.shape[0]
    us = np.asarray(used_sigmas, F32).reshape(B, 1)
    target = (F32(-1) / (us ** 2)).astype(F32) * noise.reshape(B, -1)
    d = scores.reshape(B, -1).astype(F32) - target
    return (F32(0.5) * np.sum(d * d, axis=-1, dtype=F32) * us[:, 0] ** F32(anneal_power)).astype(F32)


def adam_ema_step(p, g, m, v, shadow, t, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-3, mu=0.999):
    """One ``torch.optim.Adam`` step (weight_decay 0, amsgrad False) followed by ``EMAHelper.update``; ``t`` = 1-based
    step count.  Returns the new (p, m, v, shadow), all float32."""
    p, g, m, v, shadow = (np.asarray(a, F32) for a in (p, g, m, v, shadow))
    m = (m + (g - m) * F32(1 - beta1)).astype(F32)
    v = (v * F32(beta2) + F32(1 - beta2) * (g * g)).astype(F32)
    step_size = F32(lr / (1 - beta1 ** t))
    denom = np.sqrt(v) / F32(np.sqrt(1 - beta2 ** t)) + F32(eps)
    p = (p - step_size * (m / denom)).astype(F32)
    shadow = (F32(1. - mu) * p + F32(mu) * shadow).astype(F32)
    return p, m, v, shadow
