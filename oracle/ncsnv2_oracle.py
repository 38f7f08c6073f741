"""ORACLE -- test infrastructure, NOT product code.

CPU (numpy, float32) restatement of the score network the reference evaluates in its
annealed-Langevin loop: ``NCSNv2Deepest.forward`` (``ncsnv2/models/ncsnv2.py:269-300``)
and the blocks it is made of.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product path
(``score_based_channels_amd``) never does.

Pinning: the reference has no tests or golden vectors of its own, and
the heavy arithmetic lives in PyTorch (``torch==2.3.1`` in the reference's
``pyproject.toml:16``: ``F.conv2d``, ``F.instance_norm``, ``F.elu``, ``F.max_pool2d``,
``F.interpolate``).  This restatement is therefore pinned against outputs of the reference
classes themselves, imported on CPU in the build container by ``tests/gen_golden.py`` and
committed under ``tests/golden/`` (``tests/test_oracle_golden.py`` checks them).

Layout is NCHW like the reference so every function can be read against the cited lines.
``sd`` is a ``state_dict``-style mapping name -> ndarray with the reference's key names.
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- primitives
def conv2d(x, w, b=None, dilation=1):
    """``nn.Conv2d`` stride 1, ``padding = dilation * (k // 2)`` (layers.py:28-60: conv1x1,
    conv3x3, dilated_conv3x3; cross-correlation, zero padding).  GEMM over K = k*k*C."""
    x = np.asarray(x, F32)
    w = np.asarray(w, F32)
    n, c, h, wd = x.shape
    o, _, kh, kw = w.shape
    ph, pw = dilation * (kh // 2), dilation * (kw // 2)
    xp = np.zeros((c, n, h + 2 * ph, wd + 2 * pw), F32)
    xp[:, :, ph:ph + h, pw:pw + wd] = x.transpose(1, 0, 2, 3)
    cols = np.empty((kh * kw, c, n, h * wd), F32)                  # im2col, K = (tap, C)
    for i in range(kh):
        for j in range(kw):
            cols[i * kw + j] = xp[:, :, i * dilation:i * dilation + h,
                                  j * dilation:j * dilation + wd].reshape(c, n, h * wd)
    wm = w.transpose(0, 2, 3, 1).reshape(o, kh * kw * c)           # [O, (tap, C)]
    out = np.matmul(wm, cols.reshape(kh * kw * c, n * h * wd))     # one sgemm: [O, N*HW]
    if b is not None:
        out = out + np.asarray(b, F32)[:, None]
    return np.ascontiguousarray(out.reshape(o, n, h, wd).transpose(1, 0, 2, 3)).astype(F32)


def elu(x):
    """``nn.ELU()`` with alpha = 1 (layers.py:12-13)."""
    x = np.asarray(x, F32)
    return np.where(x > 0, x, np.expm1(np.minimum(x, F32(0)))).astype(F32)


def instance_norm_plus(x, alpha, gamma, beta):
    """``InstanceNorm2dPlus.forward`` (normalization.py:163-176): biased per-plane variance
    for the instance norm (eps 1e-5), *unbiased* variance across channels for the
    mean-of-means branch (eps 1e-5)."""
    x = np.asarray(x, F32)
    means = x.mean(axis=(2, 3), dtype=F32)                              # [N, C]
    m = means.mean(axis=-1, keepdims=True, dtype=F32)
    v = means.var(axis=-1, keepdims=True, ddof=1, dtype=F32)
    means_n = (means - m) / np.sqrt(v + F32(1e-5))
    mu = means[:, :, None, None]
    var = np.mean((x - mu) ** 2, axis=(2, 3), keepdims=True, dtype=F32)
    h = (x - mu) / np.sqrt(var + F32(1e-5))
    h = h + means_n[:, :, None, None] * alpha[None, :, None, None]
    return (gamma[None, :, None, None] * h + beta[None, :, None, None]).astype(F32)


def max_pool5(x):
    """``nn.MaxPool2d(kernel_size=5, stride=1, padding=2)`` (layers.py:69; -inf padding)."""
    n, c, h, w = x.shape
    xp = np.full((n, c, h + 4, w + 4), -np.inf, F32)
    xp[:, :, 2:2 + h, 2:2 + w] = x
    out = np.full_like(x, -np.inf)
    for i in range(5):
        for j in range(5):
            out = np.maximum(out, xp[:, :, i:i + h, j:j + w])
    return out


def bilinear_align_corners(x, size):
    """``F.interpolate(mode='bilinear', align_corners=True)`` (layers.py:182).  Source
    coordinate = dst * (in - 1) / (out - 1); same-size calls are exact copies."""
    n, c, h, w = x.shape
    oh, ow = int(size[0]), int(size[1])
    if (oh, ow) == (h, w):
        return x.copy()

    def axis(inp, out):
        scale = F32(inp - 1) / F32(out - 1) if out > 1 else F32(0)
        src = (scale * np.arange(out, dtype=F32)).astype(F32)
        i0 = np.minimum(np.floor(src).astype(np.int64), inp - 1)
        i1 = np.minimum(i0 + 1, inp - 1)
        l1 = (src - i0.astype(F32)).astype(F32)
        return i0, i1, (F32(1) - l1).astype(F32), l1

    h0, h1, lh0, lh1 = axis(h, oh)
    w0, w1, lw0, lw1 = axis(w, ow)
    top = x[:, :, h0][:, :, :, w0] * lw0 + x[:, :, h0][:, :, :, w1] * lw1
    bot = x[:, :, h1][:, :, :, w0] * lw0 + x[:, :, h1][:, :, :, w1] * lw1
    return (lh0[None, None, :, None] * top + lh1[None, None, :, None] * bot).astype(F32)


def mean_pool2(x):
    """The pooling of ``ConvMeanPool.forward`` (layers.py:311-312): python ``sum`` of the
    four stride-2 views, i.e. ``(((0 + a) + b) + c) + d`` with a=[0::2,0::2], b=[1::2,0::2],
    c=[0::2,1::2], d=[1::2,1::2], then ``/ 4``."""
    return ((((x[:, :, ::2, ::2] + x[:, :, 1::2, ::2]) + x[:, :, ::2, 1::2])
             + x[:, :, 1::2, 1::2]) / F32(4.)).astype(F32)


# ----------------------------------------------------------------------------- blocks
def _norm(sd, prefix, x):
    return instance_norm_plus(x, sd[prefix + 'alpha'], sd[prefix + 'gamma'], sd[prefix + 'beta'])


def residual_block(sd, prefix, x, resample, dilation):
    """``ResidualBlock.forward`` (layers.py:443-456) for the three variants NCSNv2Deepest
    instantiates: plain, 'down' with ConvMeanPool (dilation None), and dilated (no pooling
    even when resample == 'down', layers.py:411-415)."""
    d = 1 if dilation is None else dilation
    pooled = resample == 'down' and dilation is None
    out = elu(_norm(sd, prefix + 'normalize1.', x))
    out = conv2d(out, sd[prefix + 'conv1.weight'], sd[prefix + 'conv1.bias'], d)
    out = elu(_norm(sd, prefix + 'normalize2.', out))
    if pooled:
        out = mean_pool2(conv2d(out, sd[prefix + 'conv2.conv.weight'], sd[prefix + 'conv2.conv.bias']))
        shortcut = mean_pool2(conv2d(x, sd[prefix + 'shortcut.conv.weight'],
                                     sd[prefix + 'shortcut.conv.bias']))
    else:
        out = conv2d(out, sd[prefix + 'conv2.weight'], sd[prefix + 'conv2.bias'], d)
        if (prefix + 'shortcut.weight') in sd:
            shortcut = conv2d(x, sd[prefix + 'shortcut.weight'], sd[prefix + 'shortcut.bias'], d)
        else:
            shortcut = x
    return (shortcut + out).astype(F32)


def rcu_block(sd, prefix, x, n_blocks, n_stages=2):
    """``RCUBlock.forward`` (layers.py:126-134); convs have no bias (layers.py:118)."""
    for i in range(n_blocks):
        residual = x
        for j in range(n_stages):
            x = conv2d(elu(x), sd[prefix + '%d_%d_conv.weight' % (i + 1, j + 1)])
        x = x + residual
    return x.astype(F32)


def crp_block(sd, prefix, x, n_stages=2):
    """``CRPBlock.forward`` (layers.py:76-83) with max-pooling (RefineBlock default)."""
    x = elu(x)
    path = x
    for i in range(n_stages):
        path = conv2d(max_pool5(path), sd[prefix + 'convs.%d.weight' % i])
        x = path + x
    return x.astype(F32)


def msf_block(sd, prefix, xs, shape):
    """``MSFBlock.forward`` (layers.py:178-184): fp32 zero accumulator, inputs in order."""
    sums = None
    for i, xi in enumerate(xs):
        h = conv2d(xi, sd[prefix + 'convs.%d.weight' % i], sd[prefix + 'convs.%d.bias' % i])
        h = bilinear_align_corners(h, shape)
        sums = (np.zeros_like(h) + h) if sums is None else sums + h
    return sums.astype(F32)


def refine_block(sd, prefix, xs, shape, start=False, end=False):
    """``RefineBlock.forward`` (layers.py:234-249)."""
    hs = [rcu_block(sd, prefix + 'adapt_convs.%d.' % i, xi, 2) for i, xi in enumerate(xs)]
    h = msf_block(sd, prefix + 'msf.', hs, shape) if len(xs) > 1 else hs[0]
    h = crp_block(sd, prefix + 'crp.', h)
    return rcu_block(sd, prefix + 'output_convs.', h, 3 if end else 1)


# ----------------------------------------------------------------------------- network
def score_forward(sd, x, labels, return_stages=False):
    """``NCSNv2Deepest.forward(x, y)`` (ncsnv2.py:269-300).

    x: float32 ``[B, 2, Nt, Nr]`` (real view of the Hermitian channel), labels: int ``[B]``
    noise-level indices.  Includes the ``h = 2*x - 1`` input map (ncsnv2.py:270-273; both
    config flags are falsy for checkpoints written by train_score.py) and the final
    ``output / sigmas[y]`` (ncsnv2.py:295-298).
    """
    x = np.asarray(x, F32)
    st = {}
    h = (F32(2) * x - F32(1.)).astype(F32)
    out = conv2d(h, sd['begin_conv.weight'], sd['begin_conv.bias'])
    st['begin'] = out
    plan = [('res1', None, None), ('res2', 'down', None), ('res3', 'down', None),
            ('res31', 'down', None), ('res4', 'down', 2), ('res5', 'down', 4)]
    layers = []
    for name, resample, dil in plan:
        out = residual_block(sd, name + '.0.', out, resample, dil)
        out = residual_block(sd, name + '.1.', out, None, dil)
        layers.append(out)
        st[name] = out
    l1, l2, l3, l31, l4, l5 = layers
    ref1 = refine_block(sd, 'refine1.', [l5], l5.shape[2:], start=True)
    ref2 = refine_block(sd, 'refine2.', [l4, ref1], l4.shape[2:])
    ref31 = refine_block(sd, 'refine31.', [l31, ref2], l31.shape[2:])
    ref3 = refine_block(sd, 'refine3.', [l3, ref31], l3.shape[2:])
    ref4 = refine_block(sd, 'refine4.', [l2, ref3], l2.shape[2:])
    out = refine_block(sd, 'refine5.', [l1, ref4], l1.shape[2:], end=True)
    st.update(refine1=ref1, refine2=ref2, refine31=ref31, refine3=ref3, refine4=ref4, refine5=out)
    out = elu(_norm(sd, 'normalizer.', out))
    out = conv2d(out, sd['end_conv.weight'], sd['end_conv.bias'])
    used = np.asarray(sd['sigmas'], F32)[np.asarray(labels, np.int64)].reshape(-1, 1, 1, 1)
    out = (out / used).astype(F32)
    return (out, st) if return_stages else out


def conv_flops_per_sample(ngf, nt, nr, channels=2):
    """Algorithmic convolution work of one score evaluation: 2 * MACs over all 113
    ``nn.Conv2d`` calls, bias / norm / ELU / pooling excluded (SURVEY.md section 8(d):
    0.820772864 GFLOP at 64x16, 13.132365824 GFLOP at 256x64)."""
    from score_based_channels_amd.weights import state_dict_spec
    res = {'begin_conv': 0, 'end_conv': 0, 'res1': 0, 'res2.0': 0, 'res2.1': 1, 'res3.0': 1,
           'res3.1': 2, 'res31.0': 2, 'res31.1': 3, 'res4': 3, 'res5': 3, 'refine1': 3,
           'refine2': 3, 'refine31': 3}
    total = 0
    for name, shape in state_dict_spec(ngf, channels, 1):
        if not name.endswith('.weight') or len(shape) != 4:
            continue
        lvl = None
        for k, v in res.items():
            if name.startswith(k + '.'):
                lvl = v
        if lvl is None:                       # refine3/4/5: adapt_convs.i / msf.convs.i run at the
            top = {'refine3': 2, 'refine4': 1, 'refine5': 0}[name.split('.')[0]]   # input's level
            second = ('adapt_convs.1.' in name) or ('msf.convs.1.' in name)
            lvl = top + 1 if second else top
        px = (nt >> lvl) * (nr >> lvl)
        total += 2 * px * shape[0] * shape[1] * shape[2] * shape[3]
    return total
