"""Checkpoint tensor grammar of the NCSNv2Deepest score network, seed-derived
weights, and the packing the MFMA convolution kernels consume.

The key names / shapes mirror what ``NCSNv2Deepest.state_dict()`` produces
(``ncsnv2/models/ncsnv2.py:198-262`` with the blocks of
``ncsnv2/models/layers.py:62-134,165-249,291-313,401-441`` and
``ncsnv2/models/normalization.py:150-162``) so that a real ``final_model.pt``
(``train_score.py:211-216``) loads unchanged.  The pretrained blobs are not
shipped with the reference, so tests and the benchmark use weights drawn from a
fixed numpy bit-generator ("seed-derived weights").
"""
import numpy as np


def _residual_block(prefix, cin, cout, resample, dilation):
    """Tensor list of one ResidualBlock, in ``state_dict`` order (layers.py:401-441)."""
    down = resample == 'down'
    pooled = down and dilation is None            # ConvMeanPool wraps the conv in ``.conv``
    c1_out = cin if down else cout
    t = [(prefix + 'conv1.weight', (c1_out, cin, 3, 3)), (prefix + 'conv1.bias', (c1_out,))]
    t += [(prefix + 'normalize2.' + k, (c1_out,)) for k in ('alpha', 'gamma', 'beta')]
    c2 = prefix + ('conv2.conv.' if pooled else 'conv2.')
    t += [(c2 + 'weight', (cout, c1_out, 3, 3)), (c2 + 'bias', (cout,))]
    if cin != cout or resample is not None:
        sc = prefix + ('shortcut.conv.' if pooled else 'shortcut.')
        k = 1 if pooled else 3
        t += [(sc + 'weight', (cout, cin, k, k)), (sc + 'bias', (cout,))]
    t += [(prefix + 'normalize1.' + k, (cin,)) for k in ('alpha', 'gamma', 'beta')]
    return t


def _refine_block(prefix, in_planes, features, start=False, end=False):
    """Tensor list of one RefineBlock (layers.py:214-232): RCU adapt, RCU out, MSF, CRP."""
    t = []
    for i, c in enumerate(in_planes):
        for blk in (1, 2):
            for stage in (1, 2):
                t.append((prefix + 'adapt_convs.%d.%d_%d_conv.weight' % (i, blk, stage), (c, c, 3, 3)))
    for blk in range(1, (3 if end else 1) + 1):
        for stage in (1, 2):
            t.append((prefix + 'output_convs.%d_%d_conv.weight' % (blk, stage), (features, features, 3, 3)))
    if not start:
        for i, c in enumerate(in_planes):
            t.append((prefix + 'msf.convs.%d.weight' % i, (features, c, 3, 3)))
            t.append((prefix + 'msf.convs.%d.bias' % i, (features,)))
    for i in range(2):
        t.append((prefix + 'crp.convs.%d.weight' % i, (features, features, 3, 3)))
    return t


def state_dict_spec(ngf=32, channels=2, num_classes=2311):
    """Ordered ``[(name, shape)]`` of every tensor in the reference ``state_dict``."""
    t = [('sigmas', (num_classes,)),
         ('begin_conv.weight', (ngf, channels, 3, 3)), ('begin_conv.bias', (ngf,))]
    t += [('normalizer.' + k, (ngf,)) for k in ('alpha', 'gamma', 'beta')]
    t += [('end_conv.weight', (channels, ngf, 3, 3)), ('end_conv.bias', (channels,))]
    stages = [('res1', ngf, ngf, None, None, ngf, None),
              ('res2', ngf, 2 * ngf, 'down', None, 2 * ngf, None),
              ('res3', 2 * ngf, 2 * ngf, 'down', None, 2 * ngf, None),
              ('res31', 2 * ngf, 2 * ngf, 'down', None, 2 * ngf, None),
              ('res4', 2 * ngf, 4 * ngf, 'down', 2, 4 * ngf, 2),
              ('res5', 4 * ngf, 4 * ngf, 'down', 4, 4 * ngf, 4)]
    for name, cin, cout, resample, dil, _, _ in stages:
        t += _residual_block(name + '.0.', cin, cout, resample, dil)
        t += _residual_block(name + '.1.', cout, cout, None, dil)
    t += _refine_block('refine1.', [4 * ngf], 4 * ngf, start=True)
    t += _refine_block('refine2.', [4 * ngf, 4 * ngf], 2 * ngf)
    t += _refine_block('refine3.', [2 * ngf, 2 * ngf], 2 * ngf)
    t += _refine_block('refine31.', [2 * ngf, 2 * ngf], 2 * ngf)
    t += _refine_block('refine4.', [2 * ngf, 2 * ngf], ngf)
    t += _refine_block('refine5.', [ngf, ngf], ngf, end=True)
    return t


def get_sigmas(config):
    """Geometric noise schedule: float64 ``exp(linspace(log s1, log sL, L))`` rounded to
    float32 (``ncsnv2/models/__init__.py:4-8``)."""
    m = config.model
    if m.sigma_dist != 'geometric':
        raise NotImplementedError('only the geometric sigma schedule is on the hot path')
    return np.exp(np.linspace(np.log(m.sigma_begin), np.log(m.sigma_end),
                              m.num_classes)).astype(np.float32)


def seeded_state_dict(config, seed=2024):
    """Deterministic stand-in for the missing pretrained weights.

    Convolutions: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias (the
    distribution ``nn.Conv2d`` starts from); InstanceNorm++ ``alpha, gamma ~ N(1, 0.02)``
    (``normalization.py:158-159``), ``beta ~ N(0, 0.05)`` (non-zero so the term is exercised).
    Drawn tensor by tensor, in ``state_dict_spec`` order, from ``PCG64(seed)``.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for name, shape in state_dict_spec(config.model.ngf, config.data.channels,
                                       config.model.num_classes):
        if name == 'sigmas':
            sd[name] = get_sigmas(config)
        elif name.endswith('.weight'):
            fan_in = shape[1] * shape[2] * shape[3]
            b = 1.0 / np.sqrt(fan_in)
            sd[name] = rng.uniform(-b, b, size=shape).astype(np.float32)
        elif name.endswith('.bias'):
            wshape = sd[name[:-4] + 'weight'].shape
            b = 1.0 / np.sqrt(wshape[1] * wshape[2] * wshape[3])
            sd[name] = rng.uniform(-b, b, size=shape).astype(np.float32)
        elif name.endswith('.beta'):
            sd[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:  # alpha, gamma
            sd[name] = (1.0 + 0.02 * rng.standard_normal(shape)).astype(np.float32)
    return sd


def check_state_dict(sd, config):
    """Raise ``KeyError``/``ValueError`` like ``load_state_dict(strict=True)`` would."""
    spec = state_dict_spec(config.model.ngf, config.data.channels, config.model.num_classes)
    missing = [n for n, _ in spec if n not in sd]
    unexpected = [n for n in sd if n not in dict(spec)]
    if missing or unexpected:
        raise KeyError('state_dict mismatch: missing %s, unexpected %s' % (missing[:5], unexpected[:5]))
    for n, shape in spec:
        if tuple(sd[n].shape) != tuple(shape):
            raise ValueError('size mismatch for %s: %s vs %s' % (n, tuple(sd[n].shape), shape))


def pack_conv_weight(w):
    """Re-order an ``[O, C, k, k]`` convolution weight into MFMA B-operand fragments.

    Result ``[k*k, C/8, O/32, 64, 4]`` float32: for tap ``t = kh*k + kw``, channel group
    ``g`` and output block ``n``, lane ``l`` of a wavefront holds the four values
    ``w[n*32 + (l & 31), g*8 + 4*(l >> 5) + j, kh, kw]``, ``j = 0..3`` -- i.e. one 16-byte load
    per lane feeds four ``v_mfma_f32_32x32x2_f32`` issues (lanes 0-31 supply k-row 0, lanes
    32-63 k-row 1 of each).  Requires ``C % 8 == 0`` and ``O % 32 == 0``.
    """
    w = np.asarray(w, dtype=np.float32)
    o, c, kh, kw = w.shape
    if c % 8 or o % 32:
        raise ValueError('pack_conv_weight needs C %% 8 == 0 and O %% 32 == 0, got %s' % (w.shape,))
    a = w.reshape(o // 32, 32, c // 8, 2, 4, kh * kw)     # [nb, l31, g, half, j, tap]
    a = a.transpose(5, 2, 0, 3, 1, 4)                      # [tap, g, nb, half, l31, j]
    return np.ascontiguousarray(a).reshape(kh * kw, c // 8, o // 32, 64, 4)


_WINO_G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], np.float64)


def pack_conv_weight_winograd(w):
    """Winograd F(2x2, 3x3) weights of a 3x3 convolution in MFMA B-operand fragment order.

    ``U[xi][nu] = (G g G^T)[xi][nu]`` per (cout, cin) pair (Lavin & Gray 2016), computed in float64 and rounded
    once to float32, then packed exactly like ``pack_conv_weight`` with the 16 transform positions in place of the 9
    taps: ``[16, C/8, O/32, 64, 4]``.  The kernel multiplies them with the transformed input ``B^T d B`` and applies
    ``A^T . A``; 2.25x fewer multiplications than the direct 3x3 stencil, fp32 error ~1e-6 relative.
    """
    w = np.asarray(w, np.float64)
    if w.shape[2:] != (3, 3):
        raise ValueError('Winograd F(2x2,3x3) needs a 3x3 kernel, got %s' % (w.shape,))
    u = np.einsum('ij,ocjk,lk->ocil', _WINO_G, w, _WINO_G)          # [O, C, 4, 4]
    return pack_conv_weight(u.astype(np.float32))


def bf16_round(x):
    """float32 -> nearest-even bfloat16, returned as float32 values (what ``v_cvt_pk_bf16_f32`` produces)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split_bf16x3(x):
    """Exact three-term bfloat16 expansion ``x = h + m + l`` of float32 values (each term returned as float32)."""
    x = np.asarray(x, np.float32)
    h = bf16_round(x)
    r = x - h
    m = bf16_round(r)
    return h, m, bf16_round(r - m)


def pack_conv_weight_split(w):
    """Split-bf16 form of an ``[O, C, k, k]`` weight for the bf16-matrix-core convolution (``csrc/conv_x3.hip``).

    Result ``[k*k, C/16, O/32, 3, 64, 8]`` uint16 (bf16 bit patterns): for tap ``t``, 16-channel group ``g``, output block
    ``n`` and term ``s`` (0 = high, 1 = middle, 2 = low), lane ``l`` holds the eight values
    ``term_s(w[n*32 + (l & 31), g*16 + 8*(l >> 5) + j, kh, kw])``, ``j = 0..7`` -- the B operand of one
    ``v_mfma_f32_32x32x16_bf16``.  Requires ``C % 16 == 0`` and ``O % 32 == 0``.
    """
    w = np.asarray(w, np.float32)
    o, c, kh, kw = w.shape
    if c % 16 or o % 32:
        raise ValueError('pack_conv_weight_split needs C %% 16 == 0 and O %% 32 == 0, got %s' % (w.shape,))
    terms = np.stack([t.view(np.uint32) >> 16 for t in split_bf16x3(w)]).astype(np.uint16)   # [3, O, C, kh, kw]
    a = terms.reshape(3, o // 32, 32, c // 16, 2, 8, kh * kw)       # [s, nb, l31, g, half, j, tap]
    a = a.transpose(6, 3, 1, 0, 4, 2, 5)                            # [tap, g, nb, s, half, l31, j]
    return np.ascontiguousarray(a).reshape(kh * kw, c // 16, o // 32, 3, 64, 8)


def pack_conv_weight_winograd_split(w):
    """Winograd F(2x2, 3x3) weights ``U = G g G^T`` (float64, rounded once to float32) split into three bf16 terms in
    MFMA B-operand fragment order ``[16, C/16, O/32, 3, 64, 8]`` uint16 (``csrc/conv_wx3.hip``)."""
    w = np.asarray(w, np.float64)
    if w.shape[2:] != (3, 3):
        raise ValueError('Winograd F(2x2,3x3) needs a 3x3 kernel, got %s' % (w.shape,))
    u = np.einsum('ij,ocjk,lk->ocil', _WINO_G, w, _WINO_G)          # [O, C, 4, 4]
    return pack_conv_weight_split(u.astype(np.float32))


def round_fp16(x):
    """float32 -> nearest-even float16 -> float32: what ``tensor.half().float()`` gives (BASELINE config 5, "fp16 score-net
    weights")."""
    return np.asarray(x, np.float32).astype(np.float16).astype(np.float32)


def fp16_state_dict(sd):
    """Every learnable tensor of a ``state_dict`` rounded to fp16 (``module.half()`` semantics for parameters); the
    ``sigmas`` buffer -- the noise schedule, not a weight -- is kept in float32."""
    return {k: (np.asarray(v, np.float32) if k == 'sigmas' else round_fp16(v)) for k, v in sd.items()}


def pack_conv_weight_f16(w):
    """Single-term fp16 form of an ``[O, C, k, k]`` weight: the layout of ``pack_conv_weight_split`` without the term axis,
    ``[k*k, C/16, O/32, 64, 8]`` uint16 (fp16 bit patterns, round to nearest even) -- the B operand of one
    ``v_mfma_f32_32x32x16_f16`` (``SBC_CONV_F16W``)."""
    w = np.asarray(w, np.float32)
    o, c, kh, kw = w.shape
    if c % 16 or o % 32:
        raise ValueError('pack_conv_weight_f16 needs C %% 16 == 0 and O %% 32 == 0, got %s' % (w.shape,))
    a = w.astype(np.float16).view(np.uint16).reshape(o // 32, 32, c // 16, 2, 8, kh * kw)     # [nb, l31, g, half, j, tap]
    a = a.transpose(5, 2, 0, 3, 1, 4)                                                        # [tap, g, nb, half, l31, j]
    return np.ascontiguousarray(a).reshape(kh * kw, c // 16, o // 32, 64, 8)


def pack_conv_weight_winograd_f16(w):
    """Winograd F(2x2, 3x3) weights ``U = G g G^T`` (float64, rounded to float32 and then once to fp16) in the
    ``pack_conv_weight_f16`` layout with the 16 transform positions in place of the taps: ``[16, C/16, O/32, 64, 8]``."""
    w = np.asarray(w, np.float64)
    if w.shape[2:] != (3, 3):
        raise ValueError('Winograd F(2x2,3x3) needs a 3x3 kernel, got %s' % (w.shape,))
    u = np.einsum('ij,ocjk,lk->ocil', _WINO_G, w, _WINO_G)          # [O, C, 4, 4]
    return pack_conv_weight_f16(u.astype(np.float32))


# ---- f16x2 forms (SBC_CONV_F16X2, conv_mode 'f16x2') -----------------------------------------------------------------
F16X2_ACT_SHIFT = 0          # (include/sbc_hip.h: SBC_F16X2_ACT_SHIFT) the packers' own act_scale is 2^0; calibration sets the real one


def f16x2_shift(w):
    """The power of two that puts the largest magnitude of ``w`` into ``[2^13, 2^14)``."""
    m = float(np.max(np.abs(w))) if np.size(w) else 0.0
    if not (m > 0.0) or not np.isfinite(m):
        return 0
    _, e = np.frexp(np.float32(m))                # m = f * 2^e, f in [0.5, 1)
    return int(min(100, max(-100, 14 - int(e))))


def split_f16x2(x):
    """Two-term fp16 expansion ``x ~ h + l`` of float32 values (``h = fp16(x)``, ``l = fp16(x - h)``; both returned as
    float16): the representation error is at most 2^-22 |x| while ``l`` is a normal fp16 number."""
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16)
    l = (x - h.astype(np.float32)).astype(np.float16)
    return h, l


def f16x2_trailer(s, act_scale=1.0):
    """The 16-byte record behind the fragments: (act_scale, descale = 1 / (act_scale 2^s), the weights' own descale 2^-s, 0).
    ``act_scale`` must be a power of two (the library's calibration, ``sbc_f16x2_calibrate``, rewrites the first two words on
    the device copy); the packers write 1."""
    a = float(act_scale) * 2.0 ** F16X2_ACT_SHIFT
    if a <= 0 or np.frexp(a)[0] != 0.5:
        raise ValueError('act_scale must be a positive power of two, got %r' % (act_scale,))
    return np.array([a, 2.0 ** -s / a, 2.0 ** -s, 0], np.float32).view(np.uint16)


def _pack_f16x2(w, act_scale=1.0):
    """``[O, C, T]`` float32 (T = taps or the 16 Winograd positions) -> ``[T, C/16, O/32, 2, 64, 8]`` uint16 + trailer."""
    o, c, t = w.shape
    if c % 16 or o % 32:
        raise ValueError('f16x2 packing needs C %% 16 == 0 and O %% 32 == 0, got %s' % (w.shape,))
    s = f16x2_shift(w)
    terms = np.stack([v.view(np.uint16) for v in split_f16x2(np.ldexp(w, s).astype(np.float32))])   # [2, O, C, T]
    a = terms.reshape(2, o // 32, 32, c // 16, 2, 8, t)              # [s, nb, l31, g, half, j, tap]
    a = np.ascontiguousarray(a.transpose(6, 3, 1, 0, 4, 2, 5)).reshape(-1)   # [tap, g, nb, s, half, l31, j]
    return np.concatenate([a, f16x2_trailer(s, act_scale)])


def pack_conv_weight_f16x2(w, act_scale=1.0):
    """Two-term fp16 form of an ``[O, C, k, k]`` weight (``csrc/conv_x3.hip``, ``TERMS = 2``): the layer's weights scaled by
    the power of two of ``f16x2_shift`` and split into ``h + l`` fp16 terms, in the fragment order of
    ``pack_conv_weight_split`` with two terms, flat uint16, followed by the 16-byte trailer
    ``(act_scale, descale, 0, 0)`` float32 the kernels read their scales from."""
    w = np.asarray(w, np.float32)
    o, c, kh, kw = w.shape
    return _pack_f16x2(w.reshape(o, c, kh * kw), act_scale)


def pooled_filter(w):
    """The filter of ``meanpool2(conv_k(x))`` as ONE stride-2 convolution (ConvMeanPool, layers.py:309-313): ``[O, C, k, k]`` ->
    ``[O, C, k + 1, k + 1]`` with ``W'[p][q] = 1/4 sum_{a,b in {0,1}} W[p - a][q - b]``, formed in float64, rounded once."""
    w = np.asarray(w, np.float64)
    o, c, k, _ = w.shape
    out = np.zeros((o, c, k + 1, k + 1), np.float64)
    for a in range(2):
        for b in range(2):
            out[:, :, a:a + k, b:b + k] += w
    return (0.25 * out).astype(np.float32)


def pack_conv_weight_pooled_f16x2(w, act_scale=1.0):
    """``pack_conv_weight_f16x2`` of ``pooled_filter(w)`` (``csrc/conv_down.hip``; identical to ``sbc_pack_conv_weight_pooled_f16x2``)."""
    return pack_conv_weight_f16x2(pooled_filter(w), act_scale)


def pack_conv_weight_winograd_f16x2(w, act_scale=1.0):
    """Winograd F(2x2, 3x3) weights ``U = G g G^T`` (float64, rounded once to float32) in the ``pack_conv_weight_f16x2``
    form with the 16 transform positions in place of the taps (``csrc/conv_wx3.hip``, ``MODE = 2``)."""
    w = np.asarray(w, np.float64)
    if w.shape[2:] != (3, 3):
        raise ValueError('Winograd F(2x2,3x3) needs a 3x3 kernel, got %s' % (w.shape,))
    u = np.einsum('ij,ocjk,lk->ocil', _WINO_G, w, _WINO_G).astype(np.float32)          # [O, C, 4, 4]
    return _pack_f16x2(u.reshape(u.shape[0], u.shape[1], 16), act_scale)
