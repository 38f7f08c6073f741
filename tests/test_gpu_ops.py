"""GPU parity of every HIP operator against the CPU oracle's primitives, called through the C ABI
(``sbc_op_launch``).  Run on the MI355X box: ``python -m pytest tests -m gpu``."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err, rel_err_elementwise
from oracle import ncsnv2_oracle as O
from plan_interp import inorm_stats

pytestmark = pytest.mark.gpu

F32 = np.float32
TOL = 2e-5          # fp32 accumulate in a different order than the oracle's sgemm


@pytest.fixture(scope='module')
def gpu():
    import torch
    from score_based_channels_amd import _lib
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    _lib.lib()
    return torch, _lib


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _launch(gpu, op):
    torch, _lib = gpu
    _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()


def _p(t):
    return C.c_void_p(t.data_ptr())


CONV_CASES = [
    # cin, cout, k, dil, B, H, W, flags-set
    (32, 32, 3, 1, 3, 64, 16, 'norm_elu_res'),     # ResidualBlock conv2 at full resolution (TM=128 at B=3)
    (32, 32, 3, 1, 130, 64, 16, 'elu_res'),        # big tile (TM=256) with a partial last wave of samples
    (32, 32, 3, 1, 5, 32, 8, 'crp2'),              # CRP second conv: res1 with ELU + res2
    (32, 64, 3, 1, 3, 64, 16, 'norm_elu_pool_res'),  # ConvMeanPool 3x3 + pooled shortcut
    (32, 64, 1, 1, 3, 64, 16, 'pool'),             # ConvMeanPool 1x1 shortcut
    (64, 64, 3, 1, 7, 32, 8, 'norm_elu_res'),
    (64, 64, 3, 1, 600, 32, 8, 'plain'),           # TM=256 with 64 channels
    (64, 64, 3, 1, 9, 16, 4, 'norm_elu_pool_res'),  # pooling with W=4 (pairs cross half-waves)
    (64, 64, 1, 1, 9, 16, 4, 'pool'),
    (64, 64, 3, 2, 21, 8, 2, 'norm_elu'),          # dilated, tiles span several samples, ragged tail
    (64, 128, 3, 2, 21, 8, 2, 'norm_elu_res'),
    (128, 128, 3, 4, 19, 8, 2, 'norm_elu_res'),
    (128, 128, 3, 1, 19, 8, 2, 'crp2'),
    (128, 64, 3, 1, 19, 8, 2, 'up_same'),          # MSF with same-size second input
    (64, 64, 3, 1, 6, 16, 4, 'up_2x'),             # MSF with bilinear x2 resize of the second input
    (64, 32, 3, 1, 6, 32, 8, 'up_2x'),
    (32, 32, 3, 1, 2, 64, 16, 'up_2x'),
    (32, 32, 3, 1, 1, 256, 64, 'norm_elu_res'),    # large-array config: row tiles with halo at W=64
    (64, 64, 3, 1, 1100, 16, 4, 'elu_res'),        # TM=128 variant
    (32, 32, 3, 1, 3, 24, 16, 'norm_elu_res'),     # H not a power of two: generic index math
    (32, 64, 3, 1, 2, 48, 8, 'norm_elu_pool_res'),
    (64, 64, 3, 2, 5, 24, 8, 'up_same'),
    (32, 32, 3, 1, 900, 32, 8, 'crp2'),            # many tiles at 32x8: several residency rounds per CU
    (64, 64, 3, 1, 2500, 32, 8, 'norm_elu_res'),   # > 2^16 pixels x 64 channels: XCD-contiguous tile order active
    # enough tiles for the persistent tile walk of conv_wx3 (fp16-form modes): every prologue / epilogue family at 64x16
    (32, 32, 3, 1, 200, 64, 16, 'norm_elu_res'),
    (32, 32, 3, 1, 210, 64, 16, 'crp2'),
    (32, 32, 3, 1, 195, 64, 16, 'up_2x'),
    (32, 64, 3, 1, 200, 64, 16, 'norm_elu_pool_res'),
    # SBC_PRO_NORM_SELF: the launch computes the InstanceNorm++ statistics of its own input (tiles of whole samples); `stats`
    # carries the norm's parameters.  Winograd and direct kernels, one and two wave groups, pooled output, ragged last tile
    (64, 64, 3, 1, 9, 16, 4, 'selfnorm_elu_res'),
    (64, 64, 3, 1, 9, 16, 4, 'selfnorm_elu_pool_res'),
    (64, 64, 3, 1, 21, 8, 2, 'selfnorm_elu_res'),
    (64, 64, 3, 1, 5000, 8, 2, 'selfnorm_elu'),
    (64, 64, 3, 2, 21, 8, 2, 'selfnorm_elu'),
    (64, 128, 3, 2, 21, 8, 2, 'selfnorm_elu_res'),
    (128, 128, 3, 4, 19, 8, 2, 'selfnorm_elu_res'),
    (128, 128, 3, 1, 19, 8, 2, 'selfnorm_elu_res'),
    (128, 128, 3, 2, 4100, 8, 2, 'selfnorm_elu'),
    (32, 32, 3, 1, 37, 8, 8, 'selfnorm_elu_res'),
    # 64 -> 64 at the low-resolution levels without a norm prologue: in conv_mode f16x2 the direct persistent kernel (conv_dp.hip) --
    # tiles of four whole 8x2 samples (ragged last tile), of one 16x4 sample, of 8 rows of a W = 8 image (halo rows, H not a power of
    # two), every epilogue family, one tile per workgroup and several
    (64, 64, 3, 1, 21, 8, 2, 'elu_res'),
    (64, 64, 3, 1, 21, 8, 2, 'crp2'),
    (64, 64, 3, 1, 1, 8, 2, 'plain'),
    (64, 64, 3, 1, 2302, 8, 2, 'elu'),
    (64, 64, 3, 1, 9, 16, 4, 'crp2'),
    (64, 64, 3, 1, 700, 16, 4, 'elu'),
    (64, 64, 3, 1, 5, 24, 8, 'elu_res'),
    (64, 64, 3, 1, 3, 8, 8, 'crp2'),
    (64, 64, 3, 1, 2500, 32, 8, 'elu_res'),
    # ... and its 32-channel form (eight-wave workgroups) at 16- and 8-pixel rows
    (32, 32, 3, 1, 3, 24, 16, 'elu_res'),
    (32, 32, 3, 1, 7, 8, 8, 'plain'),
    (32, 32, 3, 1, 1300, 32, 8, 'elu'),
    (32, 32, 3, 1, 300, 64, 16, 'elu'),
]


@pytest.mark.parametrize('algo', ['direct', 'winograd', 'bf16x3', 'winograd_bf16x3', 'f16w', 'winograd_f16w', 'f16x2',
                                  'winograd_f16x2'])
@pytest.mark.parametrize('cin,cout,k,dil,B,H,W,mode', CONV_CASES)
def test_conv_matches_oracle(gpu, cin, cout, k, dil, B, H, W, mode, algo):
    """The three multipliers behind SBC_OP_CONV: the fp32-MFMA direct implicit GEMM (weight); when the op also carries
    weight_wino, fp32 Winograd F(2x2,3x3) for undilated 3x3 convs (shapes it does not cover fall back to direct); and,
    when it carries weight_split, the split-bf16 kernel (three exact bf16 terms per fp32 operand, six bf16 MFMAs)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import (pack_conv_weight, pack_conv_weight_f16, pack_conv_weight_f16x2,
                                                  pack_conv_weight_split, pack_conv_weight_winograd,
                                                  pack_conv_weight_winograd_f16, pack_conv_weight_winograd_f16x2,
                                                  pack_conv_weight_winograd_split, round_fp16)
    if algo.startswith('winograd') and (k != 3 or dil != 1):
        pytest.skip('Winograd F(2x2,3x3) applies to undilated 3x3 convolutions')
    rng = np.random.default_rng(hash((cin, cout, k, dil, B, H, W)) % (2 ** 31))
    x = rng.standard_normal((B, H, W, cin)).astype(F32) * 1.5 + 0.3
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(F32)
    bias = rng.standard_normal(cout).astype(F32) if 'crp' not in mode else None
    pool = 'pool' in mode
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    flags = 0
    v = x
    stats = None
    if 'norm' in mode:
        flags |= P.PRO_NORM
        agb = [(1 + 0.1 * rng.standard_normal(cin)).astype(F32), (1 + 0.1 * rng.standard_normal(cin)).astype(F32),
               (0.1 * rng.standard_normal(cin)).astype(F32)]
        stats = inorm_stats(x, *agb)
        v = (v - stats[:, None, None, 0]) * stats[:, None, None, 1] + stats[:, None, None, 2]
        if 'selfnorm' in mode:
            if algo in ('direct', 'winograd'):
                pytest.skip('SBC_PRO_NORM_SELF belongs to the matrix-core weight forms')
            flags |= P.PRO_NORM_SELF
            stats = np.concatenate(agb)              # what the launch is given: (alpha | gamma | beta)
    if 'elu' in mode:
        flags |= P.PRO_ELU
        v = O.elu(v)
    tol = TOL
    if algo.endswith('f16w'):
        # SBC_CONV_F16W: fp16 weights, activations rounded to fp16 as they enter the matrix cores, fp32 accumulation.  The
        # direct kernel is held to the same products evaluated in fp32 (fp16 x fp16 is exact in fp32; a staged value one
        # fp32 ulp off may round to the neighbouring fp16); the Winograd kernel rounds the TRANSFORMED operands instead
        # (2^-11 relative each), so it is held to that rounding level.
        w_ref, v_ref = round_fp16(w), round_fp16(v)
        tol = 1e-4 if algo == 'f16w' else 4e-3
    else:
        w_ref, v_ref = w, v
    ref = O.conv2d(v_ref.transpose(0, 3, 1, 2), w_ref, bias, dil)
    if pool:
        flags |= P.EPI_POOL
        ref = O.mean_pool2(ref)
    ref = ref.transpose(0, 2, 3, 1)
    res1 = res2 = up = None
    if 'res' in mode:
        res1 = rng.standard_normal((B, Ho, Wo, cout)).astype(F32)
        ref = ref + res1
    if mode == 'crp2':
        flags |= P.EPI_RES1_ELU
        res1 = rng.standard_normal((B, Ho, Wo, cout)).astype(F32)
        res2 = rng.standard_normal((B, Ho, Wo, cout)).astype(F32)
        ref = ref + (res2 + O.elu(res1))
    if mode.startswith('up'):
        flags |= P.EPI_UP
        uh, uw = (H, W) if mode == 'up_same' else (H // 2, W // 2)
        up = rng.standard_normal((B, uh, uw, cout)).astype(F32)
        ref = ref + O.bilinear_align_corners(up.transpose(0, 3, 1, 2), (H, W)).transpose(0, 2, 3, 1)

    d = {k_: _dev(torch, a) for k_, a in dict(x=x, w=pack_conv_weight(w), bias=bias, stats=stats, res1=res1,
                                              res2=res2, up=up).items() if a is not None}
    out = torch.full((B, Ho, Wo, cout), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.CONV, flags=flags, B=B, H=H, W=W, cin=cin, cout=cout, ksize=k, dil=dil,
                     in_=_p(d['x']), out=_p(out), weight=_p(d['w']))
    for name in ('bias', 'stats', 'res1', 'res2', 'up'):
        if name in d:
            setattr(op, name, _p(d[name]))
    if up is not None:
        op.up_h, op.up_w = up.shape[1], up.shape[2]
    if algo == 'winograd':
        ww = _dev(torch, pack_conv_weight_winograd(w))
        op.weight_wino = _p(ww)
    if algo == 'winograd_bf16x3':
        wws = _dev(torch, pack_conv_weight_winograd_split(w).view(np.float32))
        op.weight_wino_split = _p(wws)
    if algo == 'bf16x3':
        ws = _dev(torch, pack_conv_weight_split(w).view(np.float32))
        op.weight_split = _p(ws)
        op.weight = None                      # the split kernel needs nothing else
    if algo == 'f16w':
        wf = _dev(torch, pack_conv_weight_f16(w).view(np.float32))
        op.weight_split, op.weight, op.flags = _p(wf), None, flags | P.CONV_F16W
    if algo == 'winograd_f16w':
        wf = _dev(torch, pack_conv_weight_f16(w).view(np.float32))
        wwf = _dev(torch, pack_conv_weight_winograd_f16(w).view(np.float32))
        op.weight_split, op.weight_wino_split, op.weight, op.flags = _p(wf), _p(wwf), None, flags | P.CONV_F16W
    if algo == 'f16x2':
        # two fp16 terms per (scaled) operand, three fp16 MFMAs per product: held to the fp32 tolerance of the other modes
        wx = _dev(torch, pack_conv_weight_f16x2(w).view(np.float32))
        op.weight_split, op.weight, op.flags = _p(wx), None, flags | P.CONV_F16X2
    if algo == 'winograd_f16x2':
        wx = _dev(torch, pack_conv_weight_f16x2(w).view(np.float32))
        wwx = _dev(torch, pack_conv_weight_winograd_f16x2(w).view(np.float32))
        op.weight_split, op.weight_wino_split, op.weight, op.flags = _p(wx), _p(wwx), None, flags | P.CONV_F16X2
    _launch(gpu, op)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    if algo.endswith('f16x2'):
        assert _lib.range_flag() == 0
    assert rel_err(got, ref) < tol
    # and element by element for every output of at least 5 % of the largest magnitude: a 5x tighter statement than the
    # norm-wise bound implies for those elements
    # (fp16 kernels: a staged value one fp32 ulp off may round to the neighbouring fp16, a 2^-11 change of one product)
    assert rel_err_elementwise(got, ref, 0.05) < {'f16w': 4e-3, 'winograd_f16w': 4e-2}.get(algo, 4 * tol)


@pytest.mark.parametrize('cin,cout,B', [(128, 128, 53), (64, 128, 53), (128, 64, 53), (64, 64, 120)])
def test_conv_two_wave_groups_are_race_free(gpu, cin, cout, B):
    """Launches with fewer tiles than CUs run conv_wx3 with two wave groups per workgroup (output blocks dealt between
    them, T planes per group).  The shortest epilogue (no bias / residual) leaves the least time between a group's finish
    reads and its next T-plane writes: repeat the launch and require bit-identical, correct outputs every time."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_winograd_split
    rng = np.random.default_rng(cin + cout)
    H, W = 8, 2
    x = rng.standard_normal((B, H, W, cin)).astype(F32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(F32)
    ref = O.conv2d(x.transpose(0, 3, 1, 2), w, None, 1).transpose(0, 2, 3, 1)
    dx, dw = _dev(torch, x), _dev(torch, pack_conv_weight_winograd_split(w).view(np.float32))
    outs = [torch.full((B, H, W, cout), float('nan'), dtype=torch.float32, device='cuda') for _ in range(2)]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    first = None
    for rep in range(300):
        out = outs[rep & 1]
        op = _lib.sbc_op(kind=P.CONV, B=B, H=H, W=W, cin=cin, cout=cout, ksize=3, dil=1, in_=_p(dx), out=_p(out),
                         weight_wino_split=_p(dw), weight_split=_p(dw))
        _lib.check(_lib.lib().sbc_op_launch(C.byref(op), stream))
        if rep < 2 or rep % 50 == 49:
            torch.cuda.synchronize()
            got = out.cpu().numpy()
            if first is None:
                first = got
                assert rel_err(got, ref) < TOL
            assert np.array_equal(got, first), rep
    # every launch of the burst: accumulate a checksum on the device instead of copying 300 tensors back
    sums = []
    for rep in range(200):
        out = outs[rep & 1]
        out.fill_(float('nan'))
        op = _lib.sbc_op(kind=P.CONV, B=B, H=H, W=W, cin=cin, cout=cout, ksize=3, dil=1, in_=_p(dx), out=_p(out),
                         weight_wino_split=_p(dw), weight_split=_p(dw))
        _lib.check(_lib.lib().sbc_op_launch(C.byref(op), stream))
        sums.append((out.view(torch.int32).to(torch.int64)).sum())
    sums = torch.stack(sums).cpu().numpy()
    assert (sums == sums[0]).all()


def test_conv_rejects_unsupported_shapes(gpu):
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    t = torch.zeros(16, device='cuda')
    op = _lib.sbc_op(kind=P.CONV, B=1, H=8, W=8, cin=48, cout=32, ksize=3, dil=1, in_=_p(t), out=_p(t), weight=_p(t))
    rc = _lib.lib().sbc_op_launch(C.byref(op), None)
    assert rc == -3 and b'no kernel' in _lib.lib().sbc_last_error()
    op = _lib.sbc_op(kind=P.CONV, B=1, H=8, W=8, cin=32, cout=32, ksize=5, dil=1, in_=_p(t), out=_p(t), weight=_p(t))
    assert _lib.lib().sbc_op_launch(C.byref(op), None) == -1


@pytest.mark.parametrize('C_,B,H,W', [(32, 5, 64, 16), (64, 3, 32, 8), (64, 4, 16, 4), (128, 7, 8, 2), (32, 1, 256, 64)])
def test_inorm_stats_matches_oracle(gpu, C_, B, H, W):
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(C_ + B)
    x = (rng.standard_normal((B, H, W, C_)) * rng.uniform(0.5, 2, C_) + rng.standard_normal(C_) * 3).astype(F32)
    agb = np.stack(((1 + 0.1 * rng.standard_normal(C_)), (1 + 0.1 * rng.standard_normal(C_)),
                    0.1 * rng.standard_normal(C_))).astype(F32)
    dx, dagb = _dev(torch, x), _dev(torch, agb)
    out = torch.zeros(B, 3, C_, device='cuda')
    _launch(gpu, _lib.sbc_op(kind=P.INORM_STATS, B=B, H=H, W=W, cin=C_, cout=C_, in_=_p(dx), out=_p(out), weight=_p(dagb)))
    ref = inorm_stats(x, agb[0], agb[1], agb[2])
    got = out.cpu().numpy()
    # applying both to the data is the meaningful comparison
    y_ref = (x - ref[:, None, None, 0]) * ref[:, None, None, 1] + ref[:, None, None, 2]
    y_got = (x - got[:, None, None, 0]) * got[:, None, None, 1] + got[:, None, None, 2]
    assert rel_err(y_got, y_ref) < 1e-5
    assert rel_err(y_ref, O.instance_norm_plus(x.transpose(0, 3, 1, 2), agb[0], agb[1], agb[2]).transpose(0, 2, 3, 1)) < 1e-5


@pytest.mark.parametrize('C_,B,H,W,elu', [(32, 3, 64, 16, True), (64, 5, 16, 4, False), (128, 9, 8, 2, True),
                                          (64, 4, 32, 8, True), (32, 2, 32, 8, False), (32, 1, 256, 64, True),
                                          (64, 2, 128, 32, False), (128, 3, 32, 16, True), (32, 2, 48, 24, True)])
def test_maxpool5_matches_oracle(gpu, C_, B, H, W, elu):
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    x = np.random.default_rng(3).standard_normal((B, H, W, C_)).astype(F32)
    dx = _dev(torch, x)
    out = torch.zeros_like(dx)
    _launch(gpu, _lib.sbc_op(kind=P.MAXPOOL5, flags=P.PRO_ELU if elu else 0, B=B, H=H, W=W, cin=C_, cout=C_,
                             in_=_p(dx), out=_p(out)))
    ref = O.max_pool5(x.transpose(0, 3, 1, 2)).transpose(0, 2, 3, 1)
    ref = O.elu(ref) if elu else ref
    assert rel_err(out.cpu().numpy(), ref) < 1e-6


@pytest.mark.parametrize('B,H,W', [(3, 64, 16), (1, 256, 64), (5, 8, 8)])
def test_begin_conv_matches_oracle(gpu, B, H, W):
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(4)
    x = rng.standard_normal((B, H, W, 2)).astype(F32)
    w = (rng.standard_normal((32, 2, 3, 3)) / 4).astype(F32)
    b = rng.standard_normal(32).astype(F32)
    dx, dw, db = _dev(torch, x), _dev(torch, w), _dev(torch, b)
    out = torch.zeros(B, H, W, 32, device='cuda')
    _launch(gpu, _lib.sbc_op(kind=P.BEGIN_CONV, B=B, H=H, W=W, cin=2, cout=32, ksize=3, dil=1, in_=_p(dx), out=_p(out),
                             weight=_p(dw), bias=_p(db)))
    ref = O.conv2d((2 * x - 1).transpose(0, 3, 1, 2), w, b).transpose(0, 2, 3, 1)
    assert rel_err(out.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize('B,H,W', [(5, 64, 16), (1, 256, 64), (40, 8, 8)])
def test_end_conv_matches_oracle(gpu, B, H, W):
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, H, W, 32)).astype(F32)
    agb = np.stack(((1 + 0.1 * rng.standard_normal(32)), (1 + 0.1 * rng.standard_normal(32)),
                    0.1 * rng.standard_normal(32))).astype(F32)
    stats = inorm_stats(x, agb[0], agb[1], agb[2])
    w = (rng.standard_normal((2, 32, 3, 3)) / 17).astype(F32)
    b = rng.standard_normal(2).astype(F32)
    sigmas = np.exp(np.linspace(np.log(39.15), np.log(3.6e-4), 50)).astype(F32)
    labels = rng.integers(0, 50, B)
    d = [_dev(torch, a) for a in (x, stats, w, b, sigmas, labels.astype(np.int64))]
    out = torch.zeros(B, H, W, 2, device='cuda')
    ext = _lib.sbc_endconv(sigmas=_p(d[4]), labels=_p(d[5]))
    _launch(gpu, _lib.sbc_op(kind=P.END_CONV, B=B, H=H, W=W, cin=32, cout=2, ksize=3, dil=1, in_=_p(d[0]), out=_p(out),
                             weight=_p(d[2]), bias=_p(d[3]), stats=_p(d[1]), ext=C.cast(C.pointer(ext), C.c_void_p)))
    v = O.elu((x - stats[:, None, None, 0]) * stats[:, None, None, 1] + stats[:, None, None, 2])
    ref = O.conv2d(v.transpose(0, 3, 1, 2), w, b).transpose(0, 2, 3, 1) / sigmas[labels][:, None, None, None]
    assert rel_err(out.cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize('B,H,W', [(5, 64, 16), (1, 64, 16), (300, 64, 16), (4, 32, 32), (3, 16, 64), (2, 128, 8)])
def test_end_conv_with_its_own_statistics_matches_oracle(gpu, B, H, W):
    """SBC_OP_END_CONV with SBC_PRO_NORM_SELF (csrc/ops.hip: end_conv_self_kernel): `stats` is the normalizer's (alpha | gamma | beta) and
    the launch forms the InstanceNorm++ statistics itself -- persistent workgroups that hold whole samples, the tensor read once --
    against the oracle's normalisation + ELU + convolution + division by sigma (ncsnv2.py:291-298), with per-channel offsets and scales,
    more samples than workgroups, and results that do not depend on the batch."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(B + H)
    # (offsets of a few standard deviations: the fp32 oracle's own mean subtraction is good to ~1e-6 of the normalised values there)
    x = (rng.standard_normal((B, H, W, 32)) * (0.3 + rng.random(32) * 3) + 2 * rng.standard_normal(32)).astype(F32)
    agb = np.stack(((1 + 0.1 * rng.standard_normal(32)), (1 + 0.1 * rng.standard_normal(32)), 0.1 * rng.standard_normal(32))).astype(F32)
    w = (rng.standard_normal((2, 32, 3, 3)) / 17).astype(F32)
    b = rng.standard_normal(2).astype(F32)
    sigmas = np.exp(np.linspace(np.log(39.15), np.log(3.6e-4), 50)).astype(F32)
    labels = rng.integers(0, 50, B)
    d = [_dev(torch, a) for a in (x, agb, w, b, sigmas, labels.astype(np.int64))]

    def run(lo, hi):
        out = torch.full((hi - lo, H, W, 2), float('nan'), device='cuda')
        ext = _lib.sbc_endconv(sigmas=_p(d[4]), labels=_p(d[5][lo:hi]))
        _launch(gpu, _lib.sbc_op(kind=P.END_CONV, flags=P.PRO_NORM_SELF, B=hi - lo, H=H, W=W, cin=32, cout=2, ksize=3, dil=1, in_=_p(d[0][lo:hi]),
                                 out=_p(out), weight=_p(d[2]), bias=_p(d[3]), stats=_p(d[1]), ext=C.cast(C.pointer(ext), C.c_void_p)))
        return out.cpu().numpy()
    got = run(0, B)
    v = O.elu(O.instance_norm_plus(x.transpose(0, 3, 1, 2), agb[0], agb[1], agb[2]))
    ref = O.conv2d(v, w, b).transpose(0, 2, 3, 1) / sigmas[labels][:, None, None, None]
    assert np.isfinite(got).all() and rel_err(got, ref) < 1e-5
    if B > 1:
        assert np.array_equal(run(B // 2, B), got[B // 2:])                    # a sample's numbers do not depend on its neighbours
    with pytest.raises(_lib.SbcError):                                           # shapes the kernel does not take are refused, not mangled
        out = torch.zeros(2, 8, 8, 2, device='cuda')
        ext = _lib.sbc_endconv(sigmas=_p(d[4]), labels=_p(d[5]))
        _launch(gpu, _lib.sbc_op(kind=P.END_CONV, flags=P.PRO_NORM_SELF, B=2, H=8, W=8, cin=32, cout=2, ksize=3, dil=1, in_=_p(d[0]), out=_p(out),
                                 weight=_p(d[2]), bias=_p(d[3]), stats=_p(d[1]), ext=C.cast(C.pointer(ext), C.c_void_p)))


@pytest.mark.parametrize('wino', [False, True])
def test_f16x2_range_flag(gpu, wino):
    """conv_mode f16x2 stages activations as two fp16 terms: an activation beyond 16000 could overflow the
    high term (or the Winograd transform's 4-term sums), so the kernels raise the device's range flag instead of returning
    silently wrong numbers -- and stay quiet, and accurate, just below the limit."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
    rng = np.random.default_rng(5)
    B, H, W, c = 2, 16, 16, 32
    w = (rng.standard_normal((c, c, 3, 3)) / 17).astype(F32)
    wx = _dev(torch, pack_conv_weight_f16x2(w).view(np.float32))
    wwx = _dev(torch, pack_conv_weight_winograd_f16x2(w).view(np.float32))
    _lib.range_flag()                                        # clear
    for peak, expect in ((15000.0, 0), (17000.0, 1)):
        x = rng.standard_normal((B, H, W, c)).astype(F32)
        x[1, 7, 9, 3] = peak
        out = torch.full((B, H, W, c), float('nan'), dtype=torch.float32, device='cuda')
        dx = _dev(torch, x)
        op = _lib.sbc_op(kind=P.CONV, flags=P.CONV_F16X2, B=B, H=H, W=W, cin=c, cout=c, ksize=3, dil=1, in_=_p(dx), out=_p(out),
                         weight_split=_p(wx))
        if wino:
            op.weight_wino_split = _p(wwx)
        _launch(gpu, op)
        assert _lib.range_flag() == expect                   # reading resets
        assert _lib.range_flag() == 0
        if not expect:
            ref = O.conv2d(x.transpose(0, 3, 1, 2), w, None, 1).transpose(0, 2, 3, 1)
            assert rel_err(out.cpu().numpy(), ref) < TOL
    with pytest.raises(_lib.SbcError):
        op.flags |= P.CONV_F16W
        _launch(gpu, op)


PAIR_CASES = [(3, 64, 16), (130, 64, 16), (600, 64, 16),     # (130 x 64x16 = 1040 tiles and up: the pipelined kernel over row rings)
              (1031, 32, 16), (2050, 16, 16), (4100, 8, 16),   # the same kernel with 4 / 2 / 1 tiles per sample
              (1, 8, 16), (5, 32, 8), (70, 32, 8), (2, 16, 16), (1, 256, 64), (3, 12, 64), (40, 16, 64),
              (2, 128, 32), (33, 8, 32), (2, 64, 16, 64), (35, 16, 16, 64), (2, 128, 32, 64), (17, 12, 32, 64)]     # (.., C = 64)


@pytest.mark.parametrize('mode', ['f16x2', 'f16w'])
@pytest.mark.parametrize('case', PAIR_CASES)
def test_conv_pair_matches_oracle(gpu, case, mode):
    """SBC_OP_CONV_PAIR: one RCU block, out = x + conv2(ELU(conv1(ELU(x)))) (layers.py:126-134), with the intermediate kept in
    LDS (csrc/conv_pair.hip) -- against the oracle's two convolutions, and against the two SBC_OP_CONV launches it replaces."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16, pack_conv_weight_f16x2, round_fp16
    B, H, W = case[:3]
    Cc = case[3] if len(case) > 3 else 32
    if (W >= 32 or Cc == 64) and mode != 'f16w':
        pytest.skip('32- / 64-pixel rows and 64 channels: the pair kernel exists in the fp16-weight mode only (BASELINE config 5)')
    rng = np.random.default_rng(B * 1000 + H + W + Cc)
    x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(F32)
    w1 = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(F32)
    w2 = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc) * 0.05).astype(F32)   # a different weight scale per convolution
    if mode == 'f16x2':
        pack, flag, tol = pack_conv_weight_f16x2, P.CONV_F16X2, TOL
        t = O.conv2d(O.elu(x).transpose(0, 3, 1, 2), w1, None, 1)
        ref = x + O.conv2d(O.elu(t), w2, None, 1).transpose(0, 2, 3, 1)
    else:
        pack, flag, tol = pack_conv_weight_f16, P.CONV_F16W, 2e-4
        t = O.conv2d(round_fp16(O.elu(x)).transpose(0, 3, 1, 2), round_fp16(w1), None, 1)
        ref = x + O.conv2d(round_fp16(O.elu(t)), round_fp16(w2), None, 1).transpose(0, 2, 3, 1)
    dx = _dev(torch, x)
    d1, d2 = _dev(torch, pack(w1).view(np.float32)), _dev(torch, pack(w2).view(np.float32))
    out = torch.full((B, H, W, Cc), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.CONV_PAIR, flags=flag, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(dx), out=_p(out),
                     weight_split=_p(d1), weight2_split=_p(d2))
    _launch(gpu, op)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    conv_part, ref_part = got - x, ref - x                          # the residual must not mask the convolution's error
    assert rel_err(conv_part, ref_part) < tol
    assert _lib.range_flag() == 0
    # the two launches it replaces
    mid = torch.empty_like(out)
    out2 = torch.empty_like(out)
    a = _lib.sbc_op(kind=P.CONV, flags=flag | P.PRO_ELU, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(dx), out=_p(mid),
                    weight_split=_p(d1))
    b = _lib.sbc_op(kind=P.CONV, flags=flag | P.PRO_ELU, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(mid), out=_p(out2),
                    weight_split=_p(d2), res1=_p(dx))
    _launch(gpu, a)
    _launch(gpu, b)
    assert rel_err(got - x, out2.cpu().numpy() - x) < tol
    if W == 16 and Cc == 32 and B * (H // 8) >= 1024:
        # the pipelined kernel (contiguous runs of tiles per workgroup, halo rows kept in LDS rings; launches of 1024 tiles or more) adds
        # the same products in the same order as the tile-at-a-time kernel small batches get: identical bit for bit
        piece = 1023 // (H // 8)
        for lo in list(range(0, B, piece))[:6]:
            hi = min(lo + piece, B)
            assert (hi - lo) * (H // 8) < 1024
            part = torch.full((hi - lo, H, W, Cc), float('nan'), dtype=torch.float32, device='cuda')
            sub = _lib.sbc_op(kind=P.CONV_PAIR, flags=flag, B=hi - lo, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(dx[lo:hi]),
                              out=_p(part), weight_split=_p(d1), weight2_split=_p(d2))
            _launch(gpu, sub)
            assert np.array_equal(part.cpu().numpy(), got[lo:hi])


@pytest.mark.parametrize('mode', ['f16x2', 'f16w'])
@pytest.mark.parametrize('stage', ['pool_elu', 'pool_crp2'])
@pytest.mark.parametrize('B,H', [(3, 64), (130, 64), (600, 64), (1, 8), (5, 16), (37, 24)])
def test_conv_pool_matches_oracle(gpu, B, H, stage, mode):
    """SBC_OP_CONV_POOL: one CRP stage, out = conv3x3(ELU?(MaxPool5x5(x))) [+ (res2 + ELU(res1))] (layers.py:76-83), with the pooled
    tensor kept in LDS (csrc/conv_pair.hip: conv_pool_kernel) -- against the oracle's max pool + convolution, and against the
    SBC_OP_MAXPOOL5 + SBC_OP_CONV launches it replaces.  Both stages of the block: `pool_elu` (ELU of the pooled input, no
    residual) and `pool_crp2` (the running sum path + (path0 + ELU(x)) in the epilogue)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import (pack_conv_weight_f16, pack_conv_weight_f16x2, pack_conv_weight_winograd_f16,
                                                  pack_conv_weight_winograd_f16x2, round_fp16)
    W, Cc = 16, 32
    rng = np.random.default_rng(B * 100 + H)
    x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(F32)
    w = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(F32)
    elu = stage == 'pool_elu'
    v = O.max_pool5(x.transpose(0, 3, 1, 2))
    if elu:
        v = O.elu(v)
    res1 = res2 = None
    flags = P.PRO_ELU if elu else 0
    if mode == 'f16x2':
        pack, packw, mflag, tol = pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2, P.CONV_F16X2, TOL
        ref = O.conv2d(v, w, None, 1).transpose(0, 2, 3, 1)
    else:
        pack, packw, mflag, tol = pack_conv_weight_f16, pack_conv_weight_winograd_f16, P.CONV_F16W, 2e-4
        ref = O.conv2d(round_fp16(v), round_fp16(w), None, 1).transpose(0, 2, 3, 1)
    conv_ref = ref
    if not elu:
        flags |= P.EPI_RES1_ELU
        res1 = rng.standard_normal((B, H, W, Cc)).astype(F32)
        res2 = rng.standard_normal((B, H, W, Cc)).astype(F32)
        ref = ref + (res2 + O.elu(res1))
    dx, dw = _dev(torch, x), _dev(torch, pack(w).view(np.float32))
    out = torch.full((B, H, W, Cc), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.CONV_POOL, flags=flags | mflag, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(dx), out=_p(out),
                     weight_split=_p(dw))
    keep = []
    if res1 is not None:
        keep = [_dev(torch, res1), _dev(torch, res2)]
        op.res1, op.res2 = _p(keep[0]), _p(keep[1])
    _launch(gpu, op)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got - (ref - conv_ref), conv_ref) < tol            # (the residual operands must not mask the convolution's error)
    assert _lib.range_flag() == 0
    # the two launches it replaces
    pooled = torch.empty_like(out)
    out2 = torch.empty_like(out)
    _launch(gpu, _lib.sbc_op(kind=P.MAXPOOL5, flags=P.PRO_ELU if elu else 0, B=B, H=H, W=W, cin=Cc, cout=Cc, in_=_p(dx), out=_p(pooled)))
    dww = _dev(torch, packw(w).view(np.float32))
    b = _lib.sbc_op(kind=P.CONV, flags=(flags & P.EPI_RES1_ELU) | mflag, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(pooled),
                    out=_p(out2), weight_split=_p(dw), weight_wino_split=_p(dww))
    if res1 is not None:
        b.res1, b.res2 = _p(keep[0]), _p(keep[1])
    _launch(gpu, b)
    # (f16w: the Winograd launch it replaces rounds the TRANSFORMED operands to fp16, this one the operands themselves)
    assert rel_err(got, out2.cpu().numpy()) < (tol if mode == 'f16x2' else 4e-3)
    with pytest.raises(_lib.SbcError):                                # shapes it does not take are refused, not mangled
        _launch(gpu, _lib.sbc_op(kind=P.CONV_POOL, flags=mflag, B=B, H=H, W=8, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(dx), out=_p(out),
                                 weight_split=_p(dw)))


@pytest.mark.parametrize('B,H,W', [(1, 64, 16), (3, 256, 64), (300, 64, 16)])
def test_direct_conv_writes_tile_moments_in_f16w(gpu, B, H, W):
    """Round 6, conv_mode f16w (BASELINE config 5): ONE matrix instruction per product, so the direct kernel (csrc/conv_x3.hip) is faster
    than Winograd for every layer and takes them all -- including the 32-channel producers of tile moments (SBC_EPI_MOMENTS_OUT), whose
    (mean, M2) per 128-pixel tile now come out of its epilogue (conv_epilogue.h): both tile sizes (128 pixels at small batches, 256 at
    large ones), with a residual operand, against numpy on the launch's own output; and the output against the oracle."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16, pack_conv_weight_winograd_f16, round_fp16
    Cc = 32
    rng = np.random.default_rng(B + H)
    x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(F32)
    res = rng.standard_normal((B, H, W, Cc)).astype(F32)
    w = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(F32)
    bias = (0.3 * rng.standard_normal(Cc)).astype(F32)
    d = {k: _dev(torch, a) for k, a in dict(x=x, res=res, w=pack_conv_weight_f16(w).view(np.float32), ww=pack_conv_weight_winograd_f16(w).view(np.float32),
                                              b=bias).items()}
    out = torch.full((B, H, W, Cc), float('nan'), dtype=torch.float32, device='cuda')
    nt = H * W // 128
    pm = torch.full((B, nt, Cc, 2), float('nan'), dtype=torch.float32, device='cuda')
    # (the Winograd form rides along as the host binds it: the f16w dispatch must not take it)
    op = _lib.sbc_op(kind=P.CONV, flags=P.CONV_F16W | P.PRO_ELU | P.EPI_MOMENTS_OUT, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(d['x']),
                     out=_p(out), bias=_p(d['b']), res1=_p(d['res']), weight_split=_p(d['w']), weight_wino_split=_p(d['ww']), aux=_p(pm))
    _launch(gpu, op)
    got = out.cpu().numpy()
    nref = min(B, 4)
    ref = O.conv2d(round_fp16(O.elu(x[:nref])).transpose(0, 3, 1, 2), round_fp16(w), bias, 1).transpose(0, 2, 3, 1) + res[:nref]
    assert np.isfinite(got).all() and rel_err(got[:nref], ref) < 1e-4          # (the direct f16w tolerance of test_conv_matches_oracle)
    tiles = got.astype(np.float64).reshape(B, nt, 128, Cc)
    gm = pm.cpu().numpy()
    assert np.isfinite(gm).all()
    assert np.abs(gm[..., 0] - tiles.mean(2)).max() < 1e-5 * max(1.0, np.abs(tiles.mean(2)).max())
    assert rel_err(gm[..., 1], ((tiles - tiles.mean(2, keepdims=True)) ** 2).sum(2)) < 1e-5
    # the same launch without moments: the same output, bit for bit
    out2 = torch.empty_like(out)
    op.flags, op.aux, op.out = P.CONV_F16W | P.PRO_ELU, None, _p(out2)
    _launch(gpu, op)
    assert torch.equal(out, out2)


@pytest.mark.parametrize('B', [1, 3, 700])
def test_direct_conv_with_norm_prologue_and_tile_moments(gpu, B):
    """Round 6: the 64 -> 64 layers at 8-pixel rows with an InstanceNorm++ prologue and a tile-moment output (res2.1's convolutions,
    res3.0.conv1 of a 64 x 16 array) run on the direct persistent kernel (csrc/conv_dp.hip, its NM instantiation) instead of Winograd:
    norm -> ELU -> conv + bias + res1 against the oracle, the (mean, M2) of the output's 128-pixel tiles -- two tiles of the kernel, merged
    in registers / LDS by the workgroup that takes them back to back -- against numpy on the launch's own output, and batch independence
    bit for bit (a piece of the batch alone: other run boundaries, other workgroups)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
    H, W, Cc = 32, 8, 64
    rng = np.random.default_rng(500 + B)
    x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(F32)
    res = rng.standard_normal((B, H, W, Cc)).astype(F32)
    w = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(F32)
    bias = (0.3 * rng.standard_normal(Cc)).astype(F32)
    agb = [(1 + 0.1 * rng.standard_normal(Cc)).astype(F32), (1 + 0.1 * rng.standard_normal(Cc)).astype(F32), (0.1 * rng.standard_normal(Cc)).astype(F32)]
    st = inorm_stats(x, *agb)
    v = O.elu((x - st[:, None, None, 0]) * st[:, None, None, 1] + st[:, None, None, 2])
    nref = min(B, 4)
    ref = O.conv2d(v[:nref].transpose(0, 3, 1, 2), w, bias, 1).transpose(0, 2, 3, 1) + res[:nref]
    d = {k: _dev(torch, a) for k, a in dict(x=x, res=res, st=st, b=bias, w=pack_conv_weight_f16x2(w).view(np.float32),
                                              ww=pack_conv_weight_winograd_f16x2(w).view(np.float32)).items()}
    nt = H * W // 128

    def run(lo, hi, moments=True):
        out = torch.full((hi - lo, H, W, Cc), float('nan'), dtype=torch.float32, device='cuda')
        pm = torch.full((hi - lo, nt, Cc, 2), float('nan'), dtype=torch.float32, device='cuda')
        op = _lib.sbc_op(kind=P.CONV, flags=P.CONV_F16X2 | P.PRO_NORM | P.PRO_ELU | (P.EPI_MOMENTS_OUT if moments else 0), B=hi - lo, H=H, W=W, cin=Cc, cout=Cc,
                         ksize=3, dil=1, in_=_p(d['x'][lo:hi]), out=_p(out), stats=_p(d['st'][lo:hi]), bias=_p(d['b']), res1=_p(d['res'][lo:hi]),
                         weight_split=_p(d['w']), weight_wino_split=_p(d['ww']), aux=_p(pm) if moments else None)
        _launch(gpu, op)
        return out.cpu().numpy(), pm.cpu().numpy()
    got, gm = run(0, B)
    assert np.isfinite(got).all() and _lib.range_flag() == 0
    assert rel_err(got[:nref] - res[:nref], ref - res[:nref]) < TOL
    tiles = got.astype(np.float64).reshape(B, nt, 128, Cc)
    assert np.isfinite(gm).all()
    assert np.abs(gm[..., 0] - tiles.mean(2)).max() < 1e-5 * max(1.0, np.abs(tiles.mean(2)).max())
    assert rel_err(gm[..., 1], ((tiles - tiles.mean(2, keepdims=True)) ** 2).sum(2)) < 1e-5
    plain, _ = run(0, B, moments=False)                                  # without the moment output: the same numbers
    assert np.array_equal(plain, got)
    if B > 2:
        lo, hi = B // 3, B // 3 + max(1, B // 5)
        part, pmp = run(lo, hi)
        assert np.array_equal(part, got[lo:hi]) and np.array_equal(pmp, gm[lo:hi])


@pytest.mark.parametrize('B', [1, 3, 300, 700])
def test_res_block_matches_oracle(gpu, B):
    """SBC_OP_RES_BLOCK: one whole ResidualBlock without resampling (layers.py:443-456) in one launch -- a workgroup owns a sample and
    forms the InstanceNorm++ statistics of the intermediate itself (csrc/conv_res.hip) -- against the oracle's norm -> ELU -> conv ->
    norm -> ELU -> conv -> + x, against the three records it replaces, and the tile moments of its output against numpy.  Batches
    below and above the number of CUs (one / several samples per workgroup)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
    H, W, Cc = 64, 16, 32
    rng = np.random.default_rng(77 + B)
    x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(F32)
    w1, w2 = [(rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(F32) for _ in range(2)]
    b1, b2 = [(0.3 * rng.standard_normal(Cc)).astype(F32) for _ in range(2)]
    agb1, agb2 = [[(1 + 0.1 * rng.standard_normal(Cc)).astype(F32), (1 + 0.1 * rng.standard_normal(Cc)).astype(F32),
                   (0.1 * rng.standard_normal(Cc)).astype(F32)] for _ in range(2)]
    norm = lambda v, st: (v - st[:, None, None, 0]) * st[:, None, None, 1] + st[:, None, None, 2]
    s1 = inorm_stats(x, *agb1)
    t = O.conv2d(O.elu(norm(x, s1)).transpose(0, 3, 1, 2), w1, b1, 1).transpose(0, 2, 3, 1)
    s2 = inorm_stats(t, *agb2)
    main = O.conv2d(O.elu(norm(t, s2)).transpose(0, 3, 1, 2), w2, b2, 1).transpose(0, 2, 3, 1)
    ref = x + main
    d = {k: _dev(torch, a) for k, a in dict(x=x, s1=s1, w1=pack_conv_weight_f16x2(w1).view(np.float32),
                                              w2=pack_conv_weight_f16x2(w2).view(np.float32), b1=b1, b2=b2,
                                              n2=np.concatenate(agb2)).items()}
    out = torch.full((B, H, W, Cc), float('nan'), dtype=torch.float32, device='cuda')
    pm = torch.full((B, 8, Cc, 2), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.RES_BLOCK, flags=P.CONV_F16X2 | P.EPI_MOMENTS_OUT, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1,
                     in_=_p(d['x']), out=_p(out), stats=_p(d['s1']), weight_split=_p(d['w1']), weight2_split=_p(d['w2']),
                     bias=_p(d['b1']), bias2=_p(d['b2']), norm2=_p(d['n2']), aux=_p(pm))
    _launch(gpu, op)
    got = out.cpu().numpy()
    assert np.isfinite(got).all() and _lib.range_flag() == 0
    assert rel_err(got - x, main) < TOL                               # (the residual operand must not mask the error of the main path)
    assert rel_err_elementwise(got - x, main, 0.05) < 4 * TOL
    # tile moments of the output: (mean, M2) of each channel over the 8 tiles of 128 pixels
    tiles = got.astype(np.float64).reshape(B, 8, 128, Cc)
    gm = pm.cpu().numpy()
    assert np.abs(gm[..., 0] - tiles.mean(2)).max() < 1e-5 * max(1.0, np.abs(tiles.mean(2)).max())
    assert rel_err(gm[..., 1], ((tiles - tiles.mean(2, keepdims=True)) ** 2).sum(2)) < 1e-5
    # the records it replaces: conv (norm, ELU) -> statistics -> conv (norm, ELU, + x)
    wino = [_dev(torch, pack_conv_weight_winograd_f16x2(w).view(np.float32)) for w in (w1, w2)]
    tt, st2, out2 = torch.empty_like(out), torch.empty((B, 3, Cc), dtype=torch.float32, device='cuda'), torch.empty_like(out)
    fl = P.CONV_F16X2 | P.PRO_NORM | P.PRO_ELU
    _launch(gpu, _lib.sbc_op(kind=P.CONV, flags=fl, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(d['x']), out=_p(tt),
                             stats=_p(d['s1']), bias=_p(d['b1']), weight_split=_p(d['w1']), weight_wino_split=_p(wino[0])))
    _launch(gpu, _lib.sbc_op(kind=P.INORM_STATS, B=B, H=H, W=W, cin=Cc, cout=Cc, in_=_p(tt), out=_p(st2), weight=_p(d['n2'])))
    _launch(gpu, _lib.sbc_op(kind=P.CONV, flags=fl, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(tt), out=_p(out2),
                             stats=_p(st2), bias=_p(d['b2']), res1=_p(d['x']), weight_split=_p(d['w2']), weight_wino_split=_p(wino[1])))
    assert rel_err(got - x, out2.cpu().numpy() - x) < TOL
    # batch independence, bit for bit: sample 0 alone
    if B > 1:
        out1 = torch.empty((1, H, W, Cc), dtype=torch.float32, device='cuda')
        op.B, op.out, op.aux, op.flags = 1, _p(out1), None, P.CONV_F16X2
        _launch(gpu, op)
        assert torch.equal(out1[0], out[0])
    with pytest.raises(_lib.SbcError):                                # shapes it does not take are refused
        _launch(gpu, _lib.sbc_op(kind=P.RES_BLOCK, flags=P.CONV_F16X2, B=B, H=32, W=16, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(d['x']),
                                 out=_p(out), stats=_p(d['s1']), weight_split=_p(d['w1']), weight2_split=_p(d['w2']), bias=_p(d['b1']),
                                 bias2=_p(d['b2']), norm2=_p(d['n2'])))


@pytest.mark.parametrize('wino', [True, False])
@pytest.mark.parametrize('mode', ['f16x2', 'bf16x3', 'f16w'])
@pytest.mark.parametrize('cin,cout', [(32, 32), (32, 64), (64, 32), (64, 64), (64, 128), (128, 64), (128, 128)])
def test_winograd_variants_are_accurate_and_batch_size_independent(gpu, cin, cout, mode, wino):
    """Which instantiation of the Winograd kernel runs (one or two wave groups, one or two output blocks per phase) -- and which
    tile size of the direct kernel -- depends on the number of tiles in the launch, i.e. on the batch size.  Every one of them must (a) agree with a float64 convolution and
    (b) return, for a sample, the same bits whatever the batch around it -- the sub-batch streams of a run rely on that.  (Round 3:
    the two-block 128 -> 64 instantiation of the two-term fp16 mode, which only runs beyond 256 tiles, summed the last 16 input
    channels wrongly; no test had a launch that large.)"""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import (pack_conv_weight_f16, pack_conv_weight_f16x2, pack_conv_weight_split,
                                                  pack_conv_weight_winograd_f16, pack_conv_weight_winograd_f16x2,
                                                  pack_conv_weight_winograd_split)
    rng = np.random.default_rng(cin * 131 + cout)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(F32)
    pk, pkw, flag, tol = {'f16x2': (pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2, P.CONV_F16X2, 1e-5),
                          'bf16x3': (pack_conv_weight_split, pack_conv_weight_winograd_split, 0, 1e-5),
                          'f16w': (pack_conv_weight_f16, pack_conv_weight_winograd_f16, P.CONV_F16W, 1e-2)}[mode]
    ws, ww = _dev(torch, pk(w).view(np.float32)), _dev(torch, pkw(w).view(np.float32))
    wd = torch.from_numpy(w).cuda().double()
    for H, W, sizes in ((8, 2, (2400, 1000, 100)), (16, 4, (600, 50)), (32, 8, (300, 10))):
        x = torch.randn(sizes[0], H, W, cin, device='cuda', generator=torch.Generator('cuda').manual_seed(H)) * 1.5 + 0.3
        ref = torch.nn.functional.conv2d(torch.nn.functional.elu(x).permute(0, 3, 1, 2).double(), wd, padding=1).permute(0, 2, 3, 1)
        outs = []
        for B in sizes:
            out = torch.full((B, H, W, cout), float('nan'), dtype=torch.float32, device='cuda')
            op = _lib.sbc_op(kind=P.CONV, flags=flag | P.PRO_ELU, B=B, H=H, W=W, cin=cin, cout=cout, ksize=3, dil=1, in_=_p(x), out=_p(out),
                             weight_split=_p(ws))
            if wino:                                   # (without it: the direct kernels and their tile sizes, chosen by the pixel count)
                op.weight_wino_split = _p(ww)
            _launch(gpu, op)
            # (direct kernels: one fp32 chain over 9 taps x cin / 16 steps x terms -- a few 1e-6 more than Winograd's 16 short chains)
            assert float((out.double() - ref[:B]).abs().max()) < (tol if wino else 3 * tol), (H, W, B)
            outs.append(out)
        for out in outs[1:]:
            assert torch.equal(out, outs[0][:out.shape[0]]), (H, W, out.shape[0])


# ---- conv_mode f16x2 outside O(1) activations (round 4): scale, guard in both directions, library calibration -------------------
def _f64_conv(torch, v, w, dil):
    """float64 convolution of an NHWC float32 tensor ``v`` (already through the prologue) with a torch-layout weight."""
    return torch.nn.functional.conv2d(v.permute(0, 3, 1, 2).double(), torch.from_numpy(w).cuda().double(), padding=dil,
                                      dilation=dil).permute(0, 2, 3, 1)


def _pow2_scale(amax):
    """The act_scale ``sbc_f16x2_calibrate`` picks for a layer whose staged maximum is ``amax``: amax * s in [2^8, 2^9)."""
    return 2.0 ** (9 - int(np.frexp(np.float32(amax))[1]))


F16X2_SCALE_CASES = [(32, 32, 1, 6, 64, 16, True), (64, 64, 1, 9, 32, 8, True), (64, 64, 1, 30, 16, 4, False),
                     (128, 128, 4, 19, 8, 2, False), (64, 128, 2, 21, 8, 2, True)]


@pytest.mark.parametrize('wino', [False, True])
@pytest.mark.parametrize('log2_scale', [15, 0, -4, -8, -12, -14])
@pytest.mark.parametrize('cin,cout,dil,B,H,W,elu', F16X2_SCALE_CASES)
def test_f16x2_is_fp32_class_at_every_input_scale(gpu, cin, cout, dil, B, H, W, elu, log2_scale, wino):
    """The reference multiplies in IEEE fp32 (test_score.py:25-26), whose relative precision does not depend on the magnitude of
    an activation.  The two-term fp16 split does: with act_scale = 1 (what the packers write) the low term of |x| < 2^-3 is an
    fp16 denormal.  Hence (a) with the layer's act_scale set the way sbc_f16x2_calibrate sets it, the convolution holds the
    tolerance of every other case at input scales 2^15 ... 2^-14 with the range flag clear; (b) with act_scale = 1 it is EITHER
    within tolerance OR the device flag says why not (overflow / underflow bit) -- never silently degraded."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
    if wino and dil != 1:
        pytest.skip('Winograd F(2x2,3x3) applies to undilated 3x3 convolutions')
    rng = np.random.default_rng(cin * 7 + cout + H)
    x = ((rng.standard_normal((B, H, W, cin)) * 1.5 + 0.3) * 2.0 ** log2_scale).astype(F32)
    if elu:
        # positive inputs: ELU is the identity there.  (For tiny NEGATIVE inputs any two fp32 evaluations of exp(x) - 1 differ by
        # ~1e-7 absolute, i.e. by 1e-3 of the value at |x| = 1e-4: that is the ELU's conditioning, not the multiplier's.)
        x = np.abs(x)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(F32)
    dx = _dev(torch, x)
    v = dx
    ref = _f64_conv(torch, v, w, dil)
    amax = float(v.abs().max())
    _lib.range_flag()
    for act_scale in (_pow2_scale(amax), 1.0):
        wx = _dev(torch, pack_conv_weight_f16x2(w, act_scale).view(np.float32))
        out = torch.full((B, H, W, cout), float('nan'), dtype=torch.float32, device='cuda')
        op = _lib.sbc_op(kind=P.CONV, flags=P.CONV_F16X2 | (P.PRO_ELU if elu else 0), B=B, H=H, W=W, cin=cin, cout=cout, ksize=3,
                         dil=dil, in_=_p(dx), out=_p(out), weight_split=_p(wx))
        if wino:
            wwx = _dev(torch, pack_conv_weight_winograd_f16x2(w, act_scale).view(np.float32))
            op.weight_wino_split = _p(wwx)
        _launch(gpu, op)
        flag = _lib.range_flag()
        err = float((out.double() - ref).norm() / ref.norm())
        if act_scale != 1.0 or log2_scale == 0:
            assert flag == 0 and err < TOL, (act_scale, flag, err)
        else:
            assert err < TOL or flag != 0, (flag, err)
            if log2_scale == 15:
                assert flag & _lib.RANGE_OVERFLOW
            if log2_scale <= -12:
                assert flag & _lib.RANGE_UNDERFLOW and err > TOL / 4       # (the loss the guard exists for is real)


@pytest.mark.parametrize('log2_scale', [12, 0, -6, -10, -14])
def test_f16x2_library_calibration_sets_the_layer_scales(gpu, log2_scale):
    """sbc_f16x2_calibrate on a record list (one CONV with both weight forms, one CONV_PAIR whose second convolution sees an
    intermediate 2^-7 below its input): it runs the records on their first sample, and afterwards the trailers on the device hold
    act_scale = 2^(9 - exponent of the staged maximum), descale to match -- and the launches are fp32-class with a clear flag."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
    rng = np.random.default_rng(100 + log2_scale)
    B, H, W, c = 5, 64, 16, 32
    x = np.abs((rng.standard_normal((B, H, W, c)) * 1.5 + 0.3) * 2.0 ** log2_scale).astype(F32)
    w0 = (rng.standard_normal((c, c, 3, 3)) / 17).astype(F32)
    w1 = np.abs(rng.standard_normal((c, c, 3, 3)) / 17 * 2.0 ** -7).astype(F32)   # positive: the intermediate stays where ELU is the identity
    w2 = (rng.standard_normal((c, c, 3, 3)) / 17).astype(F32)
    dx = _dev(torch, x)
    dw0, dw0w = _dev(torch, pack_conv_weight_f16x2(w0).view(F32)), _dev(torch, pack_conv_weight_winograd_f16x2(w0).view(F32))
    dw1, dw2 = _dev(torch, pack_conv_weight_f16x2(w1).view(F32)), _dev(torch, pack_conv_weight_f16x2(w2).view(F32))
    o0 = torch.full((B, H, W, c), float('nan'), dtype=torch.float32, device='cuda')
    o1 = torch.full((B, H, W, c), float('nan'), dtype=torch.float32, device='cuda')
    ops = [_lib.sbc_op(kind=P.CONV, flags=P.CONV_F16X2 | P.PRO_ELU, B=B, H=H, W=W, cin=c, cout=c, ksize=3, dil=1, in_=_p(dx),
                       out=_p(o0), weight_split=_p(dw0), weight_wino_split=_p(dw0w)),
           _lib.sbc_op(kind=P.CONV_PAIR, flags=P.CONV_F16X2, B=B, H=H, W=W, cin=c, cout=c, ksize=3, dil=1, in_=_p(dx),
                       out=_p(o1), weight_split=_p(dw1), weight2_split=_p(dw2))]
    _lib.calibrate_f16x2(ops, torch.cuda.current_stream().cuda_stream)
    assert _lib.range_flag() == 0                                     # the pass clears what it raised itself
    v = dx                                                            # (positive inputs: ELU is the identity)
    t = _f64_conv(torch, v, w1, 1).float()
    te = torch.nn.functional.elu(t)
    for dw, amax in ((dw0, v[0].abs().max()), (dw0w, v[0].abs().max()), (dw1, v[0].abs().max()), (dw2, te[0].abs().max())):
        tr = dw[-4:].cpu().numpy()                                    # (the trailer: act_scale, descale, weight descale, 0)
        assert tr[0] == _pow2_scale(float(amax)) and tr[1] == tr[2] / tr[0] and tr[2] > 0, (tr, float(amax))
    _launch(gpu, ops[0])
    assert _lib.range_flag() == 0
    _launch(gpu, ops[1])
    # the fused kernel evaluates ELU as exp(x) - 1 only: where the calibration found one of its two inputs below 2^-4 it says so
    # (SBC_RANGE_ELU; the host then runs the batch in bf16x3) -- here the inputs are positive, so the numbers are right anyway
    small = min(float(v[0].abs().max()), float(te[0].abs().max())) < 2.0 ** -4
    assert _lib.range_flag() == (_lib.RANGE_ELU if small else 0)
    ref0 = _f64_conv(torch, v, w0, 1)
    ref1 = _f64_conv(torch, te, w2, 1)
    assert float((o0.double() - ref0).norm() / ref0.norm()) < TOL
    assert float(((o1 - dx).double() - ref1).norm() / ref1.norm()) < 2 * TOL   # (two convolutions; the second on a rounded fp32 t)


@pytest.mark.parametrize('log2_scale', [8, 0, -9])
def test_res_block_is_calibrated_and_guarded(gpu, log2_scale):
    """SBC_OP_RES_BLOCK in sbc_f16x2_calibrate: both convolutions get their activation scale (the first one's input is the normalised
    x, O(1) whatever the input scale; the second one's the normalised intermediate), and the launch then matches the unfused records
    at every input scale; an input far outside the calibrated range raises the range word."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2
    rng = np.random.default_rng(300 + log2_scale)
    B, H, W, c = 4, 64, 16, 32
    x = ((rng.standard_normal((B, H, W, c)) * 1.5 + 0.3) * 2.0 ** log2_scale).astype(F32)
    w1, w2 = [(rng.standard_normal((c, c, 3, 3)) / np.sqrt(9 * c)).astype(F32) for _ in range(2)]
    b1, b2 = [(0.1 * rng.standard_normal(c)).astype(F32) for _ in range(2)]
    agb1, agb2 = [[np.ones(c, F32), (1 + 0.1 * rng.standard_normal(c)).astype(F32), (0.1 * rng.standard_normal(c)).astype(F32)] for _ in range(2)]
    s1 = inorm_stats(x, *agb1)
    norm = lambda v, st: (v - st[:, None, None, 0]) * st[:, None, None, 1] + st[:, None, None, 2]
    t = O.conv2d(O.elu(norm(x, s1)).transpose(0, 3, 1, 2), w1, b1, 1).transpose(0, 2, 3, 1)
    main = O.conv2d(O.elu(norm(t, inorm_stats(t, *agb2))).transpose(0, 3, 1, 2), w2, b2, 1).transpose(0, 2, 3, 1)
    d = {k: _dev(torch, a) for k, a in dict(x=x, s1=s1, w1=pack_conv_weight_f16x2(w1).view(np.float32),
                                              w2=pack_conv_weight_f16x2(w2).view(np.float32), b1=b1, b2=b2, n2=np.concatenate(agb2)).items()}
    out = torch.full((B, H, W, c), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.RES_BLOCK, flags=P.CONV_F16X2, B=B, H=H, W=W, cin=c, cout=c, ksize=3, dil=1, in_=_p(d['x']), out=_p(out),
                     stats=_p(d['s1']), weight_split=_p(d['w1']), weight2_split=_p(d['w2']), bias=_p(d['b1']), bias2=_p(d['b2']), norm2=_p(d['n2']))
    _lib.calibrate_f16x2([op], torch.cuda.current_stream().cuda_stream)
    assert _lib.range_flag() == 0
    v1 = np.abs(O.elu(norm(x, s1))[0]).max()
    v2 = np.abs(O.elu(norm(t, inorm_stats(t, *agb2)))[0]).max()
    for dw, amax in ((d['w1'], v1), (d['w2'], v2)):
        tr = dw[-4:].cpu().numpy()
        assert tr[0] in (_pow2_scale(float(amax)), 2 * _pow2_scale(float(amax)), 0.5 * _pow2_scale(float(amax))) and tr[1] == tr[2] / tr[0], (tr, amax)
    _launch(gpu, op)
    assert _lib.range_flag() == 0
    got = out.cpu().numpy()
    # the kernel accumulates the second convolution ONTO x (csrc/conv_res.hip, phase E): where |x| >> |main path| (input scale 2^8
    # against a normalised main path) every matrix instruction rounds at the magnitude of the output -- a few ulp of the OUTPUT, which
    # is what the block is held to there; at the other scales the main path itself is held to the operator tolerance
    assert rel_err(got, x + main) < 1e-6
    if log2_scale <= 0:
        assert rel_err(got - x, main) < TOL
    # far outside the calibrated range: a statistics table that blows the normalised input up by 2^12
    s_big = s1.copy(); s_big[:, 1] *= 4096.0; s_big[:, 2] *= 4096.0
    op.stats = _p(_dev(torch, s_big))
    _launch(gpu, op)
    assert _lib.range_flag() & _lib.RANGE_OVERFLOW


def test_persistent_grid_width_does_not_change_results(gpu):
    """sbc_set_persistent_cus: the persistent kernels (fused pair, CRP stage, ResidualBlock, direct low-resolution convolution) walk
    their tiles with however many workgroups fit the CUs they are given; tiles and samples are independent, so results are
    identical bit for bit at full, half and a sliver of the device."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2
    rng = np.random.default_rng(5)
    L = _lib.lib()

    def run(ops_outs):
        res = []
        for n in (0, 128, 24):
            _lib.check(L.sbc_set_persistent_cus(n))
            try:
                for op, out in ops_outs:
                    out.fill_(float('nan'))
                    _launch(gpu, op)
                res.append([out.clone() for _, out in ops_outs])
            finally:
                _lib.check(L.sbc_set_persistent_cus(0))
        return res
    keep, ops_outs = [], []
    for kind, cc, Hh, Ww, Bb in ((P.CONV_PAIR, 32, 64, 16, 700), (P.CONV_POOL, 32, 64, 16, 300), (P.CONV, 64, 32, 8, 500), (P.CONV, 64, 8, 2, 999),
                                 (P.RES_BLOCK, 32, 64, 16, 300)):
        x = _dev(torch, (rng.standard_normal((Bb, Hh, Ww, cc)) * 1.5 + 0.3).astype(F32))
        w = [_dev(torch, pack_conv_weight_f16x2((rng.standard_normal((cc, cc, 3, 3)) / np.sqrt(9 * cc)).astype(F32)).view(np.float32)) for _ in range(2)]
        out = torch.empty_like(x)
        op = _lib.sbc_op(kind=kind, flags=P.CONV_F16X2 | (P.PRO_ELU if kind in (P.CONV, P.CONV_POOL) else 0), B=Bb, H=Hh, W=Ww, cin=cc, cout=cc,
                         ksize=3, dil=1, in_=_p(x), out=_p(out), weight_split=_p(w[0]))
        if kind in (P.CONV_PAIR, P.RES_BLOCK):
            op.weight2_split = _p(w[1])
        if kind == P.RES_BLOCK:
            extra = [_dev(torch, a) for a in (np.stack([np.zeros((Bb, cc), F32), np.ones((Bb, cc), F32), np.zeros((Bb, cc), F32)], 1),
                                              np.zeros(cc, F32), np.zeros(cc, F32), np.concatenate([np.ones(cc, F32), np.ones(cc, F32), np.zeros(cc, F32)]))]
            op.stats, op.bias, op.bias2, op.norm2 = _p(extra[0]), _p(extra[1]), _p(extra[2]), _p(extra[3])
            keep += extra
        keep += [x, w]
        ops_outs.append((op, out))
    full, half, sliver = run(ops_outs)
    for a, b, c_ in zip(full, half, sliver):
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c_)


def test_f16x2_calibration_input_is_fixed():
    """The calibration pattern is a pure function of the element index (same on every host, CN(0,1)-like)."""
    from score_based_channels_amd import _lib
    a, b = _lib.calibration_input(4096), _lib.calibration_input(8192)
    assert np.array_equal(a, b[:4096]) and abs(float(b.mean())) < 0.03 and abs(float(b.var()) - 0.5) < 0.03


def _chain_reference(x, blocks, ws):
    """RCU / CRP / RES blocks in sequence with the oracle's primitives (layers.py:76-83, 126-134, 443-456); x NHWC.
    ``blocks``: tokens 'R' (RCU), 'C' (CRP), 'S<d>' (ResidualBlock, identity shortcut, dilation d), 'X<d>' (ResidualBlock with a
    shortcut convolution); ``ws``: per block (w1, w2, extra) with extra = dict(b1, b2, n1, n2[, w3, b3]) for the RES blocks."""
    out = x
    for tok, (w1, w2, ex) in zip(blocks, ws):
        if tok == 'R':
            t = O.conv2d(O.elu(out).transpose(0, 3, 1, 2), w1, None, 1)
            out = out + O.conv2d(O.elu(t), w2, None, 1).transpose(0, 2, 3, 1)
        elif tok == 'C':
            out = O.elu(out)
            path = O.conv2d(O.max_pool5(out.transpose(0, 3, 1, 2)), w1, None, 1)
            out = path.transpose(0, 2, 3, 1) + out
            path = O.conv2d(O.max_pool5(path), w2, None, 1)
            out = path.transpose(0, 2, 3, 1) + out
        else:
            d = int(tok[1:])
            v = O.elu(O.instance_norm_plus(out.transpose(0, 3, 1, 2), *ex['n1']))
            t = O.conv2d(v, w1, ex['b1'], d)
            u = O.elu(O.instance_norm_plus(t, *ex['n2']))
            sc = out if tok[0] == 'S' else O.conv2d(out.transpose(0, 3, 1, 2), ex['w3'], ex['b3'], d).transpose(0, 2, 3, 1)
            out = sc + O.conv2d(u, w2, ex['b2'], d).transpose(0, 2, 3, 1)
    return out.astype(F32)


CHAIN_CASES = [('R',), ('C',), ('R', 'R'), ('C', 'R'), ('R', 'R', 'C', 'R'), ('S1',), ('S1', 'R'), ('X1', 'S1'), ('X4', 'S4', 'R', 'R', 'C', 'R'), ('S2',), ('X2', 'C')]


@pytest.mark.parametrize('blocks', CHAIN_CASES, ids=['-'.join(b) for b in CHAIN_CASES])
@pytest.mark.parametrize('B', [1, 8, 13, 203])
@pytest.mark.parametrize('Cc,H,W', [(64, 8, 2), (128, 8, 2), (64, 16, 4), (32, 32, 8), (64, 32, 8)])
def test_chain_matches_oracle(gpu, Cc, H, W, B, blocks):
    """SBC_OP_CHAIN (csrc/conv_chain.hip): runs of RCU blocks, CRP blocks and ResidualBlocks at the 8 x 2 (and 16 x 4) level in one
    launch -- eight (four) samples per workgroup, the running tensor in registers, operands in LDS, column units that skip the taps
    which only read padding, InstanceNorm++ statistics formed in registers, dilated convolutions as their three live taps --
    against the oracle's convolutions / max pools / norms / ELUs block by block, with a different weight scale per convolution,
    ragged last groups, and bit-identical results for a sample whatever batch it is part of."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2
    if W == 4 and any(t[0] in 'SX' and int(t[1:]) > 1 for t in blocks):
        pytest.skip('dilated ResidualBlocks exist at a width of two only (res4 / res5)')
    if W == 8 and any(t[0] in 'SX' and int(t[1:]) > 1 for t in blocks):
        pytest.skip('dilated ResidualBlocks exist at a width of two only (res4 / res5)')
    if W == 8 and B > 100:
        B = 37
    rng = np.random.default_rng(B * 1000 + Cc + len(blocks) + W)
    x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(F32)

    def conv_w(k, second):
        return (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc) * ((0.05 if k % 2 else 0.4) if second else (0.6 if k % 2 else 1.3))).astype(F32)
    ws = []
    for k, tok in enumerate(blocks):
        ex = None
        if tok[0] in 'SX':
            nrm = lambda: tuple((a + 0.1 * rng.standard_normal(Cc)).astype(F32) for a in (1.0, 1.0, 0.0))   # noqa: E731  alpha, gamma, beta
            ex = dict(b1=(0.2 * rng.standard_normal(Cc)).astype(F32), b2=(0.2 * rng.standard_normal(Cc)).astype(F32), n1=nrm(), n2=nrm())
            if tok[0] == 'X':
                ex.update(w3=conv_w(k, False), b3=(0.2 * rng.standard_normal(Cc)).astype(F32))
        ws.append((conv_w(k, False), conv_w(k, True), ex))
    ref = _chain_reference(x, blocks, ws)
    dx = _dev(torch, x)
    keep = []

    def dev(a):
        keep.append(_dev(torch, a))
        return keep[-1].data_ptr()
    ch = _lib.sbc_chain(n_blocks=len(blocks))
    for k, (tok, (w1, w2, ex)) in enumerate(zip(blocks, ws)):
        ch.type[k] = {'R': 0, 'C': 1}.get(tok[0], 2)
        ch.w1[k], ch.w2[k] = dev(pack_conv_weight_f16x2(w1).view(np.float32)), dev(pack_conv_weight_f16x2(w2).view(np.float32))
        if ex is not None:
            ch.dil[k] = int(tok[1:])
            ch.bias1[k], ch.bias2[k] = dev(ex['b1']), dev(ex['b2'])
            ch.norm1[k], ch.norm2[k] = dev(np.concatenate(ex['n1'])), dev(np.concatenate(ex['n2']))
            if 'w3' in ex:
                ch.w3[k], ch.bias3[k] = dev(pack_conv_weight_f16x2(ex['w3']).view(np.float32)), dev(ex['b3'])

    def run(xin):
        out = torch.full(tuple(xin.shape), float('nan'), dtype=torch.float32, device='cuda')
        op = _lib.sbc_op(kind=P.CHAIN, flags=P.CONV_F16X2, B=xin.shape[0], H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(xin), out=_p(out),
                         ext=C.cast(C.pointer(ch), C.c_void_p))
        _launch(gpu, op)
        return out.cpu().numpy()
    got = run(dx)
    assert np.isfinite(got).all()
    if blocks[0] in 'RC':
        base = O.elu(x) if blocks == ('C',) else x                      # what the convolutions' sum is added to
        assert rel_err(got - base, ref - base) < TOL, rel_err(got - base, ref - base)
    assert rel_err(got, ref) < TOL, rel_err(got, ref)
    assert _lib.range_flag() == 0
    if B >= 8:
        assert np.array_equal(run(dx[:5].contiguous()), got[:5])       # batch independence, bit for bit


@pytest.mark.parametrize('Cc,blocks', [(128, ('R', 'C')), (128, ('X2', 'S2')), (64, ('C', 'R')), (64, ('S1',))])
def test_chain_full_and_half_groups_agree(gpu, Cc, blocks):
    """At 8 x 2 the chain kernel exists with eight (four) samples per workgroup and, for batches that would leave most of the chip
    idle, with half of that (csrc/conv_chain.hip: GD = 2; launch_chain picks by batch size) and, at 128 channels, a quarter (GD = 4: two
    samples).  2100 samples take the full groups, pieces of 700 the half groups, pieces of 300 the quarter (128 channels) / half (64)
    groups: the same sums in the same order -- identical bit for bit -- and the oracle's numbers."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2
    B, H, W = 2100, 8, 2
    rng = np.random.default_rng(Cc + len(blocks))
    x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(F32)
    keep, ws = [], []

    def dev(a):
        keep.append(_dev(torch, a))
        return keep[-1].data_ptr()
    ch = _lib.sbc_chain(n_blocks=len(blocks))
    for k, tok in enumerate(blocks):
        w1 = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(F32)
        w2 = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc) * 0.3).astype(F32)
        ex = None
        ch.type[k] = {'R': 0, 'C': 1}.get(tok[0], 2)
        ch.w1[k], ch.w2[k] = dev(pack_conv_weight_f16x2(w1).view(np.float32)), dev(pack_conv_weight_f16x2(w2).view(np.float32))
        if tok[0] in 'SX':
            nrm = lambda: tuple((a + 0.1 * rng.standard_normal(Cc)).astype(F32) for a in (1.0, 1.0, 0.0))   # noqa: E731
            ex = dict(b1=(0.2 * rng.standard_normal(Cc)).astype(F32), b2=(0.2 * rng.standard_normal(Cc)).astype(F32), n1=nrm(), n2=nrm())
            ch.dil[k] = int(tok[1:])
            ch.bias1[k], ch.bias2[k] = dev(ex['b1']), dev(ex['b2'])
            ch.norm1[k], ch.norm2[k] = dev(np.concatenate(ex['n1'])), dev(np.concatenate(ex['n2']))
            if tok[0] == 'X':
                ex.update(w3=(rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(F32), b3=(0.2 * rng.standard_normal(Cc)).astype(F32))
                ch.w3[k], ch.bias3[k] = dev(pack_conv_weight_f16x2(ex['w3']).view(np.float32)), dev(ex['b3'])
        ws.append((w1, w2, ex))
    dx = _dev(torch, x)

    def run(xin):
        out = torch.full(tuple(xin.shape), float('nan'), dtype=torch.float32, device='cuda')
        op = _lib.sbc_op(kind=P.CHAIN, flags=P.CONV_F16X2, B=xin.shape[0], H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=_p(xin), out=_p(out),
                         ext=C.cast(C.pointer(ch), C.c_void_p))
        _launch(gpu, op)
        return out.cpu().numpy()
    full = run(dx)
    assert np.isfinite(full).all() and _lib.range_flag() == 0
    for lo in range(0, B, 300):                                  # (128 channels: quarter groups since round 6; 64: half groups)
        assert np.array_equal(run(dx[lo:lo + 300].contiguous()), full[lo:lo + 300]), lo
    for lo in range(0, B, 700):                                  # (half groups at both channel counts)
        assert np.array_equal(run(dx[lo:lo + 700].contiguous()), full[lo:lo + 700]), lo
    ref = _chain_reference(x[:64], blocks, ws)
    assert rel_err(full[:64], ref) < 3 * TOL


@pytest.mark.parametrize('B', [1, 3, 40])
@pytest.mark.parametrize('cin,cout,H,W', [(32, 64, 64, 16), (32, 64, 16, 16), (64, 64, 32, 8), (64, 64, 16, 8)])
def test_conv_down_matches_oracle(gpu, cin, cout, H, W, B):
    """SBC_OP_CONV_DOWN (csrc/conv_down.hip): the tail of a downsampling ResidualBlock,
    meanpool2(conv3x3(ELU(norm(a))) + b) + meanpool2(conv1x1(x) + bs) (layers.py:443-456, 309-313), as a 4x4 and a 2x2 stride-2
    convolution with the pooled filters -- against the oracle's full-resolution convolutions followed by its mean pool, and bit for
    bit independent of the batch a sample is part of."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_pooled_f16x2
    rng = np.random.default_rng(B * 100 + H + W + cin)
    a = (rng.standard_normal((B, H, W, cin)) * 1.5 + 0.3).astype(F32)
    x = (rng.standard_normal((B, H, W, cin)) * 0.8 - 0.2).astype(F32)
    w3 = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(F32)
    w1 = (rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin) * 0.3).astype(F32)
    b3, b1 = (0.2 * rng.standard_normal(cout)).astype(F32), (0.2 * rng.standard_normal(cout)).astype(F32)
    al, ga, be = ((v + 0.1 * rng.standard_normal(cin)).astype(F32) for v in (1.0, 1.0, 0.0))
    st = inorm_stats(a, al, ga, be)                                              # [B, 3, C]: (mu, scale, shift)
    v = O.elu((a - st[:, None, None, 0]) * st[:, None, None, 1] + st[:, None, None, 2])
    ref = (O.mean_pool2(O.conv2d(v.transpose(0, 3, 1, 2), w3, b3, 1)) + O.mean_pool2(O.conv2d(x.transpose(0, 3, 1, 2), w1, b1, 1))).transpose(0, 2, 3, 1)
    da, dx, dst = _dev(torch, a), _dev(torch, x), _dev(torch, st)
    d3, d1 = _dev(torch, pack_conv_weight_pooled_f16x2(w3).view(np.float32)), _dev(torch, pack_conv_weight_pooled_f16x2(w1).view(np.float32))
    db3, db1 = _dev(torch, b3), _dev(torch, b1)

    def run(n):
        out = torch.full((n, H // 2, W // 2, cout), float('nan'), dtype=torch.float32, device='cuda')
        op = _lib.sbc_op(kind=P.CONV_DOWN, flags=P.CONV_F16X2, B=n, H=H, W=W, cin=cin, cout=cout, ksize=3, dil=1, in_=_p(da), out=_p(out),
                         res1=_p(dx), stats=_p(dst), weight_split=_p(d3), weight2_split=_p(d1), bias=_p(db3), bias2=_p(db1))
        _launch(gpu, op)
        return out.cpu().numpy()
    got = run(B)
    assert np.isfinite(got).all()
    assert rel_err(got, ref) < TOL, rel_err(got, ref)
    assert rel_err_elementwise(got, ref, floor=0.05) < 1e-4
    assert _lib.range_flag() == 0
    if B >= 3:
        assert np.array_equal(run(2), got[:2])
