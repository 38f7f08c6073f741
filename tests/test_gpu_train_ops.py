"""GPU parity of the training operators (reverse-mode counterparts of the score-network operators, the
DSM loss and the Adam + EMA step), each launched through the C ABI (``sbc_op_launch``) and compared with PyTorch autograd
of the same operator evaluated in float64 on the CPU.  Run on the MI355X box: ``python -m pytest tests -m gpu``."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

F32 = np.float32
TOL = 2e-5


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
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _nchw(torch, a):
    """NHWC numpy -> NCHW float64 torch leaf."""
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2))).double().requires_grad_(True)


def _nhwc(t):
    return t.detach().numpy().transpose(0, 2, 3, 1)


def _inorm_plus(torch, x, alpha, gamma, beta):
    """InstanceNorm2dPlus.forward (normalization.py:163-176) restated with torch primitives, float64."""
    means = x.mean(dim=(2, 3))
    m = means.mean(dim=-1, keepdim=True)
    v = means.var(dim=-1, keepdim=True)
    means_n = (means - m) / torch.sqrt(v + 1e-5)
    h = torch.nn.functional.instance_norm(x, eps=1e-5)
    h = h + means_n[..., None, None] * alpha[None, :, None, None]
    return gamma[None, :, None, None] * h + beta[None, :, None, None]


def _forward_stats(gpu, x, agb):
    """The forward statistics op on the device: [B][3][C]."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    B, H, W, Cc = x.shape
    st = torch.empty(B, 3, Cc, dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.INORM_STATS, B=B, H=H, W=W, cin=Cc, cout=Cc, in_=_p(x), out=_p(st), weight=_p(agb))
    _launch(gpu, op)
    return st


@pytest.mark.parametrize('elu,accum', [(False, False), (True, False), (True, True)])
def test_grad_add(gpu, elu, accum):
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 8, 4, 32)).astype(F32) * 2
    g = rng.standard_normal(x.shape).astype(F32)
    o0 = rng.standard_normal(x.shape).astype(F32)
    ref = g * (np.where(x > 0, 1.0, np.exp(x.astype(np.float64))) if elu else 1.0) + (o0 if accum else 0.0)
    dx, dg, do = _dev(torch, x), _dev(torch, g), _dev(torch, o0)
    op = _lib.sbc_op(kind=P.GRAD_ADD, flags=(P.PRO_ELU if elu else 0) | (P.BWD_ACCUM if accum else 0), B=3, H=8, W=4, cin=32,
                     in_=_p(dx), grad=_p(dg), out=_p(do))
    _launch(gpu, op)
    assert rel_err(do.cpu().numpy(), ref) < 1e-6


@pytest.mark.parametrize('C_,B,H,W,elu,accum', [(32, 3, 64, 16, True, False), (64, 5, 32, 8, True, True),
                                                 (64, 4, 16, 4, True, False), (128, 7, 8, 2, True, True),
                                                 (32, 2, 64, 16, False, False), (32, 1, 256, 64, True, False),
                                                 (128, 2, 32, 8, True, True)])
def test_inorm_backward_matches_autograd(gpu, C_, B, H, W, elu, accum):
    """d/dx and d/d(alpha, gamma, beta) of ELU(InstanceNorm2dPlus(x)) (normalization.py:163-176, layers.py:444-449)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(C_ + B)
    x = (rng.standard_normal((B, H, W, C_)) * 1.3 + 0.2 * rng.standard_normal((B, 1, 1, C_))).astype(F32)
    agb = np.stack([1 + 0.1 * rng.standard_normal(C_), 1 + 0.1 * rng.standard_normal(C_), 0.1 * rng.standard_normal(C_)]).astype(F32)
    g = rng.standard_normal(x.shape).astype(F32)
    o0 = rng.standard_normal(x.shape).astype(F32)
    xt = _nchw(torch, x)
    pt = [torch.from_numpy(agb[i]).double().requires_grad_(True) for i in range(3)]
    y = _inorm_plus(torch, xt, *pt)
    if elu:
        y = torch.nn.functional.elu(y)
    y.backward(torch.from_numpy(g.transpose(0, 3, 1, 2).copy()).double())
    ref_dx = _nhwc(xt.grad) + (o0 if accum else 0.0)
    ref_dp = np.stack([p.grad.numpy() for p in pt])
    dx, dagb, dg, do = _dev(torch, x), _dev(torch, agb), _dev(torch, g), _dev(torch, o0)
    st = _forward_stats(gpu, dx, dagb)
    aux = torch.empty(B * 6 * C_, dtype=torch.float32, device='cuda')
    dp = torch.full((3, C_), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.INORM_BWD, flags=(P.PRO_ELU if elu else 0) | (P.BWD_ACCUM if accum else 0), B=B, H=H, W=W, cin=C_,
                     in_=_p(dx), stats=_p(st), weight=_p(dagb), grad=_p(dg), out=_p(do), aux=_p(aux), wgrad=_p(dp))
    _launch(gpu, op)
    assert rel_err(do.cpu().numpy(), ref_dx) < TOL
    assert rel_err(dp.cpu().numpy(), ref_dp) < TOL


@pytest.mark.parametrize('C_,B,H,W,elu,accum', [(32, 2, 64, 16, True, False), (64, 3, 32, 8, False, True),
                                                 (128, 5, 8, 2, True, True), (64, 2, 16, 4, False, False)])
def test_maxpool5_backward_matches_autograd(gpu, C_, B, H, W, elu, accum):
    """CRPBlock pooling (layers.py:69,77-80): maxpool(ELU(x)) when ``elu``."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(C_ + H)
    x = rng.standard_normal((B, H, W, C_)).astype(F32)
    g = rng.standard_normal(x.shape).astype(F32)
    o0 = rng.standard_normal(x.shape).astype(F32)
    xt = _nchw(torch, x)
    y = torch.nn.functional.max_pool2d(torch.nn.functional.elu(xt) if elu else xt, 5, 1, 2)
    y.backward(torch.from_numpy(g.transpose(0, 3, 1, 2).copy()).double())
    ref = _nhwc(xt.grad) + (o0 if accum else 0.0)
    dx, dg, do = _dev(torch, x), _dev(torch, g), _dev(torch, o0)
    aux = torch.empty(x.size, dtype=torch.uint8, device='cuda')
    op = _lib.sbc_op(kind=P.MAXPOOL5_BWD, flags=(P.PRO_ELU if elu else 0) | (P.BWD_ACCUM if accum else 0), B=B, H=H, W=W,
                     cin=C_, in_=_p(dx), grad=_p(dg), out=_p(do), aux=_p(aux))
    _launch(gpu, op)
    assert rel_err(do.cpu().numpy(), ref) < 1e-6


@pytest.mark.parametrize('C_,B,H,W,uh,uw,accum', [(64, 3, 16, 4, 8, 2, False), (32, 2, 64, 16, 32, 8, True),
                                                   (128, 4, 8, 2, 8, 2, False), (64, 2, 32, 8, 16, 4, True),
                                                   (32, 1, 24, 12, 12, 6, False)])
def test_upsample_backward_matches_autograd(gpu, C_, B, H, W, uh, uw, accum):
    """Adjoint of F.interpolate(mode='bilinear', align_corners=True) (MSFBlock, layers.py:182)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(H + uh)
    g = rng.standard_normal((B, H, W, C_)).astype(F32)
    o0 = rng.standard_normal((B, uh, uw, C_)).astype(F32)
    u = _nchw(torch, o0)
    y = torch.nn.functional.interpolate(u, size=(H, W), mode='bilinear', align_corners=True)
    y.backward(torch.from_numpy(g.transpose(0, 3, 1, 2).copy()).double())
    ref = _nhwc(u.grad) + (o0 if accum else 0.0)
    dg, do = _dev(torch, g), _dev(torch, o0)
    op = _lib.sbc_op(kind=P.UPSAMPLE_BWD, flags=P.BWD_ACCUM if accum else 0, B=B, H=H, W=W, cin=C_, up_h=uh, up_w=uw,
                     grad=_p(dg), out=_p(do))
    _launch(gpu, op)
    assert rel_err(do.cpu().numpy(), ref) < 2e-6


def test_pool_backward(gpu):
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(5)
    g = rng.standard_normal((3, 8, 4, 64)).astype(F32)
    dg = _dev(torch, g)
    out = torch.full((3, 16, 8, 64), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.POOL_BWD, B=3, H=16, W=8, cin=64, grad=_p(dg), out=_p(out))
    _launch(gpu, op)
    ref = np.repeat(np.repeat(g, 2, axis=1), 2, axis=2) * 0.25
    assert np.array_equal(out.cpu().numpy(), ref.astype(F32))


WGRAD_CASES = [
    # cin, cout, k, dil, B, H, W, prologue
    (32, 32, 3, 1, 3, 64, 16, 'norm_elu'),
    (32, 32, 3, 1, 5, 64, 16, 'elu'),
    (32, 64, 3, 1, 2, 64, 16, 'norm_elu'),
    (32, 64, 1, 1, 2, 64, 16, ''),
    (64, 64, 3, 1, 3, 32, 8, 'norm_elu'),
    (64, 64, 1, 1, 3, 32, 8, ''),
    (64, 64, 3, 1, 5, 16, 4, 'elu'),
    (64, 64, 3, 2, 9, 8, 2, 'norm_elu'),
    (64, 128, 3, 2, 9, 8, 2, 'norm_elu'),
    (128, 128, 3, 4, 7, 8, 2, 'norm_elu'),
    (128, 128, 3, 1, 7, 8, 2, ''),
    (128, 64, 3, 1, 6, 8, 2, ''),
    (64, 32, 3, 1, 2, 32, 8, ''),
    (32, 32, 3, 1, 70, 64, 16, 'elu'),          # more tiles than chunks: workgroups walk several tiles
    (32, 32, 3, 1, 1, 256, 64, 'norm_elu'),     # large-array config (Nt256 x Nr64): one image row per tile, halo rows
    (64, 64, 3, 1, 1, 128, 32, 'elu'),
    (128, 128, 3, 4, 1, 32, 8, 'norm_elu'),     # dilation 4 with a real halo (tiles of 8 rows inside a 32-row image)
    (64, 128, 3, 2, 2, 32, 8, 'norm_elu'),
]


@pytest.mark.parametrize('cin,cout,k,dil,B,H,W,pro', WGRAD_CASES)
def test_conv_weight_gradient_matches_autograd(gpu, cin, cout, k, dil, B, H, W, pro):
    """dL/dW, dL/db of nn.Conv2d (layers.py:28-60) behind the forward prologue (InstanceNorm++ affine, ELU)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(hash((cin, cout, k, dil, B)) % (2 ** 31))
    x = (rng.standard_normal((B, H, W, cin)) * 1.2 + 0.1).astype(F32)
    g = rng.standard_normal((B, H, W, cout)).astype(F32)
    agb = np.stack([1 + 0.1 * rng.standard_normal(cin), 1 + 0.1 * rng.standard_normal(cin), 0.1 * rng.standard_normal(cin)]).astype(F32)
    dx, dg, dagb = _dev(torch, x), _dev(torch, g), _dev(torch, agb)
    a = torch.from_numpy(x.transpose(0, 3, 1, 2).copy()).double()
    flags, st = 0, None
    if 'norm' in pro:
        flags |= P.PRO_NORM
        st = _forward_stats(gpu, dx, dagb)
        a = _inorm_plus(torch, a, *[torch.from_numpy(agb[i]).double() for i in range(3)])
    if 'elu' in pro:
        flags |= P.PRO_ELU
        a = torch.nn.functional.elu(a)
    w = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(a, w, b, padding=dil * (k // 2), dilation=dil)
    y.backward(torch.from_numpy(g.transpose(0, 3, 1, 2).copy()).double())
    n_scr = int(_lib.lib().sbc_wgrad_scratch_floats(B, H, W, cin, cout, k))
    assert n_scr > 0
    aux = torch.zeros(n_scr, dtype=torch.float32, device='cuda')
    dw = torch.full((cout, cin, k, k), float('nan'), dtype=torch.float32, device='cuda')
    db = torch.full((cout,), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.CONV_WGRAD, flags=flags, B=B, H=H, W=W, cin=cin, cout=cout, ksize=k, dil=dil, in_=_p(dx),
                     stats=_p(st), grad=_p(dg), aux=_p(aux), wgrad=_p(dw), bgrad=_p(db))
    _launch(gpu, op)
    assert rel_err(dw.cpu().numpy(), w.grad.numpy()) < TOL
    assert rel_err(db.cpu().numpy(), b.grad.numpy()) < TOL
    first = dw.clone()
    dw.fill_(float('nan'))
    _launch(gpu, op)                                   # same scratch again: the arrival counters were left at zero
    assert torch.equal(dw, first)


@pytest.mark.parametrize('cin,cout,k', [(32, 32, 3), (32, 64, 3), (32, 64, 1), (64, 64, 1), (64, 128, 3), (128, 64, 3),
                                        (128, 128, 3)])
def test_device_weight_packing_matches_host_packer(gpu, cin, cout, k):
    """SBC_OP_PACK_WEIGHT writes exactly what sbc_pack_conv_weight_split writes on the host; with SBC_PACK_ADJOINT, what
    the host packer writes for the flipped, transposed weight."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    from score_based_channels_amd.weights import pack_conv_weight_split
    rng = np.random.default_rng(cin + cout + k)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(F32)
    dw = _dev(torch, w)
    from score_based_channels_amd.weights import pack_conv_weight_winograd_split
    for adj in (False, True):
        wa = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)) if adj else w
        for wino in ((False, True) if k == 3 else (False,)):
            host = pack_conv_weight_winograd_split(wa) if wino else pack_conv_weight_split(wa)
            out = torch.zeros(host.size, dtype=torch.int16, device='cuda')
            op = _lib.sbc_op(kind=P.PACK_WEIGHT, flags=(P.PACK_ADJOINT if adj else 0) | (P.PACK_WINOGRAD if wino else 0),
                             cin=cin, cout=cout, ksize=k, in_=_p(dw), out=_p(out))
            _launch(gpu, op)
            assert np.array_equal(out.cpu().numpy().view(np.uint16).ravel(), host.ravel()), (adj, wino)


@pytest.mark.parametrize('mode', ['plain', 'elu', 'elu_plus_other', 'elu_in_place'])
@pytest.mark.parametrize('cin,cout,k,dil,B,H,W', [(32, 64, 1, 1, 2, 64, 16), (32, 64, 3, 1, 2, 64, 16), (64, 128, 3, 2, 5, 8, 2),
                                                   (128, 64, 3, 1, 5, 8, 2), (64, 64, 3, 1, 3, 16, 4), (128, 128, 3, 4, 5, 8, 2),
                                                   (32, 32, 3, 1, 3, 64, 16)])
def test_input_gradient_is_a_conv_with_adjoint_weights(gpu, cin, cout, k, dil, B, H, W, mode):
    """dL/d(conv input) = SBC_OP_CONV(grad, adjoint-packed weight): the route train.py takes for every convolution.  With
    SBC_EPI_ELUGRAD the epilogue also goes back through the forward ELU prologue (times ELU'(x)) and adds the gradient
    collected so far, from another buffer or from the output buffer itself."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(cin * 3 + cout + k)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(F32)
    g = rng.standard_normal((B, H, W, cout)).astype(F32)
    x = (rng.standard_normal((B, H, W, cin)) * 1.5).astype(F32)
    other = rng.standard_normal((B, H, W, cin)).astype(F32)
    xt = _nchw(torch, x)
    a = torch.nn.functional.elu(xt) if mode != 'plain' else xt
    y = torch.nn.functional.conv2d(a, torch.from_numpy(w).double(), None, padding=dil * (k // 2), dilation=dil)
    y.backward(torch.from_numpy(g.transpose(0, 3, 1, 2).copy()).double())
    ref = _nhwc(xt.grad) + (other if mode in ('elu_plus_other', 'elu_in_place') else 0.0)
    dw, dg, dx, dother = _dev(torch, w), _dev(torch, g), _dev(torch, x), _dev(torch, other)
    packed = torch.zeros(w.size * 3, dtype=torch.int16, device='cuda')
    op = _lib.sbc_op(kind=P.PACK_WEIGHT, flags=P.PACK_ADJOINT, cin=cin, cout=cout, ksize=k, in_=_p(dw), out=_p(packed))
    _launch(gpu, op)
    out = dother if mode == 'elu_in_place' else torch.full((B, H, W, cin), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.CONV, B=B, H=H, W=W, cin=cout, cout=cin, ksize=k, dil=dil, in_=_p(dg), out=_p(out),
                     weight_split=_p(packed))
    if mode != 'plain':
        op.flags, op.res2 = P.EPI_ELUGRAD, _p(dx)
    if mode in ('elu_plus_other', 'elu_in_place'):
        op.res1 = _p(dother)
    _launch(gpu, op)
    assert rel_err(out.cpu().numpy(), ref) < TOL
    if k == 3 and dil == 1:
        # the same through the Winograd kernel (adjoint Winograd form packed on the device)
        wpacked = torch.zeros(cout * cin * 16 * 3, dtype=torch.int16, device='cuda')
        _launch(gpu, _lib.sbc_op(kind=P.PACK_WEIGHT, flags=P.PACK_ADJOINT | P.PACK_WINOGRAD, cin=cin, cout=cout, ksize=3,
                                 in_=_p(dw), out=_p(wpacked)))
        if mode == 'elu_in_place':
            dother.copy_(_dev(torch, other))
        else:
            out.fill_(float('nan'))
        op.weight_wino_split = _p(wpacked)
        _launch(gpu, op)
        assert rel_err(out.cpu().numpy(), ref) < TOL


def test_end_conv_backward_matches_autograd(gpu):
    """normalizer -> ELU -> end_conv -> / sigma (ncsnv2.py:291-298): gradient wrt the ELU output, the weight and the bias."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(9)
    B, H, W, Cc = 3, 64, 16, 32
    x = rng.standard_normal((B, H, W, Cc)).astype(F32)
    agb = np.stack([1 + 0.1 * rng.standard_normal(Cc), 1 + 0.1 * rng.standard_normal(Cc), 0.1 * rng.standard_normal(Cc)]).astype(F32)
    w = (rng.standard_normal((2, Cc, 3, 3)) / 17).astype(F32)
    g = rng.standard_normal((B, H, W, 2)).astype(F32)
    sigmas = np.array([3.0, 0.7, 0.05], F32)
    labels = np.array([2, 0, 1], np.int64)
    a = torch.nn.functional.elu(_inorm_plus(torch, torch.from_numpy(x.transpose(0, 3, 1, 2).copy()).double(),
                                            *[torch.from_numpy(agb[i]).double() for i in range(3)])).requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    bt = torch.zeros(2, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(a, wt, bt, padding=1) / torch.from_numpy(sigmas[labels]).double()[:, None, None, None]
    y.backward(torch.from_numpy(g.transpose(0, 3, 1, 2).copy()).double())
    dx, dagb, dwt, dg = _dev(torch, x), _dev(torch, agb), _dev(torch, w), _dev(torch, g)
    dsig, dlab = _dev(torch, sigmas), _dev(torch, labels)
    st = _forward_stats(gpu, dx, dagb)
    ext = _lib.sbc_endconv(sigmas=_p(dsig), labels=_p(dlab))
    aux = torch.zeros(int(_lib.lib().sbc_wgrad_scratch_floats(B, H, W, Cc, 2, 3)), dtype=torch.float32, device='cuda')
    out = torch.full((B, H, W, Cc), float('nan'), dtype=torch.float32, device='cuda')
    gw = torch.full((2, Cc, 3, 3), float('nan'), dtype=torch.float32, device='cuda')
    gb = torch.full((2,), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.END_CONV_BWD, B=B, H=H, W=W, cin=Cc, cout=2, ksize=3, dil=1, in_=_p(dx), stats=_p(st),
                     weight=_p(dwt), grad=_p(dg), out=_p(out), aux=_p(aux), wgrad=_p(gw), bgrad=_p(gb),
                     ext=C.cast(C.pointer(ext), C.c_void_p))
    _launch(gpu, op)
    assert rel_err(out.cpu().numpy(), _nhwc(a.grad)) < TOL
    assert rel_err(gw.cpu().numpy(), wt.grad.numpy()) < TOL
    assert rel_err(gb.cpu().numpy(), bt.grad.numpy()) < TOL


def test_begin_conv_backward_matches_autograd(gpu):
    """h = 2x - 1 -> begin_conv (ncsnv2.py:270-275): weight and bias gradients."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(10)
    B, H, W = 3, 64, 16
    x = rng.standard_normal((B, H, W, 2)).astype(F32)
    g = rng.standard_normal((B, H, W, 32)).astype(F32)
    wt = torch.zeros(32, 2, 3, 3, dtype=torch.float64, requires_grad=True)
    bt = torch.zeros(32, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(2 * torch.from_numpy(x.transpose(0, 3, 1, 2).copy()).double() - 1, wt, bt, padding=1)
    y.backward(torch.from_numpy(g.transpose(0, 3, 1, 2).copy()).double())
    dx, dg = _dev(torch, x), _dev(torch, g)
    aux = torch.zeros(int(_lib.lib().sbc_wgrad_scratch_floats(B, H, W, 2, 32, 3)), dtype=torch.float32, device='cuda')
    gw = torch.full((32, 2, 3, 3), float('nan'), dtype=torch.float32, device='cuda')
    gb = torch.full((32,), float('nan'), dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.BEGIN_CONV_BWD, B=B, H=H, W=W, cin=2, cout=32, ksize=3, dil=1, in_=_p(dx), grad=_p(dg),
                     aux=_p(aux), wgrad=_p(gw), bgrad=_p(gb))
    _launch(gpu, op)
    assert rel_err(gw.cpu().numpy(), wt.grad.numpy()) < TOL
    assert rel_err(gb.cpu().numpy(), bt.grad.numpy()) < TOL


def test_dsm_perturb_and_loss_match_reference_formula(gpu):
    """ncsnv2/losses/dsm.py:14-32 with replayed noise: perturbed samples, per-sample loss and d(mean loss)/d(scores)."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(11)
    B, H, W = 5, 64, 16
    n = H * W * 2
    x = rng.standard_normal((B, n)).astype(F32)
    z = rng.standard_normal((B, n)).astype(F32)
    sigmas = np.exp(np.linspace(np.log(39.15), np.log(0.0004), 50)).astype(F32)
    labels = np.array([0, 49, 17, 30, 5], np.int64)
    dx, dz, dsig, dlab = _dev(torch, x), _dev(torch, z), _dev(torch, sigmas), _dev(torch, labels)
    ext = _lib.sbc_dsm(sigmas=_p(dsig), labels=_p(dlab), noise=_p(dz), anneal_power=2.0)
    pert = torch.empty(B, n, dtype=torch.float32, device='cuda')
    noise = torch.empty(B, n, dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.DSM_PERTURB, B=B, H=H, W=W, cin=2, in_=_p(dx), out=_p(pert), aux=_p(noise),
                     ext=C.cast(C.pointer(ext), C.c_void_p))
    _launch(gpu, op)
    us = sigmas[labels][:, None]
    assert np.array_equal(noise.cpu().numpy(), z * us) and np.array_equal(pert.cpu().numpy(), x + z * us)
    s = torch.from_numpy(rng.standard_normal((B, n)) / us.astype(np.float64)).requires_grad_(True)
    ust = torch.from_numpy(us.astype(np.float64))
    target = -1 / ust ** 2 * torch.from_numpy((z * us).astype(np.float64))
    per = 0.5 * ((s - target) ** 2).sum(dim=-1) * ust.squeeze() ** 2.0
    per.mean(dim=0).backward()
    ds_ = _dev(torch, s.detach().numpy().astype(F32))
    loss = torch.empty(B, dtype=torch.float32, device='cuda')
    dsc = torch.empty(B, n, dtype=torch.float32, device='cuda')
    op = _lib.sbc_op(kind=P.DSM_LOSS, B=B, H=H, W=W, cin=2, in_=_p(ds_), grad=_p(noise), out=_p(loss), aux=_p(dsc),
                     ext=C.cast(C.pointer(ext), C.c_void_p))
    _launch(gpu, op)
    assert np.max(np.abs(loss.cpu().numpy() / per.detach().numpy() - 1)) < 1e-5
    assert rel_err(dsc.cpu().numpy(), s.grad.numpy()) < 1e-5
    # Philox path: standard-normal moments, reproducible, keyed by (seed, sample id, offset)
    dlab0 = _dev(torch, np.zeros(B, np.int64))
    ext2 = _lib.sbc_dsm(sigmas=_p(dsig), labels=_p(dlab0), seed=7, offset=3, anneal_power=2.0)
    outs = []
    for _ in range(2):
        op = _lib.sbc_op(kind=P.DSM_PERTURB, B=B, H=H, W=W, cin=2, in_=_p(dx), out=_p(pert), aux=_p(noise),
                         ext=C.cast(C.pointer(ext2), C.c_void_p))
        _launch(gpu, op)
        outs.append(noise.cpu().numpy() / sigmas[0])
    assert np.array_equal(outs[0], outs[1])
    assert abs(outs[0].mean()) < 0.02 and abs(outs[0].std() - 1) < 0.02 and not np.array_equal(outs[0][0], outs[0][1])


def test_adam_ema_matches_torch_optimizer(gpu):
    """torch.optim.Adam(lr=1e-4, betas=(0.9, 0.999), eps=1e-3) + EMAHelper(mu=0.999).update (train_score.py:43-49,
    models/ema.py:17-22) over four steps from the device step counter."""
    torch, _lib = gpu
    from score_based_channels_amd import plan as P
    rng = np.random.default_rng(12)
    n = 10007
    p0 = rng.standard_normal(n).astype(F32)
    grads = [(rng.standard_normal(n) * 10.0 ** rng.uniform(-4, 1, n)).astype(F32) for _ in range(4)]
    pt = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), amsgrad=False, eps=1e-3)
    shadow = pt.data.clone()
    dp = _dev(torch, p0)
    state = torch.zeros(3, n, dtype=torch.float32, device='cuda')
    state[2] = dp
    step = torch.zeros(1, dtype=torch.int32, device='cuda')
    for k, g in enumerate(grads):
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        shadow = (1. - 0.999) * pt.data + 0.999 * shadow
        dg = _dev(torch, g)
        ext = _lib.sbc_adam(n=n, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-3, ema_mu=0.999, step=_p(step))
        op = _lib.sbc_op(kind=P.ADAM_EMA, in_=_p(dg), out=_p(dp), aux=_p(state), ext=C.cast(C.pointer(ext), C.c_void_p))
        _launch(gpu, op)
        step += 1
        assert np.max(np.abs(dp.cpu().numpy() - pt.data.numpy())) < 5e-7 * (k + 1), k
    assert np.max(np.abs(state[2].cpu().numpy() - shadow.numpy())) < 1e-6
    assert rel_err(state[0].cpu().numpy(), opt.state[pt]['exp_avg'].numpy()) < 1e-6
    assert rel_err(state[1].cpu().numpy(), opt.state[pt]['exp_avg_sq'].numpy()) < 1e-6
