import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import pack_conv_weight_f16, pack_conv_weight_f16x2
B, H, W, Cc = 600, 64, 16, 32
rng = np.random.default_rng(1)
x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(np.float32)
w1 = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)
w2 = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)
dx = torch.from_numpy(x).cuda()
st = torch.cuda.current_stream().cuda_stream
for mode, pack, flag in (('f16x2', pack_conv_weight_f16x2, P.CONV_F16X2), ('f16w', pack_conv_weight_f16, P.CONV_F16W)):
    d1 = torch.from_numpy(pack(w1).view(np.float32)).cuda(); d2 = torch.from_numpy(pack(w2).view(np.float32)).cuda()
    out = torch.full((B, H, W, Cc), float('nan'), device='cuda')
    op = _lib.sbc_op(kind=P.CONV_PAIR, flags=flag, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=dx.data_ptr(), out=out.data_ptr(), weight_split=d1.data_ptr(), weight2_split=d2.data_ptr())
    _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st))); torch.cuda.synchronize()
    got = out.cpu().numpy()
    parts = []
    for lo, hi in ((0, 200), (200, 400), (400, 600)):
        part = torch.full((hi - lo, H, W, Cc), float('nan'), device='cuda')
        sub = _lib.sbc_op(kind=P.CONV_PAIR, flags=flag, B=hi - lo, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=dx[lo:hi].data_ptr(), out=part.data_ptr(), weight_split=d1.data_ptr(), weight2_split=d2.data_ptr())
        _lib.check(_lib.lib().sbc_op_launch(C.byref(sub), C.c_void_p(st))); torch.cuda.synchronize()
        parts.append(part.cpu().numpy())
    ref = np.concatenate(parts)
    bad = (got != ref)
    print(mode, 'differing values', int(bad.sum()), 'of', bad.size, 'nan', int(np.isnan(got).sum()))
    if bad.any():
        idx = np.argwhere(bad)
        print(' samples', np.unique(idx[:, 0])[:20], len(np.unique(idx[:, 0])))
        rows, cnt = np.unique(idx[:, 1], return_counts=True); print(' rows', dict(zip(rows.tolist(), cnt.tolist())))
        cols, cnt = np.unique(idx[:, 2], return_counts=True); print(' cols', dict(zip(cols.tolist(), cnt.tolist())))
        ch, cnt = np.unique(idx[:, 3], return_counts=True); print(' ch', dict(zip(ch.tolist(), cnt.tolist())))
        print(' max abs diff', float(np.abs(got - ref)[bad].max()))
