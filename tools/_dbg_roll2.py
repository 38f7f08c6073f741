import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import pack_conv_weight_f16, pack_conv_weight_f16x2
B, H, W, Cc = 600, 64, 16, 32
rng = np.random.default_rng(1)
x = (rng.standard_normal((B, H, W, Cc)) * 1.5 + 0.3).astype(np.float32)
def tapw(dy, dx):
    w = np.zeros((Cc, Cc, 3, 3), np.float32)
    for i in range(Cc): w[i, i, dy + 1, dx + 1] = 1.0
    return w
dx_ = torch.from_numpy(x).cuda()
st = torch.cuda.current_stream().cuda_stream
def run(flag, d1, d2, lo, hi):
    out = torch.full((hi - lo, H, W, Cc), float('nan'), device='cuda')
    op = _lib.sbc_op(kind=P.CONV_PAIR, flags=flag, B=hi - lo, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=dx_[lo:hi].data_ptr(), out=out.data_ptr(), weight_split=d1.data_ptr(), weight2_split=d2.data_ptr())
    _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st))); torch.cuda.synchronize()
    return out.cpu().numpy()
for name, w1, w2 in (('center/center', tapw(0, 0), tapw(0, 0)), ('up/center', tapw(-1, 0), tapw(0, 0)), ('down/center', tapw(1, 0), tapw(0, 0)), ('center/up', tapw(0, 0), tapw(-1, 0)), ('center/down', tapw(0, 0), tapw(1, 0))):
    for mode, pack, flag in (('f16w', pack_conv_weight_f16, P.CONV_F16W),):
        d1 = torch.from_numpy(pack(w1).view(np.float32)).cuda(); d2 = torch.from_numpy(pack(w2).view(np.float32)).cuda()
        got = run(flag, d1, d2, 0, B)
        ref = np.concatenate([run(flag, d1, d2, lo, lo + 200) for lo in (0, 200, 400)])
        bad = got != ref
        idx = np.argwhere(bad)
        rows, cnt = np.unique(idx[:, 1], return_counts=True) if len(idx) else ([], [])
        print(name, mode, 'bad', int(bad.sum()), 'rows', dict(zip(np.asarray(rows).tolist(), np.asarray(cnt).tolist())))
        if len(idx):
            n, r, c, ch = idx[0]
            print('  first', idx[0], 'got', got[n, r, c, ch] - x[n, r, c, ch], 'ref', ref[n, r, c, ch] - x[n, r, c, ch])
