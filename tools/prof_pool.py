"""Performance triage helper (not part of the product): time one SBC_OP_CONV_POOL launch (a CRP stage) against the max-pool +
convolution launches it replaces.   usage: prof_pool.py [B H] [--mode f16x2|f16w] [--stage a|b]"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import (pack_conv_weight_f16, pack_conv_weight_f16x2, pack_conv_weight_winograd_f16,
                                              pack_conv_weight_winograd_f16x2)
ap = argparse.ArgumentParser()
ap.add_argument('shape', nargs='*', type=int, default=[1700, 64])
ap.add_argument('--mode', default='f16x2')
ap.add_argument('--stage', default='b')
ap.add_argument('--iters', type=int, default=30)
a = ap.parse_args()
B, H = a.shape
W = 16
torch.manual_seed(3); np.random.seed(3)
x = torch.randn(B, H, W, 32, device='cuda')
r1, r2 = torch.randn_like(x), torch.randn_like(x)
mid, out, out2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
w1 = np.random.randn(32, 32, 3, 3).astype(np.float32) / 17
pk, pkw, flag = ((pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2, P.CONV_F16X2) if a.mode == 'f16x2' else
                 (pack_conv_weight_f16, pack_conv_weight_winograd_f16, P.CONV_F16W))
d = [torch.from_numpy(f(w1).view(np.float32)).cuda() for f in (pk, pkw)]
fa = P.PRO_ELU if a.stage == 'a' else P.EPI_RES1_ELU
pool = _lib.sbc_op(kind=P.CONV_POOL, flags=flag | fa, B=B, H=H, W=W, cin=32, cout=32, ksize=3, dil=1, in_=x.data_ptr(),
                   out=out.data_ptr(), weight_split=d[0].data_ptr())
mp = _lib.sbc_op(kind=P.MAXPOOL5, flags=P.PRO_ELU if a.stage == 'a' else 0, B=B, H=H, W=W, cin=32, cout=32, in_=x.data_ptr(), out=mid.data_ptr())
cv = _lib.sbc_op(kind=P.CONV, flags=flag | (fa & P.EPI_RES1_ELU), B=B, H=H, W=W, cin=32, cout=32, ksize=3, dil=1, in_=mid.data_ptr(),
                 out=out2.data_ptr(), weight_split=d[0].data_ptr(), weight_wino_split=d[1].data_ptr())
if a.stage == 'b':
    for o in (pool, cv):
        o.res1, o.res2 = r1.data_ptr(), r2.data_ptr()
h = _lib.lib()
def run(ops, n):
    for _ in range(n):
        for o in ops:
            _lib.check(h.sbc_op_launch(C.byref(o), None))
def timeit(ops):
    run(ops, 3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(ops, a.iters); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3
tp, tm, tc = timeit([pool]), timeit([mp]), timeit([cv])
if 'pt' in os.environ.get('SBC_LIB_PATH', ''):
    dbg = torch.zeros(10, dtype=torch.int64, device='cuda'); pool.aux = dbg.data_ptr()
    run([pool], 1); torch.cuda.synchronize()
    v = dbg.tolist()
    names = ['barrier', 'dma issue', 'convert', 'load wait', 'barrier', 'K loop', 'residual wait', 'store']
    for lo, who in ((0, 'conversion wave 0'), (4, 'matrix wave 0')):
        tot = sum(v[lo:lo + 4]) or 1
        print('%s, cycles per phase: ' % who + ', '.join('%s %.1f%%' % (names[lo + i], 100.0 * v[lo + i] / tot) for i in range(4)), '| per WG %.0f' % (tot / 256))
err = float((out - out2).abs().max() / out2.abs().max())
by = 4.0 * B * H * W * 32
print('%s %s stage %s: fused %.1f us, max pool %.1f us + convolution %.1f us = %.1f us; max deviation %.2e'
      % (a.mode, a.shape, a.stage, tp, tm, tc, tm + tc, err))
