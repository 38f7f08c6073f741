"""Performance triage helper (not part of the product): time one SBC_OP_CONV_PAIR launch against the two launches it replaces.
usage: prof_pair.py [B H W] [--mode f16x2|f16w]"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import (pack_conv_weight_f16, pack_conv_weight_f16x2, pack_conv_weight_winograd_f16,
                                              pack_conv_weight_winograd_f16x2)
ap = argparse.ArgumentParser()
ap.add_argument('shape', nargs='*', type=int, default=[1700, 64, 16])
ap.add_argument('--mode', default='f16x2')
ap.add_argument('--iters', type=int, default=30)
a = ap.parse_args()
B, H, W = a.shape
torch.manual_seed(3); np.random.seed(3)
x = torch.randn(B, H, W, 32, device='cuda')
mid, out, out2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
w1, w2 = (np.random.randn(32, 32, 3, 3).astype(np.float32) / 17 for _ in range(2))
pk, pkw, flag = ((pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2, P.CONV_F16X2) if a.mode == 'f16x2' else
                 (pack_conv_weight_f16, pack_conv_weight_winograd_f16, P.CONV_F16W))
d = [torch.from_numpy(f(w).view(np.float32)).cuda() for w in (w1, w2) for f in (pk, pkw)]
pair = _lib.sbc_op(kind=P.CONV_PAIR, flags=flag, B=B, H=H, W=W, cin=32, cout=32, ksize=3, dil=1, in_=x.data_ptr(),
                   out=out.data_ptr(), weight_split=d[0].data_ptr(), weight2_split=d[2].data_ptr())
c1 = _lib.sbc_op(kind=P.CONV, flags=flag | P.PRO_ELU, B=B, H=H, W=W, cin=32, cout=32, ksize=3, dil=1, in_=x.data_ptr(),
                 out=mid.data_ptr(), weight_split=d[0].data_ptr(), weight_wino_split=d[1].data_ptr())
c2 = _lib.sbc_op(kind=P.CONV, flags=flag | P.PRO_ELU, B=B, H=H, W=W, cin=32, cout=32, ksize=3, dil=1, in_=mid.data_ptr(),
                 out=out2.data_ptr(), weight_split=d[2].data_ptr(), weight_wino_split=d[3].data_ptr(), res1=x.data_ptr())
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
tp, tu = timeit([pair]), timeit([c1, c2])
if 'pt' in os.environ.get('SBC_LIB_PATH', ''):
    dbg = torch.zeros(10, dtype=torch.int64, device='cuda'); pair.aux = dbg.data_ptr()
    run([pair], 1); torch.cuda.synchronize()
    names = ['conv2 epilogue -> loop top', 'barrier 1', 'convert', 'barrier 2', 'mid write', 'barrier 3', 'residual issue + conv2', 'wait + store', 'dma issue', 'conv1']
    tot = dbg.sum().item()
    print('wave-0 cycles per phase (sum over %d workgroups): ' % 512 + ', '.join('%s %.1f%%' % (n, 100.0 * v / tot) for n, v in zip(names, dbg.tolist())), '| cycles per WG %.0f' % (tot / 512))
if os.environ.get('DUMP'):
    np.save(os.environ['DUMP'], out.cpu().numpy())
err = float((out - out2).abs().max() / (out2 - x).abs().max())
by = 4.0 * B * H * W * 32
print('%s %s: pair %.1f us (%.2f TB/s of 2 tensors), two launches %.1f us (%.2f TB/s of 5 tensors); max deviation %.2e of the conv part'
      % (a.mode, a.shape, tp, 2 * by / tp / 1e6, tu, 5 * by / tu / 1e6, err))
