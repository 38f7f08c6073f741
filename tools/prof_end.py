#!/usr/bin/env python3
"""Stand-alone timing of the network's tail on the GPU box: SBC_OP_INORM_STATS + SBC_OP_END_CONV against SBC_OP_END_CONV with
SBC_PRO_NORM_SELF (csrc/ops.hip), 64x16 samples of 32 channels.   python tools/prof_end.py [B=1700] [reps=50]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from score_based_channels_amd import _lib, plan as P
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rng = np.random.default_rng(0)
    H, W = 64, 16
    x = torch.from_numpy(rng.standard_normal((B, H, W, 32)).astype(np.float32)).cuda()
    agb = torch.from_numpy(np.stack((np.ones(32), np.ones(32), np.zeros(32))).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.standard_normal((2, 32, 3, 3)) / 17).astype(np.float32)).cuda()
    b = torch.zeros(2, device='cuda')
    sig = torch.ones(8, device='cuda')
    lab = torch.zeros(B, dtype=torch.long, device='cuda')
    stats = torch.zeros(B, 3, 32, device='cuda')
    out = torch.zeros(B, H, W, 2, device='cuda')
    ext = _lib.sbc_endconv(sigmas=sig.data_ptr(), labels=lab.data_ptr())
    pe = C.cast(C.pointer(ext), C.c_void_p)
    st_op = _lib.sbc_op(kind=P.INORM_STATS, B=B, H=H, W=W, cin=32, cout=32, in_=x.data_ptr(), out=stats.data_ptr(), weight=agb.data_ptr())
    end_op = _lib.sbc_op(kind=P.END_CONV, B=B, H=H, W=W, cin=32, cout=2, ksize=3, dil=1, in_=x.data_ptr(), out=out.data_ptr(), weight=w.data_ptr(),
                         bias=b.data_ptr(), stats=stats.data_ptr(), ext=pe)
    self_op = _lib.sbc_op(kind=P.END_CONV, flags=P.PRO_NORM_SELF, B=B, H=H, W=W, cin=32, cout=2, ksize=3, dil=1, in_=x.data_ptr(), out=out.data_ptr(),
                          weight=w.data_ptr(), bias=b.data_ptr(), stats=agb.data_ptr(), ext=pe)
    st = torch.cuda.current_stream().cuda_stream
    for name, ops in (('statistics + end convolution', (st_op, end_op)), ('statistics only', (st_op,)), ('end convolution only', (end_op,)),
                      ('end convolution with its own statistics', (self_op,))):
        for _ in range(5):
            for op in ops:
                _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st)))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for op in ops:
                _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st)))
        e1.record()
        torch.cuda.synchronize()
        print('%-42s B=%d: %.1f us' % (name, B, e0.elapsed_time(e1) / reps * 1e3))


if __name__ == '__main__':
    main()
