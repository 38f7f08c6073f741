#!/usr/bin/env python3
"""Stand-alone timing of SBC_OP_CHAIN (csrc/conv_chain.hip) on the GPU box: hipEvent average over back-to-back launches.

    python tools/prof_chain.py [B=1700] [reps=50]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from score_based_channels_amd import _lib, plan as P
    from score_based_channels_amd.weights import pack_conv_weight_f16x2
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rng = np.random.default_rng(0)
    for Cc, blocks, H, W in ((128, 'RRCR', 8, 2), (128, 'RR', 8, 2), (64, 'RR', 8, 2), (64, 'CR', 8, 2), (64, 'RR', 16, 4), (64, 'CR', 16, 4), (32, 'RRR', 32, 8), (64, 'RR', 32, 8)):
        x = torch.from_numpy((rng.standard_normal((B, H, W, Cc)) * 1.5).astype(np.float32)).cuda()
        out = torch.empty_like(x)
        ws = [[torch.from_numpy(pack_conv_weight_f16x2((rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)).view(np.float32)).cuda()
               for _ in range(2)] for _ in blocks]
        ch = _lib.sbc_chain(n_blocks=len(blocks))
        for k, b in enumerate(blocks):
            ch.type[k], ch.w1[k], ch.w2[k] = (0 if b == 'R' else 1), ws[k][0].data_ptr(), ws[k][1].data_ptr()
        op = _lib.sbc_op(kind=P.CHAIN, flags=P.CONV_F16X2, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=x.data_ptr(), out=out.data_ptr(),
                         ext=C.cast(C.pointer(ch), C.c_void_p))
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st)))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st)))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        nconv = 2 * len(blocks)
        flops = nconv * 2.0 * B * H * W * 9 * Cc * Cc
        if 'tl' in os.path.basename(os.environ.get('SBC_LIB_PATH', '')):
            # SBC_CHAIN_TIMELINE build: the middle workgroup's waves stamp seven points of every phase (convolution)
            dbg = torch.zeros(8 * 16 * 8, dtype=torch.int64, device='cuda')
            op.aux = dbg.data_ptr()
            _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st)))
            torch.cuda.synchronize()
            op.aux = None
            v = dbg.view(8, 16, 8).cpu().numpy()
            nw = int((v[:, 0, 5] > 0).sum())
            t0 = v[:nw, 0, 0].min()
            print('   middle workgroup, %d waves; clock ticks since the first stamp; per phase and wave: parameters read, operand values formed, split + prologue, planes free, planes written, K loop done, result taken' % nw)
            for ph in range(2 * len(blocks)):
                for wv in (0, nw - 1):
                    print('   phase %d wave %d: ' % (ph, wv) + ' '.join('%7d' % (v[wv, ph, k] - t0) for k in range(7)))
        print('%dx%d C=%d blocks=%s B=%d: %.1f us per launch, %.2f us per convolution, %.0f TFLOP/s algorithmic' % (H, W, Cc, blocks, B, us, us / nconv, flops / us / 1e6))


if __name__ == '__main__':
    main()
