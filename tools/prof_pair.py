#!/usr/bin/env python3
"""Stand-alone timing of SBC_OP_CONV_PAIR / SBC_OP_CONV_POOL at the full-resolution level (64x16, 32 channels) on the GPU box:
hipEvent average over back-to-back launches.  A/B switches of csrc/conv_pair.hip are read from the environment by the library
(SBC_NO_PAIR_ROLL, SBC_NO_PAIR_P3).

    python tools/prof_pair.py [B=1700] [reps=50]
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
    H, W, Cc = 64, 16, 32
    x = torch.from_numpy((rng.standard_normal((B, H, W, Cc)) * 1.5).astype(np.float32)).cuda()
    out = torch.empty_like(x)
    w = [torch.from_numpy(pack_conv_weight_f16x2((rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)).view(np.float32)).cuda()
         for _ in range(2)]
    ops = {
        'pair': _lib.sbc_op(kind=P.CONV_PAIR, flags=P.CONV_F16X2, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=x.data_ptr(),
                            out=out.data_ptr(), weight_split=w[0].data_ptr(), weight2_split=w[1].data_ptr()),
        'pool': _lib.sbc_op(kind=P.CONV_POOL, flags=P.CONV_F16X2 | P.PRO_ELU, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=x.data_ptr(),
                            out=out.data_ptr(), weight_split=w[0].data_ptr()),
    }
    st = torch.cuda.current_stream().cuda_stream
    for name, op in ops.items():
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
        nconv = 2 if name == 'pair' else 1
        flops = nconv * 2.0 * B * H * W * 9 * Cc * Cc
        print('%s 64x16 C=32 B=%d [%s]: %.1f us per launch, %.0f TFLOP/s algorithmic' % (
            name, B, ' '.join(k for k in ('SBC_NO_PAIR_ROLL', 'SBC_NO_PAIR_P3') if os.environ.get(k)) or 'default', us, flops / us / 1e6))


if __name__ == '__main__':
    main()
