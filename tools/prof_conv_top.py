#!/usr/bin/env python3
"""Stand-alone timing of one 3x3 32 -> 32 SBC_OP_CONV at 64x16 (conv_mode f16x2) on the GPU box, whichever kernel the dispatcher picks
(csrc/conv_mfma.hip: launch_conv; A/B switches of the library come from the environment, e.g. SBC_NO_CONV_DP32).

    python tools/prof_conv_top.py [B=1700] [reps=50] [H=64] [W=16] [C=32]
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
    from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    rng = np.random.default_rng(0)
    H, W, Cc = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (64, 16, 32)
    x = torch.from_numpy((rng.standard_normal((B, H, W, Cc)) * 1.5).astype(np.float32)).cuda()
    out = torch.empty_like(x)
    w = (rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)
    wd = torch.from_numpy(pack_conv_weight_f16x2(w).view(np.float32)).cuda()
    ww = torch.from_numpy(pack_conv_weight_winograd_f16x2(w).view(np.float32)).cuda()
    bias = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32)).cuda()
    stats = torch.from_numpy(np.concatenate([rng.standard_normal((B, 1, Cc)) * 0.1, 1 + 0.1 * rng.standard_normal((B, 1, Cc)),
                                             0.1 * rng.standard_normal((B, 1, Cc))], axis=1).astype(np.float32)).cuda()
    up = torch.from_numpy(rng.standard_normal((B, H // 2, W // 2, Cc)).astype(np.float32)).cuda()
    pm = torch.zeros(B * (H * W // 128) * Cc * 2, device='cuda')
    cases = {
        'plain (ELU)': dict(flags=P.CONV_F16X2 | P.PRO_ELU),
        'norm + ELU, bias, tile moments (res2.0.conv1)': dict(flags=P.CONV_F16X2 | P.PRO_NORM | P.PRO_ELU | P.EPI_MOMENTS_OUT, stats=stats.data_ptr(),
                                                              bias=bias.data_ptr(), aux=pm.data_ptr(), tag=1),
        'bias, + resized operand (refine5.msf.convs.0)': dict(flags=P.CONV_F16X2 | P.EPI_UP, bias=bias.data_ptr(), up=up.data_ptr(), up_h=H // 2, up_w=W // 2, tag=1),
    }
    st = torch.cuda.current_stream().cuda_stream
    for name, kw in cases.items():
        op = _lib.sbc_op(kind=P.CONV, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=x.data_ptr(), out=out.data_ptr(),
                         weight_split=wd.data_ptr(), weight_wino_split=ww.data_ptr(), **kw)
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
        flops = 2.0 * B * H * W * 9 * Cc * Cc
        env = ' '.join(k for k in ('SBC_NO_CONV_DP32',) if os.environ.get(k)) or 'default'
        print('%-50s %dx%d C=%d B=%d [%s]: %.1f us per launch, %.0f TFLOP/s algorithmic, %.2f TB/s' % (name, H, W, Cc, B, env, us, flops / us / 1e6, 2 * x.numel() * 4 / us / 1e6))


if __name__ == '__main__':
    main()
