#!/usr/bin/env python3
"""Per-phase cycle sums of conv_pair32_kernel (a -DSBC_PAIR_TIMING build: tools/build_variant.sh p32t conv_pair32.hip -DSBC_PAIR_TIMING;
SBC_LIB_PATH=tools/var/libsbc_p32t.so python tools/prof_pair32_phases.py [B])."""
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
    rng = np.random.default_rng(0)
    H, W, Cc = 64, 16, 32
    x = torch.from_numpy((rng.standard_normal((B, H, W, Cc)) * 1.5).astype(np.float32)).cuda()
    out = torch.empty_like(x)
    w = [torch.from_numpy(pack_conv_weight_f16x2((rng.standard_normal((Cc, Cc, 3, 3)) / np.sqrt(9 * Cc)).astype(np.float32)).view(np.float32)).cuda()
         for _ in range(2)]
    dbg = torch.zeros(16, dtype=torch.int64, device='cuda')
    op = _lib.sbc_op(kind=P.CONV_PAIR, flags=P.CONV_F16X2, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=x.data_ptr(),
                     out=out.data_ptr(), weight_split=w[0].data_ptr(), weight2_split=w[1].data_ptr(), aux=dbg.data_ptr())
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st)))
    torch.cuda.synchronize()
    dbg.zero_()
    reps = 10
    for _ in range(reps):
        _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(st)))
    torch.cuda.synchronize()
    v = dbg.tolist()
    items = v[8]
    names = ['r0 other', 'r0 barrier', 'r0 conv1+epilogue', 'r0 DMA wait', 'r1 other/stores', 'r1 barrier', 'r1 convert+conv2', 'r1 residual wait']
    print('B=%d: %d items in %d launches; cycles per item (wave 0 of each role, mean over workgroups) [%s]'
          % (B, items, reps, os.environ.get('SBC_PAIR32_NO_ILV') and 'no interleave' or 'interleaved'))
    for k, nm in enumerate(names):
        print('  %-20s %8.0f' % (nm, v[k] / max(items, 1)))


if __name__ == '__main__':
    main()
