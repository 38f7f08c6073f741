"""Performance triage helper (not part of the product): time every distinct convolution of the score plan at
T trajectories with each multiplier (f32 = direct/Winograd fp32 MFMA, bf16x3 = split-bf16)."""
import collections, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import pack_conv_weight, pack_conv_weight_split, pack_conv_weight_winograd
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
pl = P.build_score_plan(32, 64, 16, 2)
shapes = collections.Counter()
for op in pl.ops:
    if op.kind == P.CONV:
        shapes[(op.src.c, op.dst.c, op.ksize, op.dil, op.src.h, op.src.w, op.flags, op.res1 is not None, op.res2 is not None,
                op.bias is not None, (op.up.h, op.up.w) if op.up is not None else None)] += 1
h = _lib.lib()
tot = {'f32': 0.0, 'bf16x3': 0.0, 'best': 0.0}
print('%-44s %3s %9s %9s' % ('shape (cin,cout,k,dil,H,W,flags)', 'n', 'f32 us', 'bf16x3 us'))
for key, n in sorted(shapes.items(), key=lambda kv: -kv[1]):
    cin, cout, k, dil, H, W, flags, has_r1, has_r2, has_b, up = key
    pool = bool(flags & P.EPI_POOL)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    x = torch.randn(T, H, W, cin, device='cuda')
    out = torch.empty(T, Ho, Wo, cout, device='cuda')
    wn = np.random.randn(cout, cin, k, k).astype(np.float32) / 17
    keep = [torch.from_numpy(pack_conv_weight(wn)).cuda(), torch.randn(T, 3, cin, device='cuda').abs() + 0.5,
            torch.randn(T, Ho, Wo, cout, device='cuda'), torch.randn(T, Ho, Wo, cout, device='cuda'), torch.randn(cout, device='cuda')]
    res = {}
    for mode in ('f32', 'bf16x3'):
        op = _lib.sbc_op(kind=P.CONV, flags=flags, B=T, H=H, W=W, cin=cin, cout=cout, ksize=k, dil=dil,
                         in_=x.data_ptr(), out=out.data_ptr(), weight=keep[0].data_ptr(), stats=keep[1].data_ptr())
        if has_r1: op.res1 = keep[2].data_ptr()
        if has_r2: op.res2 = keep[3].data_ptr()
        if has_b: op.bias = keep[4].data_ptr()
        if up is not None:
            u = torch.randn(T, up[0], up[1], cout, device='cuda'); keep.append(u)
            op.up, op.up_h, op.up_w = u.data_ptr(), up[0], up[1]
        if mode == 'f32' and k == 3 and dil == 1:
            keep.append(torch.from_numpy(pack_conv_weight_winograd(wn)).cuda()); op.weight_wino = keep[-1].data_ptr()
        if mode == 'bf16x3':
            keep.append(torch.from_numpy(pack_conv_weight_split(wn).view(np.float32)).cuda()); op.weight_split = keep[-1].data_ptr()
        for _ in range(2):
            _lib.check(h.sbc_op_launch(C.byref(op), None))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _lib.check(h.sbc_op_launch(C.byref(op), None))
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 10 * 1e3
    for m in ('f32', 'bf16x3'):
        tot[m] += n * res[m]
    tot['best'] += n * min(res.values())
    print('%-44s %3d %9.1f %9.1f' % (str(key[:7]), n, res['f32'], res['bf16x3']))
print('per step (ms): f32 %.3f  bf16x3 %.3f  best-of %.3f' % (tot['f32'] / 1e3, tot['bf16x3'] / 1e3, tot['best'] / 1e3))
