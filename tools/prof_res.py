"""Performance triage helper (not part of the product): time one SBC_OP_RES_BLOCK launch (a whole ResidualBlock at 64x16) against the
convolution -> statistics -> convolution launches it replaces.   usage: prof_res.py [B]
With a -DSBC_RES_TIMELINE build (tools/build_variant.sh res_tl conv_res.hip -DSBC_RES_TIMELINE; SBC_LIB_PATH=tools/var/libsbc_res_tl.so) it also
prints the per-wave timeline of three samples of workgroup 0."""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
ap = argparse.ArgumentParser()
ap.add_argument('B', nargs='?', type=int, default=1700)
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
B, H, W, Cc = a.B, 64, 16, 32
torch.manual_seed(3); np.random.seed(3)
x = torch.randn(B, H, W, Cc, device='cuda')
w = [np.random.randn(Cc, Cc, 3, 3).astype(np.float32) / 17 for _ in range(2)]
dw = [torch.from_numpy(pack_conv_weight_f16x2(v).view(np.float32)).cuda() for v in w]
dww = [torch.from_numpy(pack_conv_weight_winograd_f16x2(v).view(np.float32)).cuda() for v in w]
b1, b2 = torch.randn(Cc, device='cuda') * 0.1, torch.randn(Cc, device='cuda') * 0.1
s1 = torch.stack([torch.zeros(B, Cc), torch.ones(B, Cc), torch.zeros(B, Cc)], 1).cuda().contiguous()
n2 = torch.cat([torch.ones(Cc), torch.ones(Cc), torch.zeros(Cc)]).cuda()
out, t, out2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
pm, pm1, pm2 = [torch.empty(B, 8, Cc, 2, device='cuda') for _ in range(3)]
st2 = torch.empty(B, 3, Cc, device='cuda')
res = _lib.sbc_op(kind=P.RES_BLOCK, flags=P.CONV_F16X2 | P.EPI_MOMENTS_OUT, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=x.data_ptr(),
                  out=out.data_ptr(), stats=s1.data_ptr(), weight_split=dw[0].data_ptr(), weight2_split=dw[1].data_ptr(), bias=b1.data_ptr(),
                  bias2=b2.data_ptr(), norm2=n2.data_ptr(), aux=pm.data_ptr())
fl = P.CONV_F16X2 | P.PRO_NORM | P.PRO_ELU | P.EPI_MOMENTS_OUT
c1 = _lib.sbc_op(kind=P.CONV, flags=fl, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=x.data_ptr(), out=t.data_ptr(), stats=s1.data_ptr(),
                 bias=b1.data_ptr(), weight_split=dw[0].data_ptr(), weight_wino_split=dww[0].data_ptr(), aux=pm1.data_ptr(), tag=1)
st = _lib.sbc_op(kind=P.INORM_STATS, flags=P.PRO_NORM_MOMENTS, B=B, H=H, W=W, cin=Cc, cout=Cc, in_=pm1.data_ptr(), out=st2.data_ptr(), weight=n2.data_ptr())
c2 = _lib.sbc_op(kind=P.CONV, flags=fl, B=B, H=H, W=W, cin=Cc, cout=Cc, ksize=3, dil=1, in_=t.data_ptr(), out=out2.data_ptr(), stats=st2.data_ptr(),
                 bias=b2.data_ptr(), res1=x.data_ptr(), weight_split=dw[1].data_ptr(), weight_wino_split=dww[1].data_ptr(), aux=pm2.data_ptr(), tag=1)
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
tr, t3 = timeit([res]), timeit([c1, st, c2])
if 'tl' in os.path.basename(os.environ.get('SBC_LIB_PATH', '')):
    # SBC_RES_TIMELINE build: block 0's eight waves stamp 12 points of every sample
    nit = (B + 255) // 256 + 1
    dbg = torch.zeros(nit * 8 * 16, dtype=torch.int64, device='cuda')
    res.aux, res.flags = dbg.data_ptr(), P.CONV_F16X2
    run([res], 1); torch.cuda.synchronize()
    v = dbg.view(nit, 8, 16).cpu().numpy()
    names = ['F end', 'top barrier', 'A end', 'A barrier', 'conv1 end', 'moments end', 'stats barrier 1', 'stats barrier 2', 'D end', 'D barrier', 'conv2 end', 'F end']
    its = [i for i in range(1, nit) if v[i, :, 11].min() > 0]
    for it in its[1:4]:
        t0 = v[it, :, 1].min()
        print('sample %d of block 0 (clock ticks since the top barrier opened; min .. max over the 8 waves)' % it)
        for k in range(1, 12):
            print('   %-16s %7d .. %7d' % (names[k], v[it, :, k].min() - t0, v[it, :, k].max() - t0))
        if it == its[2]:
            print('   per wave (hf, sub) = wave & 1, wave >> 1; clock ticks of every stamp (the last four: F sums formed, next x requested, stores issued, F end):')
            for w in range(8):
                print('     wave %d: ' % w + ' '.join('%6d' % (v[it, w, k] - t0) for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14, 11)))
        if it + 1 < nit and v[it + 1, :, 1].min() > 0:
            print('   %-16s %7d' % ('next top barrier', v[it + 1, :, 1].min() - t0))
err = float((out - out2).abs().max() / out2.abs().max())
print('B = %d: fused ResidualBlock %.1f us; conv + statistics + conv %.1f us; max deviation %.2e' % (B, tr, t3, err))
