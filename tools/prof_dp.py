"""Performance triage helper (not part of the product): time one 64 -> 64 3x3 SBC_OP_CONV launch in conv_mode f16x2 -- the direct
persistent kernel (csrc/conv_dp.hip), or with SBC_NO_CONV_DP=1 in the environment the Winograd kernel it replaces.
usage: prof_dp.py [B H W] [--stage elu|elu_res|crp2|plain|norm]   (norm: InstanceNorm++ prologue + ELU + tile-moment output, the NM instantiation)"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2
ap = argparse.ArgumentParser()
ap.add_argument('shape', nargs='*', type=int, default=[1700, 32, 8])
ap.add_argument('--stage', default='elu_res')
ap.add_argument('--iters', type=int, default=30)
ap.add_argument('--channels', type=int, default=64)
a = ap.parse_args()
B, H, W = a.shape
torch.manual_seed(3); np.random.seed(3)
CH = a.channels
x = torch.randn(B, H, W, CH, device='cuda')
r1, r2 = torch.randn_like(x), torch.randn_like(x)
out = torch.empty_like(x)
w1 = np.random.randn(CH, CH, 3, 3).astype(np.float32) / (3 * CH ** 0.5)
d = [torch.from_numpy(f(w1).view(np.float32)).cuda() for f in (pack_conv_weight_f16x2, pack_conv_weight_winograd_f16x2)]
fl = {'elu': P.PRO_ELU, 'elu_res': P.PRO_ELU, 'crp2': P.EPI_RES1_ELU, 'plain': 0, 'norm': P.PRO_NORM | P.PRO_ELU | P.EPI_MOMENTS_OUT}[a.stage]
cv = _lib.sbc_op(kind=P.CONV, flags=P.CONV_F16X2 | fl, B=B, H=H, W=W, cin=CH, cout=CH, ksize=3, dil=1, in_=x.data_ptr(),
                 out=out.data_ptr(), weight_split=d[0].data_ptr(), weight_wino_split=d[1].data_ptr())
if a.stage == 'norm':
    st = torch.stack([torch.zeros(B, CH), torch.ones(B, CH), torch.zeros(B, CH)], 1).cuda().contiguous()
    pm = torch.empty(B, H * W // 128, CH, 2, device='cuda')
    cv.stats, cv.aux = st.data_ptr(), pm.data_ptr()
if a.stage in ('elu_res', 'crp2'):
    cv.res1 = r1.data_ptr()
if a.stage == 'crp2':
    cv.res2 = r2.data_ptr()
h = _lib.lib()
def run(n):
    for _ in range(n):
        _lib.check(h.sbc_op_launch(C.byref(cv), None))
run(3); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(a.iters); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / a.iters * 1e3
which = 'conv_wx3' if os.environ.get('SBC_NO_CONV_DP') else 'conv_dp'
if 'pt' in os.environ.get('SBC_LIB_PATH', '') and which == 'conv_dp':
    dbg = torch.zeros(10, dtype=torch.int64, device='cuda'); cv.aux = dbg.data_ptr()
    run(1); torch.cuda.synchronize()
    v = dbg.tolist()
    names = ['tile wait', 'barrier', 'convert', 'barrier', 'K loop', 'load wait', 'store']
    tot = sum(v[:7]) or 1
    print('wave 0, cycles per phase: ' + ', '.join('%s %.1f%%' % (names[i], 100.0 * v[i] / tot) for i in range(7)), '| per WG %.0f' % (tot / (8 * int(os.environ.get('SBC_DP_WGS', 64)))))
print('%s %s x %d %s: %.1f us' % (which, a.shape, CH, a.stage, t))
