"""Performance triage helper (not part of the product): time one convolution launch in isolation.

usage: prof_conv.py [cin cout k dil B H W] [--mode f32|wino|bf16x3] [--flags 0x..] [--no-res]
"""
import argparse, ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import (pack_conv_weight, pack_conv_weight_split, pack_conv_weight_winograd,
                                              pack_conv_weight_winograd_split, pack_conv_weight_f16x2,
                                              pack_conv_weight_winograd_f16x2, pack_conv_weight_f16, pack_conv_weight_winograd_f16)
ap = argparse.ArgumentParser()
ap.add_argument('shape', nargs='*', type=int, default=[32, 32, 3, 1, 1700, 64, 16])
ap.add_argument('--mode', default='bf16x3')
ap.add_argument('--flags', default=str(P.PRO_ELU))
ap.add_argument('--no-res', action='store_true')
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
cin, cout, k, dil, B, H, W = a.shape
torch.manual_seed(1); np.random.seed(1)
x = torch.randn(B, H, W, cin, device='cuda'); res = torch.randn(B, H, W, cout, device='cuda')
wn = np.random.randn(cout, cin, k, k).astype(np.float32) / 17
keep = [torch.from_numpy(pack_conv_weight(wn)).cuda()]
out = torch.empty(B, H, W, cout, device='cuda')
op = _lib.sbc_op(kind=P.CONV, flags=int(a.flags, 0), B=B, H=H, W=W, cin=cin, cout=cout, ksize=k, dil=dil,
                 in_=x.data_ptr(), out=out.data_ptr(), weight=keep[0].data_ptr())
if not a.no_res:
    op.res1 = res.data_ptr()
if int(a.flags, 0) & P.PRO_NORM:
    st = torch.randn(B, 3, cin, device='cuda') * 0.3 + torch.tensor([0.0, 1.0, 0.0], device='cuda').view(1, 3, 1); op.stats = st.data_ptr()
if a.mode == 'wino':
    keep.append(torch.from_numpy(pack_conv_weight_winograd(wn)).cuda()); op.weight_wino = keep[-1].data_ptr()
if a.mode == 'bf16x3':
    keep.append(torch.from_numpy(pack_conv_weight_split(wn).view(np.float32)).cuda()); op.weight_split = keep[-1].data_ptr()
if os.environ.get('SBC_LIB_PATH'):
    _lib.LIB_PATH = os.environ['SBC_LIB_PATH']
if a.mode == 'wx3':
    keep.append(torch.from_numpy(pack_conv_weight_winograd_split(wn).view(np.float32)).cuda()); op.weight_wino_split = keep[-1].data_ptr()
if a.mode in ('f16x2', 'wx2'):
    keep.append(torch.from_numpy(pack_conv_weight_f16x2(wn).view(np.float32)).cuda()); op.weight_split = keep[-1].data_ptr()
    op.flags |= P.CONV_F16X2
    if a.mode == 'wx2':
        keep.append(torch.from_numpy(pack_conv_weight_winograd_f16x2(wn).view(np.float32)).cuda()); op.weight_wino_split = keep[-1].data_ptr()
if a.mode in ('f16w', 'wf16w'):
    keep.append(torch.from_numpy(pack_conv_weight_f16(wn).view(np.float32)).cuda()); op.weight_split = keep[-1].data_ptr()
    op.flags |= P.CONV_F16W
    if a.mode == 'wf16w':
        keep.append(torch.from_numpy(pack_conv_weight_winograd_f16(wn).view(np.float32)).cuda()); op.weight_wino_split = keep[-1].data_ptr()
h = _lib.lib()
for _ in range(3):
    _lib.check(h.sbc_op_launch(C.byref(op), None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    _lib.check(h.sbc_op_launch(C.byref(op), None))
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / a.iters * 1e3
fl = 2.0 * k * k * cin * cout * B * H * W
by = 4.0 * B * H * W * (cin + cout * (1 if a.no_res else 2))
if os.environ.get('WINO_TIMING'):
    dbg = torch.zeros(4096 * 4 * 6, device='cuda'); op.up = dbg.data_ptr()
    _lib.check(h.sbc_op_launch(C.byref(op), None)); torch.cuda.synchronize()
    d = dbg.view(4096, 4, 6).cpu().numpy()
    names = ['stage', 'barrier', 'K loop', 'T write', 'barrier2', 'finish']
    print('wave 0 cycles/WG (mean):', {n: int(d[:, 0, i].mean()) for i, n in enumerate(names)}, 'sum', int(d[:, 0].sum(1).mean()))
if os.environ.get('WP_TIMING'):
    dbg = torch.zeros(256 * 10, device='cuda', dtype=torch.int64); op.up = dbg.data_ptr()
    _lib.check(h.sbc_op_launch(C.byref(op), None)); torch.cuda.synchronize()
    d = dbg.view(256, 10).cpu().numpy().astype(np.float64)
    d = d[d[:, 9] > 0]
    names = ['issue', 'V-form+MFMA', 'barrier1', 'ex write', 'commit', 'barrier2', 'last finish', 'finish']
    nb = d[:, 9].mean()
    print('conv_wp wave 0: blocks per WG %.1f; cycles per block (mean over WGs):' % nb,
          {n: int((d[:, i] / d[:, 9]).mean()) for i, n in enumerate(names)}, 'loop total per block', int((d[:, 8] / d[:, 9]).mean()))
if os.environ.get('DUMP'):
    _lib.check(h.sbc_op_launch(C.byref(op), None)); torch.cuda.synchronize()
    np.save(os.environ['DUMP'], out.cpu().numpy())
if os.environ.get('WX3_TIMING'):
    nwg = (B * H * W + 127) // 128
    dbg = torch.zeros(nwg * 8, device='cuda', dtype=torch.int64); op.up = dbg.data_ptr()
    _lib.check(h.sbc_op_launch(C.byref(op), None)); torch.cuda.synchronize()
    d = dbg.view(nwg, 8).cpu().numpy()
    t = d[:, :7].astype(np.float64) * 0.01            # us (100 MHz)
    t0 = t[:, 0].min()
    names = ['issue loads', 'wait+commit', 'barrier', 'K loop', 'T write+bar', 'finish(last blk)']
    ph = np.diff(t, axis=1)
    print('WGs', nwg, 'span %.1f us' % (t[:, 6].max() - t0), 'mean WG life %.2f us' % (t[:, 6] - t[:, 0]).mean())
    print('  mean us per phase:', {n: round(float(ph[:, i].mean()), 2) for i, n in enumerate(names)})
    hw = d[:, 7]; cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
    ucu = np.unique(cu); print('  distinct CUs', len(ucu), 'WGs per CU mean', nwg / len(ucu))
    # average number of WGs alive per CU over the span
    alive = (t[:, 6] - t[:, 0]).sum() / len(ucu) / (t[:, 6].max() - t0)
    print('  mean concurrent WGs per CU %.2f' % alive)
    order = np.argsort(t[:, 0]); print('  start times (us) of WGs by rank: ', [round(float(t[order[i], 0] - t0), 1) for i in (0, nwg // 4, nwg // 2, 3 * nwg // 4, nwg - 1)])
print('%s %s tile=%s: %.1f us  %.1f TF(direct-equivalent)  %.2f TB/s(algorithmic)' % (a.mode, a.shape, os.environ.get('SBC_TILE', 'auto'), us, fl / us / 1e6, by / us / 1e6))
