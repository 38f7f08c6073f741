"""Performance triage helper (not part of the product): run one L1 conv with per-workgroup phase timestamps."""
import ctypes as C, os, sys
import numpy as np
os.environ['SBC_DEBUG_FLAGS'] = '0x20000'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import pack_conv_weight
B, H, W, cin, cout = 1700, 64, 16, 32, 32
x = torch.randn(B, H, W, cin, device='cuda'); res = torch.randn(B, H, W, cout, device='cuda')
w = torch.from_numpy(pack_conv_weight(np.random.randn(cout, cin, 3, 3).astype(np.float32) / 17)).cuda()
out = torch.empty(B, H, W, cout, device='cuda')
op = _lib.sbc_op(kind=P.CONV, flags=P.PRO_ELU, B=B, H=H, W=W, cin=cin, cout=cout, ksize=3, dil=1,
                 in_=x.data_ptr(), out=out.data_ptr(), weight=w.data_ptr(), res1=res.data_ptr())
h = _lib.lib()
for _ in range(3):
    _lib.check(h.sbc_op_launch(C.byref(op), None))
torch.cuda.synchronize()
import time
t_0 = time.perf_counter()
for _ in range(20):
    _lib.check(h.sbc_op_launch(C.byref(op), None))
torch.cuda.synchronize()
print('avg launch us', (time.perf_counter() - t_0) / 20 * 1e6)
nb = 6800
buf = np.zeros(nb * 8, np.int64)
h.sbc_debug_read_prof.argtypes = [C.c_void_p, C.c_int]
assert h.sbc_debug_read_prof(buf.ctypes.data, nb) == 0
t = buf.reshape(nb, 8)
t0 = np.median(t[:, 0])
ph = t[:, :6] - t0
hw = t[:, 7]
xcc = hw >> 32; hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 0x1; se = (hwid >> 13) & 0x7   # gfx9 HW_ID: wave 0-3 simd 4-5 pipe 6-7 cu 8-11 sh 12 se 13-15
key = xcc * 1000 + se * 100 + sh * 16 + cu
print('distinct CUs seen:', len(np.unique(key)), 'kernel span (cycles):', ph[:, 5].max())
d = np.diff(ph, axis=1)
names = ['stage', 'sync1', 'loop', 'sync2', 'epilogue']
print('mean cycles per phase:', {n: int(v) for n, v in zip(names, d.mean(0))}, 'total', int((ph[:, 5] - ph[:, 0]).mean()))
print('p10/p50/p90 loop:', np.percentile(d[:, 2], [10, 50, 90]).astype(int), ' stage:', np.percentile(d[:, 0], [10, 50, 90]).astype(int),
      ' epi:', np.percentile(d[:, 4], [10, 50, 90]).astype(int))
# timeline of one CU
k0 = np.unique(key)[5]
idx = np.where(key == k0)[0]
idx = idx[np.argsort(ph[idx, 0])]
print('CU', k0, 'ran', len(idx), 'workgroups; (start, stage_end, loop_start, loop_end, epi_start, end) in kcycles:')
for i in idx[:14]:
    print('  wg %5d' % i, np.round(ph[i] / 1000, 1))
