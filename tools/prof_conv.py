"""Performance triage helper (not part of the product): run one L1 conv with per-workgroup phase timestamps."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from score_based_channels_amd import _lib, plan as P
from score_based_channels_amd.weights import pack_conv_weight
B, H, W, cin, cout = 1700, 64, 16, 32, 32
x = torch.randn(B, H, W, cin, device='cuda'); res = torch.randn(B, H, W, cout, device='cuda')
w = torch.from_numpy(pack_conv_weight(np.random.randn(cout, cin, 3, 3).astype(np.float32) / 17)).cuda()
out = torch.empty(B, H, W, cout, device='cuda')
FLAGS = int(sys.argv[1], 0) if len(sys.argv) > 1 else P.PRO_ELU
op = _lib.sbc_op(kind=P.CONV, flags=FLAGS, B=B, H=H, W=W, cin=cin, cout=cout, ksize=3, dil=1,
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
