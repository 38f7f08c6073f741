"""Numerics probe (tooling, CPU, imports the oracle): Winograd F(4x4,3x3) with split-bf16 products in place of every\nundilated 3x3 convolution whose image sides are multiples of 4.  Forward error vs the reference goldens: 1.3e-6 (F(2x2):\n0.8e-6, direct fp32: 1.0e-6; F(4x4) with plain fp32 products: 2.7e-6) -- i.e. numerically viable.  See DESIGN.md section 8."""
import sys, numpy as np, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import ncsnv2_oracle as O
from conftest import load_golden, rel_err
from score_based_channels_amd.config import default_config
from score_based_channels_amd.weights import seeded_state_dict, split_bf16x3
F32 = np.float32
BT = np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]], np.float64)
G = np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]], np.float64)
AT = np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]], np.float64)
orig = O.conv2d
MODE = {'split': True}
def conv_f43(x, w, b=None, dilation=1):
    x = np.asarray(x, F32); w = np.asarray(w, F32)
    n, c, h, wd = x.shape; o = w.shape[0]
    if c < 8 or w.shape[2] != 3 or dilation != 1 or h % 4 or wd % 4:
        return orig(x, w, b, dilation)
    U = np.einsum('ij,ocjk,lk->ocil', G, w.astype(np.float64), G).astype(F32)
    xp = np.zeros((n, c, h + 2, wd + 2), F32); xp[:, :, 1:-1, 1:-1] = x
    th, tw = h // 4, wd // 4
    d = np.empty((n, c, th, tw, 6, 6), F32)
    for i in range(6):
        for j in range(6):
            d[..., i, j] = xp[:, :, i:i + h:4, j:j + wd:4][:, :, :th, :tw]
    BTf = BT.astype(F32)
    R = np.einsum('xi,nctsij->nctsxj', BTf, d).astype(F32)
    V = np.einsum('nctsxj,vj->nctsxv', R, BTf).astype(F32)
    if MODE['split']:
        Vs = split_bf16x3(V); Us = split_bf16x3(U)
        M = np.zeros((n, o, th, tw, 6, 6), F32)
        for (i, j) in [(2,0),(0,2),(1,1),(1,0),(0,1),(0,0)]:
            M = (M + np.einsum('nctsxv,ocxv->notsxv', Vs[i], Us[j]).astype(F32)).astype(F32)
    else:
        M = np.einsum('nctsxv,ocxv->notsxv', V, U).astype(F32)
    ATf = AT.astype(F32)
    T = np.einsum('notsxv,bv->notsxb', M, ATf).astype(F32)
    Y = np.einsum('ax,notsxb->notsab', ATf, T).astype(F32)
    out = np.empty((n, o, h, wd), F32)
    for a in range(4):
        for bb in range(4):
            out[:, :, a::4, bb::4] = Y[..., a, bb]
    if b is not None: out = (out + np.asarray(b, F32)[None, :, None, None]).astype(F32)
    return out
cfg = default_config(); sd = seeded_state_dict(cfg, 2024)
g = load_golden('forward_64x16.npz')
# single-layer check
rng = np.random.default_rng(0)
x = rng.standard_normal((2, 32, 64, 16)).astype(F32); w = (rng.standard_normal((32, 32, 3, 3)) / 17).astype(F32)
ref = orig(x.astype(np.float64).astype(F32), w); 
ref64 = None
print('single layer F(4,3) fp32 err vs direct fp32:', rel_err(conv_f43(x, w), ref))
O.conv2d = conv_f43
for sp in (False, True):
    MODE['split'] = sp
    e = [rel_err(O.score_forward(sd, g['x'], np.full((4,), lv)), g['out'][i]) for i, lv in enumerate(g['levels'])]
    print('F(4x4,3x3) split=%s forward err' % sp, e)
