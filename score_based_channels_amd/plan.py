"""Launch plan of the NCSNv2Deepest score network (and of one full Langevin step).

``NCSNv2Deepest.forward`` (``ncsnv2/models/ncsnv2.py:269-300``) issues ~700 stock kernels per call; here
the same dataflow is expressed as ~160 fused operator records (``Op``) over NHWC buffers -- one record per
HIP launch of ``libsbc_hip.so`` (``include/sbc_hip.h``).  This module is pure Python (no torch, no HIP): it
only decides *what* is launched in which order and which logical tensors can share storage.  ``scorenet.py``
binds the records to device memory; ``tests/test_plan_cpu.py`` interprets the very same records with the CPU
oracle's primitives to prove the wiring without a GPU.

Fusion rules (reference lines in brackets):
  * InstanceNorm++ -> ELU -> conv  [layers.py:444-449]: a statistics op writes (mu, scale, shift) per
    (sample, channel); the conv applies the affine + ELU while staging its input tile (PRO_NORM|PRO_ELU).
  * ELU -> conv of RCU blocks [layers.py:130-131]: PRO_ELU; ``x += residual`` [:133] is the conv's res1.
  * ResidualBlock ``shortcut + output`` [:456]: res1 of conv2; pooled blocks run the 1x1 shortcut first and
    add it after conv2's own 2x2 mean pool [ConvMeanPool :311-312].
  * CRP [layers.py:76-83]: ``x = act(x)`` is never materialised: maxpool(ELU(x)) == ELU(maxpool(x)), and the
    running sum ``x = path + x`` is folded into the second conv's epilogue as path1 + (path0 + ELU(x)).
  * MSF [layers.py:178-184]: ``sums = 0 + conv0(h0) + resize(conv1(h1))``: conv1 runs first at its own
    resolution, conv0 adds its (bilinear, align_corners) resize in the epilogue.
  * ``h = 2x - 1`` [ncsnv2.py:270-273] lives in the begin conv; normalizer -> ELU -> end_conv -> / sigma
    [ncsnv2.py:291-298] is one kernel.
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional

# op kinds / flags: numerically identical to include/sbc_hip.h
BEGIN_CONV, INORM_STATS, CONV, MAXPOOL5, END_CONV, LANGEVIN, STEP_INC, MEASURE = 1, 2, 3, 4, 5, 6, 7, 8
PRO_ELU, PRO_NORM, PRO_NORM_SELF = 0x001, 0x002, 0x004
PRO_NORM_MOMENTS, EPI_MOMENTS_OUT = 0x4000, 0x8000
EPI_RES1_ELU, EPI_POOL, EPI_UP, EPI_ELUGRAD = 0x010, 0x020, 0x040, 0x080
CONV_F16W = 0x100
CONV_F16X2 = 0x10000
PRO_ELU_ACC = 0x20000         # ELU with fp32's relative accuracy for small negative inputs (include/sbc_hip.h)
# training operators
(DSM_PERTURB, DSM_LOSS, GRAD_ADD, INORM_BWD, MAXPOOL5_BWD, UPSAMPLE_BWD, POOL_BWD, CONV_WGRAD, PACK_WEIGHT, END_CONV_BWD,
 BEGIN_CONV_BWD, ADAM_EMA) = range(9, 21)
CONV_PAIR = 21
CONV_POOL = 22
RES_BLOCK = 23
CONV_DOWN = 25           # the tail of a downsampling ResidualBlock (pooled conv2 + pooled 1x1 shortcut) in one launch (csrc/conv_down.hip)
CHAIN = 24               # a chain of RCU / CRP blocks at the 8 x 2 level in one launch (csrc/conv_chain.hip)
CHAIN_RCU, CHAIN_CRP, CHAIN_RES, CHAIN_MAX_BLOCKS = 0, 1, 2, 6
BWD_ACCUM, PACK_ADJOINT, OP_SIDE, OP_JOIN, PACK_WINOGRAD = 0x200, 0x400, 0x800, 0x1000, 0x2000

# profiling tags (sbc_op.tag; bench.py times each class by hipEvents in a single-stream segment after its timed region):
TAG_CONV_TOP = 1        # 3x3 convs ngf -> ngf at full resolution (also their own kernel symbol in csrc/conv_wx3.hip)
TAG_PAIR_TOP = 2        # fused RCU blocks (CONV_PAIR) at full resolution
TAG_CONV_MID = 3        # undilated 3x3 convs 2 ngf -> 2 ngf at half resolution (the 32x8 level of a 64x16 array)
TAG_POOL_TOP = 4        # fused CRP stages (CONV_POOL) at full resolution
TAG_RES_TOP = 6         # fused ResidualBlocks (RES_BLOCK) at full resolution
TAG_CHAIN = 7           # CHAIN records (csrc/conv_chain.hip): 7 + the index of their kernel instantiation in CHAIN_KERNELS
CHAIN_KERNELS = ((128, 2), (64, 2), (64, 4), (64, 8), (32, 8))       # (channels, width)
TAG_DOWN = 12           # CONV_DOWN records: 12 at 16-pixel rows (res2.0), 13 at 8 (res3.0)
TAG_DIRECT_MID = 5      # the TAG_CONV_MID layers without a norm prologue / resize / tile-moment output: in conv_mode f16x2 the direct
                        # persistent kernel (csrc/conv_dp.hip) takes them, the Winograd kernel the rest


@dataclass
class SelfNorm:
    """What ``ScorePlan.stats`` returns instead of a statistics tensor when the consumer computes them itself (PRO_NORM_SELF)."""
    key: str


@dataclass
class Tensor:
    """A logical per-sample NHWC tensor ``[H][W][C]`` (batch dimension implicit)."""
    name: str
    h: int
    w: int
    c: int
    slot: int = -1            # physical buffer assigned by ``assign_slots``

    @property
    def elems(self):
        return self.h * self.w * self.c


@dataclass
class Op:
    kind: int
    name: str
    src: Optional[Tensor] = None
    dst: Optional[Tensor] = None
    weight: Optional[str] = None        # state_dict key of the conv weight / norm prefix
    weight2: Optional[str] = None       # CONV_PAIR / RES_BLOCK: state_dict key of the second convolution's weight
    bias2: Optional[str] = None         # RES_BLOCK: state_dict key of the second convolution's bias
    norm2: Optional[str] = None         # RES_BLOCK: state_dict prefix of the second norm (its alpha | gamma | beta)
    bias: Optional[str] = None
    stats: Optional[Tensor] = None
    res1: Optional[Tensor] = None
    res2: Optional[Tensor] = None
    up: Optional[Tensor] = None
    flags: int = 0
    ksize: int = 3
    dil: int = 1
    tag: int = 0
    side: bool = False        # may overlap the records that follow it, up to the next ``join`` record (SBC_OP_SIDE)
    join: bool = False        # waits for every side record issued before it (SBC_OP_JOIN)
    moments: Optional[Tensor] = None    # second output: tile moments of dst (EPI_MOMENTS_OUT), [HW/128][C][2] per sample
    geom: Optional[Tensor] = None       # INORM_STATS from tile moments: the tensor whose (H, W, C) the launch describes
    norm_key: Optional[str] = None      # CONV with PRO_NORM_SELF: state_dict prefix of the norm whose (alpha, gamma, beta) `stats` points at
    blocks: Optional[list] = None       # CHAIN: [(CHAIN_RCU | CHAIN_CRP | CHAIN_RES, weight key of conv 1, of conv 2, extra), ...]; extra: None, or for
                                        # a RES block {'bias1', 'bias2', 'norm1', 'norm2', 'dil', 'w3', 'bias3'} (w3 / bias3: shortcut conv or None)
    # launch lanes (sbc_op.lane / signal / wait, include/sbc_hip.h ABI 14; `hoist_skip_branches`)
    lane: int = 0                       # 0: the run stream; 1 .. MAX_LANES-1: a stream of the plan
    signal: int = 0                     # event id recorded behind the record (0: none)
    wait: tuple = ()                    # event ids the record's lane waits for in front of it (at most two)

    def inputs(self):
        return [t for t in (self.src, self.stats, self.res1, self.res2, self.up) if t is not None]

    def outputs(self):
        return [t for t in (self.dst, self.moments) if t is not None]


@dataclass
class ScorePlan:
    ops: List[Op]
    x: Tensor                 # network input  [Nt][Nr][2]   (complex64 view of the current estimate)
    out: Tensor               # network output [Nt][Nr][2]   (complex64 view of the score)
    tensors: List[Tensor]
    slot_elems: List[int] = field(default_factory=list)   # per-sample float32 elements of each physical slot


class _Builder:
    def __init__(self, ngf, nt, nr, overlap=False, fold_stats=False, fuse_pairs=False):
        self.ngf, self.nt, self.nr = ngf, nt, nr
        self.fuse_down = False          # pooled conv2 + pooled shortcut of a downsampling ResidualBlock as one CONV_DOWN record
        self.fuse_chain = False         # RCU / CRP runs of the 8 x 2 level as CHAIN records (csrc/conv_chain.hip)
        self.fuse_res = False           # ResidualBlocks without resampling at 64x16, 32 channels as one RES_BLOCK record (csrc/conv_res.hip)
        self.fuse_pairs = fuse_pairs    # RCU blocks as one CONV_PAIR record (csrc/conv_pair.hip): True / a tuple of (channels, width)
        self.ops, self.tensors = [], []
        self.fold_stats = fold_stats    # full-resolution InstanceNorm++ statistics from tile moments (no statistics launch)
        self.producer = {}              # id(tensor) -> the record that writes it
        self.overlap = overlap      # mark independent low-resolution branches as side records
        self.side_now = False       # records appended while set carry ``side``

    def t(self, name, h, w, c):
        x = Tensor(name, h, w, c)
        self.tensors.append(x)
        return x

    def conv(self, name, src, wkey, cout, *, bias=True, flags=0, stats=None, res1=None, res2=None, up=None,
             ksize=3, dil=1):
        pool = bool(flags & EPI_POOL)
        norm_key = None
        if isinstance(stats, SelfNorm):             # the launch computes the statistics of its input itself (PRO_NORM_SELF)
            norm_key, stats, flags = stats.key, None, flags | PRO_NORM_SELF
        dst = self.t(name, src.h // 2 if pool else src.h, src.w // 2 if pool else src.w, cout)
        tag = TAG_CONV_TOP if (ksize == 3 and src.c == self.ngf and cout == self.ngf and src.h == self.nt) else 0
        if (ksize == 3 and dil == 1 and src.c == 2 * self.ngf and cout == 2 * self.ngf and not flags & EPI_POOL
                and 2 * src.h == self.nt):
            tag = TAG_CONV_MID
        self.ops.append(Op(CONV, name, src=src, dst=dst, weight=wkey + '.weight',
                           bias=(wkey + '.bias') if bias else None, stats=stats, res1=res1, res2=res2, up=up,
                           flags=flags | (EPI_UP if up is not None else 0), ksize=ksize, dil=dil, tag=tag,
                           side=self.side_now, norm_key=norm_key))
        self.producer[id(dst)] = self.ops[-1]
        return dst

    def low_res(self, t):
        """Branches are overlapped only below full resolution: full-resolution launches fill the chip on their own, and the
        dominant kernel's per-launch time (the roofline entry of bench.py) stays the time it runs alone."""
        return self.overlap and t.h < self.nt

    def stats(self, name, src, nkey, consumer_is_conv=True):
        """InstanceNorm++ statistics of ``src`` for the norm ``nkey``.  With ``fold_stats``, for images of at most 64 pixels (the
        16x4 and 8x2 levels of a 64x16 array) there is no statistics record at all: a workgroup of the consuming 3x3 convolution
        holds whole samples and computes them itself (PRO_NORM_SELF; the return value is a ``SelfNorm`` marker that ``conv``
        understands).  With ``fold_stats``, for 32- and 64-channel tensors of whole 128-pixel tiles
        that come out of the begin convolution or an unpooled undilated 3x3 convolution: the producer also
        writes the moments of its 128-pixel tiles (EPI_MOMENTS_OUT) and the statistics record reads THOSE (PRO_NORM_MOMENTS:
        HW / 128 x C x 8 bytes per sample instead of the tensor); consumers see ordinary statistics either way."""
        prod = self.producer.get(id(src))
        hw = src.h * src.w
        if self.fold_stats and consumer_is_conv and hw <= 64 and not hw & (hw - 1) and src.w >= 2 and not src.w & (src.w - 1):
            return SelfNorm(nkey)
        dst = self.t(name, 1, 3, src.c)
        if (self.fold_stats and prod is not None and src.c in (32, 64) and hw % 128 == 0 and hw >= 256
                and 128 % (2 * src.w) == 0 and src.h % max(1, 128 // src.w) == 0
                and ((prod.kind == BEGIN_CONV and src.c == 32)
                     or (prod.kind == CONV and prod.ksize == 3 and prod.dil == 1 and not prod.flags & EPI_POOL)
                     or prod.kind == RES_BLOCK)):
            if prod.moments is None:
                prod.moments = self.t(src.name + '.moments', hw // 128, src.c, 2)     # [tile][channel][(mean, M2)]
                prod.flags |= EPI_MOMENTS_OUT
            self.ops.append(Op(INORM_STATS, name, src=prod.moments, dst=dst, weight=nkey, flags=PRO_NORM_MOMENTS, geom=src))
            return dst
        self.ops.append(Op(INORM_STATS, name, src=src, dst=dst, weight=nkey))
        return dst

    def maxpool(self, name, src, elu):
        dst = self.t(name, src.h, src.w, src.c)
        self.ops.append(Op(MAXPOOL5, name, src=src, dst=dst, flags=PRO_ELU if elu else 0))
        return dst

    # --- blocks -----------------------------------------------------------------------------------
    def residual_block(self, p, x, cout, resample, dilation):
        """layers.py:443-456."""
        d = 1 if dilation is None else dilation
        pooled = resample == 'down' and dilation is None
        c1 = x.c if resample == 'down' else cout
        if self.fuse_chain and chain_fusable(x.h, x.w, x.c, CHAIN_RES) and x.c == cout and not pooled and (d == 1 or x.w == 2):
            # the whole block as a RES block of a CHAIN record: the launch forms both norms' statistics itself (a workgroup holds
            # whole samples), so neither statistics records nor the intermediate tensor exist
            has_sc = resample is not None
            extra = {'bias1': p + 'conv1.bias', 'bias2': p + 'conv2.bias', 'norm1': p + 'normalize1', 'norm2': p + 'normalize2', 'dil': d,
                     'w3': p + 'shortcut.weight' if has_sc else None, 'bias3': p + 'shortcut.bias' if has_sc else None}
            return self.chain(p + 'chain', x, [(CHAIN_RES, p + 'conv1.weight', p + 'conv2.weight', extra)])
        s1 = self.stats(p + 'normalize1', x, p + 'normalize1')
        if self.fuse_res and res_fusable(x.h, x.w, x.c, cout, resample, dilation) and not isinstance(s1, SelfNorm):
            # the whole block in one launch: a workgroup owns a sample and forms normalize2's statistics itself; the intermediate
            # tensor, its statistics record and two tensor round trips do not exist
            out = self.t(p + 'conv2', x.h, x.w, cout)
            self.ops.append(Op(RES_BLOCK, p + 'block', src=x, dst=out, weight=p + 'conv1.weight', weight2=p + 'conv2.weight',
                               bias=p + 'conv1.bias', bias2=p + 'conv2.bias', stats=s1, norm2=p + 'normalize2',
                               side=self.side_now, tag=TAG_RES_TOP if x.h == self.nt else 0))
            self.producer[id(out)] = self.ops[-1]
            return out
        a = self.conv(p + 'conv1', x, p + 'conv1', c1, flags=PRO_NORM | PRO_ELU, stats=s1, dil=d)
        s2 = self.stats(p + 'normalize2', a, p + 'normalize2')
        has_sc = pooled or x.c != cout or resample is not None
        if pooled and self.fuse_down and down_fusable(x.h, x.w, x.c, cout) and not isinstance(s2, SelfNorm):
            # pooled conv2 + pooled 1x1 shortcut as ONE launch of stride-2 convolutions with the pooled filters: both inputs read once,
            # the pooled shortcut tensor never exists
            out = self.t(p + 'conv2', x.h // 2, x.w // 2, cout)
            self.ops.append(Op(CONV_DOWN, p + 'down', src=a, dst=out, weight=p + 'conv2.conv.weight', weight2=p + 'shortcut.conv.weight',
                               bias=p + 'conv2.conv.bias', bias2=p + 'shortcut.conv.bias', stats=s2, res1=x,
                               tag=TAG_DOWN + (0 if x.w == 16 else 1)))
            self.producer[id(out)] = self.ops[-1]
            return out
        if has_sc:
            # the shortcut convolution only meets the main branch again at conv2's residual add
            self.side_now = self.low_res(x)
            if pooled:
                sc = self.conv(p + 'shortcut', x, p + 'shortcut.conv', cout, flags=EPI_POOL, ksize=1)
            else:
                sc = self.conv(p + 'shortcut', x, p + 'shortcut', cout, dil=d)
            joined = self.side_now
            self.side_now = False
        else:
            sc, joined = x, False
        if pooled:
            out = self.conv(p + 'conv2', a, p + 'conv2.conv', cout, flags=PRO_NORM | PRO_ELU | EPI_POOL, stats=s2, res1=sc)
        else:
            out = self.conv(p + 'conv2', a, p + 'conv2', cout, flags=PRO_NORM | PRO_ELU, stats=s2, res1=sc, dil=d)
        self.ops[-1].join = joined
        return out

    def chain(self, name, x, blocks):
        """``blocks`` (RCU / CRP blocks in execution order) on ``x`` as CHAIN records of at most CHAIN_MAX_BLOCKS blocks each."""
        for k in range(0, len(blocks), CHAIN_MAX_BLOCKS):
            part = blocks[k:k + CHAIN_MAX_BLOCKS]
            dst = self.t('%s.%d' % (name, k // CHAIN_MAX_BLOCKS), x.h, x.w, x.c)
            self.ops.append(Op(CHAIN, dst.name, src=x, dst=dst, blocks=part, side=self.side_now,
                               tag=TAG_CHAIN + CHAIN_KERNELS.index((x.c, x.w))))
            self.producer[id(dst)] = self.ops[-1]
            x = dst
        return x

    @staticmethod
    def rcu_blocks(p, n_blocks):
        return [(CHAIN_RCU, p + '%d_1_conv.weight' % i, p + '%d_2_conv.weight' % i, None) for i in range(1, n_blocks + 1)]

    def rcu(self, p, x, n_blocks):
        """layers.py:126-134 (n_stages = 2, no bias)."""
        if self.fuse_chain and chain_fusable(x.h, x.w, x.c):
            return self.chain(p + 'chain', x, self.rcu_blocks(p, n_blocks))
        for i in range(1, n_blocks + 1):
            if self.fuse_pairs and pair_fusable(x.h, x.w, x.c, self.fuse_pairs if isinstance(self.fuse_pairs, tuple) else PAIR_SHAPES):
                # x + conv2(ELU(conv1(ELU(x)))) in one launch, the intermediate tensor never exists in memory
                dst = self.t(p + '%d_2_conv' % i, x.h, x.w, x.c)
                self.ops.append(Op(CONV_PAIR, p + '%d_pair' % i, src=x, dst=dst, weight=p + '%d_1_conv.weight' % i,
                                   weight2=p + '%d_2_conv.weight' % i, side=self.side_now,
                                   tag=TAG_PAIR_TOP if x.h == self.nt else 0))
                self.producer[id(dst)] = self.ops[-1]
                x = dst
                continue
            t = self.conv(p + '%d_1_conv' % i, x, p + '%d_1_conv' % i, x.c, bias=False, flags=PRO_ELU)
            x = self.conv(p + '%d_2_conv' % i, t, p + '%d_2_conv' % i, x.c, bias=False, flags=PRO_ELU, res1=x)
        return x

    def crp(self, p, x):
        """layers.py:76-83 (two stages, max pooling)."""
        if self.fuse_pairs and pool_fusable(x.h, x.w, x.c):
            # each stage -- max pool, convolution, running sum -- as ONE launch (csrc/conv_pair.hip: conv_pool_kernel): the pooled
            # tensors never exist in memory
            path0 = self.t(p + 'convs.0', x.h, x.w, x.c)
            self.ops.append(Op(CONV_POOL, p + 'pool_conv0', src=x, dst=path0, weight=p + 'convs.0.weight', flags=PRO_ELU,
                               side=self.side_now, tag=TAG_POOL_TOP if x.h == self.nt else 0))
            self.producer[id(path0)] = self.ops[-1]
            out = self.t(p + 'convs.1', x.h, x.w, x.c)
            self.ops.append(Op(CONV_POOL, p + 'pool_conv1', src=path0, dst=out, weight=p + 'convs.1.weight', flags=EPI_RES1_ELU,
                               res1=x, res2=path0, side=self.side_now, tag=TAG_POOL_TOP if x.h == self.nt else 0))
            self.producer[id(out)] = self.ops[-1]
            return out
        p0 = self.maxpool(p + 'pool0', x, elu=True)
        path0 = self.conv(p + 'convs.0', p0, p + 'convs.0', x.c, bias=False)
        p1 = self.maxpool(p + 'pool1', path0, elu=False)
        return self.conv(p + 'convs.1', p1, p + 'convs.1', x.c, bias=False, flags=EPI_RES1_ELU, res1=x, res2=path0)

    def refine(self, p, xs, features, end=False):
        """layers.py:234-249; MSF (layers.py:178-184) for two inputs, the second may be at half resolution.  The second
        input's adapt convolutions and its MSF convolution do not depend on the first input's: with ``overlap`` they are
        issued first as side records and the first input's MSF convolution (which adds their result) joins them."""
        if len(xs) == 1 and self.fuse_chain and chain_fusable(xs[0].h, xs[0].w, xs[0].c, CHAIN_CRP) and features == xs[0].c:
            # the whole RefineBlock is one chain: adapt RCU x 2, CRP, output RCU (layers.py:234-249 without the MSF of several inputs)
            return self.chain(p + 'chain', xs[0], self.rcu_blocks(p + 'adapt_convs.0.', 2)
                              + [(CHAIN_CRP, p + 'crp.convs.0.weight', p + 'crp.convs.1.weight', None)]
                              + self.rcu_blocks(p + 'output_convs.', 3 if end else 1))
        if len(xs) == 1:
            h = self.rcu(p + 'adapt_convs.0.', xs[0], 2)
        else:
            self.side_now = self.low_res(xs[0])
            h1 = self.rcu(p + 'adapt_convs.1.', xs[1], 2)
            t1 = self.conv(p + 'msf.convs.1', h1, p + 'msf.convs.1', features)
            joined = self.side_now
            self.side_now = False
            h0 = self.rcu(p + 'adapt_convs.0.', xs[0], 2)
            h = self.conv(p + 'msf.convs.0', h0, p + 'msf.convs.0', features, up=t1)
            self.ops[-1].join = joined
        if self.fuse_chain and chain_fusable(h.h, h.w, h.c, CHAIN_CRP):
            return self.chain(p + 'tail', h, [(CHAIN_CRP, p + 'crp.convs.0.weight', p + 'crp.convs.1.weight', None)]
                              + self.rcu_blocks(p + 'output_convs.', 3 if end else 1))
        h = self.crp(p + 'crp.', h)
        return self.rcu(p + 'output_convs.', h, 3 if end else 1)


# (channels, width) of the RCU blocks the plan fuses into SBC_OP_CONV_PAIR launches (csrc/conv_pair.hip; it also takes 32 channels
# at a width of 8: measured slower than two launches there)
PAIR_SHAPES = ((32, 16),)
# ... in the fp16-weight mode (BASELINE config 5, a 256 x 64 array): also 32-pixel and 64-pixel rows, and the 64-channel levels
PAIR_SHAPES_F16W = ((32, 16), (32, 32), (32, 64), (64, 16), (64, 32))


def down_fusable(h, w, cin, cout):
    """Downsampling ResidualBlocks SBC_OP_CONV_DOWN takes: 32 -> 64 channels at 16-pixel rows, 64 -> 64 at 8 (res2.0 / res3.0 of a
    64 x 16 array), heights that are multiples of 16."""
    if os.environ.get('SBC_NO_CONV_DOWN'):           # A/B aid: pooled Winograd convolution + 1x1 shortcut launch
        return False
    return h % 16 == 0 and ((w == 16 and cin == 32 and cout == 64) or (w == 8 and cin == 64 and cout == 64))


MAX_LANES, MAX_EVENTS = 4, 64

# The skip branches of the decoder: refineK's adapt convolutions of its FIRST input (the encoder output layerK, layers.py:234-249,
# ncsnv2.py:284-289) depend on nothing the decoder computes before refineK's MSF convolution adds them.  For small batches -- what a
# rank holds when the 1700 trajectories of a test_score run are sharded over 4 - 8 GPUs -- the low-resolution launches between the
# encoder output and that MSF convolution are latency-bound and leave most of the chip idle (213 trajectories: 54 workgroups of the
# 8 x 2 chain kernel on 256 CUs); the skip branches then run beside them on lanes of their own.  (branch name prefix, the record the
# branch is issued behind, lane): high-resolution branches start when the main path enters the low-resolution levels, in the order of
# their lanes' list positions; every record is the same launch with the same arguments as in the sequential plan -- results are
# identical bit for bit (tests/test_gpu_parity.py::test_skip_overlap_plan_equals_sequential_plan).
# ONE lane: measured (profiles/r06_skip_lanes.txt) -- two or three lanes are slower than none (every further HIP stream costs more in
# cross-queue dependencies than its concurrency returns), one lane -8 % per step at 213 and 425 trajectories.
DEFAULT_SKIP_SPEC = (('refine5.adapt_convs.0.', 'res3.1.', 1), ('refine4.adapt_convs.0.', 'res3.1.', 1), ('refine3.adapt_convs.0.', 'res3.1.', 1),
                     ('refine31.adapt_convs.0.', 'res31.1.', 1), ('refine2.adapt_convs.0.', 'res4.1.', 1))


def hoist_skip_branches(ops, spec=DEFAULT_SKIP_SPEC):
    """Move every branch of ``spec`` -- the contiguous run of records whose names start with the prefix -- behind the LAST record whose
    name starts with its anchor, onto its lane: the branch's first record waits for an event the anchor record signals, its last
    record signals the event its consumer (the first later record that reads the branch's result) waits for.  In place; branches that
    do not exist in this plan (other fusion settings) are skipped."""
    next_evt = [max([o.signal for o in ops] + [0]) + 1]

    def signal_of(op):
        if not op.signal:
            if next_evt[0] > MAX_EVENTS:
                raise ValueError('more than %d lane events' % MAX_EVENTS)
            op.signal = next_evt[0]
            next_evt[0] += 1
        return op.signal

    for prefix, anchor, lane in spec:
        idx = [i for i, o in enumerate(ops) if o.name.startswith(prefix)]
        if not idx:
            continue
        if idx != list(range(idx[0], idx[-1] + 1)) or not 0 < lane < MAX_LANES:
            raise ValueError('skip branch %r is not one contiguous run of records (or bad lane %d)' % (prefix, lane))
        branch = ops[idx[0]:idx[-1] + 1]
        result = branch[-1].dst
        if any(o.lane or o.side or o.join for o in branch):
            raise ValueError('skip branch %r is already on a lane' % prefix)
        anchors = [i for i, o in enumerate(ops) if o.name.startswith(anchor) and i < idx[0]]
        if not anchors:
            raise ValueError('anchor %r of skip branch %r not found in front of it' % (anchor, prefix))
        a = anchors[-1]
        # the branch reads only tensors that exist at the anchor (its own intermediates aside)
        made = {id(t) for o in branch for t in o.outputs()}
        ready = {id(t) for o in ops[:a + 1] for t in o.outputs()} | {id(ops[0].src)}
        for o in branch:
            for t in o.inputs():
                if id(t) not in made and id(t) not in ready:
                    raise ValueError('skip branch %r reads %s, which record %r does not have yet' % (prefix, t.name, ops[a].name))
        del ops[idx[0]:idx[-1] + 1]
        consumer = next(o for o in ops[idx[0]:] if any(t is result for t in o.inputs()))
        if len(consumer.wait) >= 2:
            raise ValueError('record %r already waits for two events' % consumer.name)
        for o in branch:
            o.lane = lane
        branch[0].wait = tuple(branch[0].wait) + (signal_of(ops[a]),)
        consumer.wait = tuple(consumer.wait) + (signal_of(branch[-1]),)
        ops[a + 1:a + 1] = branch


def merge_chains(ops):
    """Adjacent CHAIN records where the second one is the only consumer of the first one's output become ONE record (up to
    CHAIN_MAX_BLOCKS blocks): res5.0 + res5.1 + the whole of refine1, for instance.  In place."""
    k = 0
    while k + 1 < len(ops):
        a, b = ops[k], ops[k + 1]
        if (a.kind == CHAIN and b.kind == CHAIN and b.src is a.dst and not (a.side or b.side or a.join or b.join)
                and len(a.blocks) + len(b.blocks) <= CHAIN_MAX_BLOCKS
                and not any(a.dst in o.inputs() for o in ops[k + 2:])):
            ops[k:k + 2] = [Op(CHAIN, a.name + '+' + b.name, src=a.src, dst=b.dst, blocks=a.blocks + b.blocks, tag=a.tag)]
        else:
            k += 1


def chain_fusable(h, w, c, kind=CHAIN_RCU):
    """Shapes SBC_OP_CHAIN takes: 8 x 2 samples of 64 or 128 channels, 16 x 4 samples of 64 channels (the two lowest levels of a
    64 x 16 array), and 32 x 8 samples of 32 or 64 channels (a wave holds half a sample there; ResidualBlocks are left to their own
    launches at that size: no gain measured)."""
    if os.environ.get('SBC_NO_CHAIN'):               # A/B aid: every convolution and max pool of those levels as its own launch
        return False
    if h == 16 and w == 4 and c == 64:
        return not os.environ.get('SBC_NO_CHAIN4')   # A/B aid: the 16 x 4 level unfused
    if h == 32 and w == 8 and c in (32, 64):
        # (the kernel also takes ResidualBlocks there -- res2.1 -- but measured no gain over its two Winograd launches with folded
        # statistics: 4.26-4.32 against 4.24-4.25 ms per two-stream step; SBC_CHAIN8_RES=1 plans it, Python plans only)
        if kind == CHAIN_RES:
            return bool(os.environ.get('SBC_CHAIN8_RES')) and not os.environ.get('SBC_NO_CHAIN8')
        if kind == CHAIN_CRP and os.environ.get('SBC_NO_CHAIN8_CRP'):      # A/B aid: refine4's CRP as max pool + convolution launches
            return False
        return not os.environ.get('SBC_NO_CHAIN8')                         # A/B aid: the 32 x 8 level unfused
    return h == 8 and w == 2 and c in (64, 128)


def end_fusable(h, w, c):
    """Shapes whose normalizer statistics the END_CONV launch forms itself (PRO_NORM_SELF): 32 channels, 1024 pixels (the
    normalised sample of a 64 x 16 array fills the LDS of a CU)."""
    if os.environ.get('SBC_NO_END_SELF'):            # A/B aid: statistics record + end convolution as separate launches
        return False
    return c == 32 and h * w == 1024


def res_fusable(h, w, cin, cout, resample, dilation):
    """ResidualBlocks SBC_OP_RES_BLOCK takes: 32 -> 32 channels, no resampling or dilation, 64 x 16 samples (two fp16 operand planes
    of a whole sample fill the LDS of a CU)."""
    if os.environ.get('SBC_NO_RES_BLOCK'):           # A/B aid: the two convolutions and the statistics record as separate launches
        return False
    return cin == 32 and cout == 32 and resample is None and dilation is None and h == 64 and w == 16


def pool_fusable(h, w, c):
    """Shapes SBC_OP_CONV_POOL takes (a CRP stage as one launch): 32 channels, 16-pixel rows, heights that are multiples of 8."""
    if os.environ.get('SBC_NO_CONV_POOL'):           # A/B aid: max pool + convolution as separate launches
        return False
    return c == 32 and w == 16 and h % 8 == 0


def pair_fusable(h, w, c, shapes=PAIR_SHAPES):
    """Shapes SBC_OP_CONV_PAIR takes: (channels, width) in ``shapes``, heights that are multiples of its tile (8 rows at a width
    of 16, 4 rows at 32 / 64)."""
    return (c, w) in shapes and h % (8 if w == 16 else 4) == 0


def build_score_plan(ngf=32, nt=64, nr=16, channels=2, share_slots=True, overlap=False, fold_stats=False, fuse_pairs=False, fuse_res=False,
                     fuse_chain=False, fuse_down=False, fuse_end=False, skip_overlap=None):
    """Op list of one ``NCSNv2Deepest.forward`` for ``[B, 2, nt, nr]`` inputs (ncsnv2.py:269-300).
    ``share_slots=False`` gives every logical tensor its own storage (a training step reads every activation again on
    the way back, ``train.py``)."""
    if nt % 8 or nr % 8:
        raise ValueError('Nt and Nr must be multiples of 8 (three 2x mean-pools), got %dx%d' % (nt, nr))
    b = _Builder(ngf, nt, nr, overlap, fold_stats, fuse_pairs)
    b.fuse_res = bool(fuse_res)
    b.fuse_chain = bool(fuse_chain)
    b.fuse_down = bool(fuse_down)
    x = b.t('x', nt, nr, channels)
    h = b.t('begin_conv', nt, nr, ngf)
    b.ops.append(Op(BEGIN_CONV, 'begin_conv', src=x, dst=h, weight='begin_conv.weight', bias='begin_conv.bias'))
    b.producer[id(h)] = b.ops[-1]
    stages = [('res1', ngf, None, None), ('res2', 2 * ngf, 'down', None), ('res3', 2 * ngf, 'down', None),
              ('res31', 2 * ngf, 'down', None), ('res4', 4 * ngf, 'down', 2), ('res5', 4 * ngf, 'down', 4)]
    layers = []
    for name, cout, resample, dil in stages:
        h = b.residual_block(name + '.0.', h, cout, resample, dil)
        h = b.residual_block(name + '.1.', h, cout, None, dil)
        layers.append(h)
    l1, l2, l3, l31, l4, l5 = layers
    ref1 = b.refine('refine1.', [l5], 4 * ngf)
    ref2 = b.refine('refine2.', [l4, ref1], 2 * ngf)
    ref31 = b.refine('refine31.', [l31, ref2], 2 * ngf)
    ref3 = b.refine('refine3.', [l3, ref31], 2 * ngf)
    ref4 = b.refine('refine4.', [l2, ref3], ngf)
    ref5 = b.refine('refine5.', [l1, ref4], ngf, end=True)
    out = b.t('score', nt, nr, channels)
    if fuse_end and end_fusable(nt, nr, ngf):
        # the normalizer's statistics are formed inside the end-convolution launch (a workgroup holds whole samples: the tensor is
        # read once instead of twice, and there is no statistics record)
        b.ops.append(Op(END_CONV, 'end_conv', src=ref5, dst=out, weight='end_conv.weight', bias='end_conv.bias', flags=PRO_NORM_SELF,
                        norm_key='normalizer'))
    else:
        sn = b.stats('normalizer', ref5, 'normalizer', consumer_is_conv=False)
        b.ops.append(Op(END_CONV, 'end_conv', src=ref5, dst=out, weight='end_conv.weight', bias='end_conv.bias', stats=sn))
    merge_chains(b.ops)
    if skip_overlap:
        if overlap:
            raise ValueError('skip_overlap and overlap (SBC_OP_SIDE records) do not mix')
        hoist_skip_branches(b.ops, DEFAULT_SKIP_SPEC if skip_overlap is True else skip_overlap)
    for op in b.ops:                        # (after the statistics folding, which adds EPI_MOMENTS_OUT to producers)
        if op.tag == TAG_CONV_MID and not op.flags & (PRO_NORM | EPI_UP | EPI_MOMENTS_OUT):
            op.tag = TAG_DIRECT_MID
    plan = ScorePlan(b.ops, x, out, b.tensors)
    if share_slots:
        assign_slots(plan)
    else:
        for i, t in enumerate(plan.tensors):
            t.slot = i
        plan.slot_elems = [t.elems for t in plan.tensors]
    return plan


def assign_slots(plan):
    """Share storage between logical tensors with disjoint lifetimes (linear-scan over the op list).
    The network input and output keep private slots (they are caller-visible)."""
    # a side record may still be running until the next join record: what it reads stays live until then
    done_at, nxt = [0] * len(plan.ops), len(plan.ops) - 1
    for i in range(len(plan.ops) - 1, -1, -1):
        if plan.ops[i].join:
            nxt = i
        done_at[i] = nxt if plan.ops[i].side else i
    # a record on a lane is known to be complete at the first run-stream record that waits for an event which its lane signals at or
    # behind it (records of one lane run in list order); without one, at the end of the list (sbc_plan_run joins every lane there)
    for i, op in enumerate(plan.ops):
        if op.lane:
            done_at[i] = len(plan.ops) - 1
            for k in range(i, len(plan.ops)):
                if plan.ops[k].lane == op.lane and plan.ops[k].signal:
                    waiters = [m for m in range(k + 1, len(plan.ops)) if plan.ops[m].lane == 0 and plan.ops[k].signal in plan.ops[m].wait]
                    if waiters:
                        done_at[i] = waiters[0]
                        break
    last_use = {}
    for i, op in enumerate(plan.ops):
        for t in op.inputs():
            last_use[id(t)] = max(last_use.get(id(t), -1), done_at[i])
    pinned = {id(plan.x), id(plan.out)}
    free = {}                     # elems -> [slot]
    slot_elems = []
    live = []                     # (tensor, last use)

    def alloc(t):
        pool = free.get(t.elems, [])
        if pool and id(t) not in pinned:
            t.slot = pool.pop()
        else:
            t.slot = len(slot_elems)
            slot_elems.append(t.elems)

    alloc(plan.x)
    for i, op in enumerate(plan.ops):
        # outputs may not alias any input of the same op -> allocate before releasing
        for o in op.outputs():
            alloc(o)
            live.append(o)
        for t in list(live):
            if id(t) in pinned:
                continue
            if last_use.get(id(t), -1) <= i and not any(t is o for o in op.outputs()):
                free.setdefault(t.elems, []).append(t.slot)
                live.remove(t)
    plan.slot_elems = slot_elems
    return plan


def chain_conv_count(op):
    """Convolutions of a CHAIN record (two per block, three for a RES block with a shortcut convolution)."""
    return sum(3 if (b[3] and b[3]['w3']) else 2 for b in op.blocks)


def chain_live_tap_fraction(op):
    """Fraction of the 9 W products per output the chain kernel executes (csrc/conv_chain.hip): column units skip the taps that
    only read padding -- dx != 0 for the border columns; a dilated convolution at a width of two keeps its three dx = 0 taps."""
    w = op.src.w
    full = (3 * w + 6 * (w - 1)) / (9.0 * w)
    n = live = 0.0
    for b in op.blocks:
        k = 3 if (b[3] and b[3]['w3']) else 2
        n += k
        live += k * (3.0 / 9.0 if (b[3] and b[3]['dil'] > 1) else full)
    return live / n


def count_conv_flops(plan):
    """2 * MACs of every convolution record, per sample (820 772 864 at 64x16)."""
    total = 0
    for op in plan.ops:
        if op.kind == CONV:
            total += 2 * op.src.h * op.src.w * op.src.c * op.dst.c * op.ksize * op.ksize
        elif op.kind in (CONV_PAIR, RES_BLOCK):
            total += 2 * 2 * op.src.h * op.src.w * op.src.c * op.dst.c * 9
        elif op.kind == CONV_POOL:
            total += 2 * op.src.h * op.src.w * op.src.c * op.dst.c * 9
        elif op.kind == CONV_DOWN:               # (counted as the reference runs it: 3x3 + 1x1 at full resolution)
            total += 2 * op.src.h * op.src.w * op.src.c * op.dst.c * (9 + 1)
        elif op.kind == CHAIN:
            total += chain_conv_count(op) * 2 * op.src.h * op.src.w * op.src.c * op.dst.c * 9
        elif op.kind in (BEGIN_CONV, END_CONV):
            total += 2 * op.src.h * op.src.w * op.src.c * op.dst.c * 9
    return total
