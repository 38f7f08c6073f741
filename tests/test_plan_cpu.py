"""CPU checks of the launch plan (plan.py): wiring, fusion rules and buffer sharing."""
import numpy as np

from conftest import load_golden, rel_err
from plan_interp import run_plan
from score_based_channels_amd import plan as P


def test_plan_counts_and_flops():
    pl = P.build_score_plan(32, 64, 16)
    kinds = [op.kind for op in pl.ops]
    assert kinds.count(P.CONV) == 111 and kinds.count(P.BEGIN_CONV) == 1 and kinds.count(P.END_CONV) == 1
    assert kinds.count(P.INORM_STATS) == 25 and kinds.count(P.MAXPOOL5) == 12
    assert P.count_conv_flops(pl) == 820772864
    assert P.count_conv_flops(P.build_score_plan(32, 256, 64)) == 13132365824
    assert sum(1 for op in pl.ops if op.tag == P.TAG_CONV_TOP) == 18
    # the half-resolution 64 -> 64 layers: those a direct kernel can take (no norm prologue / resize / tile moments) and the rest
    assert sum(1 for op in pl.ops if op.tag == P.TAG_DIRECT_MID) == 4 and sum(1 for op in pl.ops if op.tag == P.TAG_CONV_MID) == 3
    assert all(not op.flags & (P.PRO_NORM | P.EPI_UP | P.EPI_MOMENTS_OUT | P.EPI_POOL) for op in pl.ops if op.tag == P.TAG_DIRECT_MID)
    # the two ResidualBlocks of the full-resolution level as one record each (optional): 4 convolution and 2 statistics records fewer
    fr = P.build_score_plan(32, 64, 16, fuse_res=True)
    kf = [op.kind for op in fr.ops]
    assert kf.count(P.RES_BLOCK) == 2 and kf.count(P.CONV) == 107 and kf.count(P.INORM_STATS) == 23
    assert P.count_conv_flops(fr) == 820772864 and P.build_score_plan(32, 32, 32, fuse_res=True).ops[3].kind != P.RES_BLOCK


def test_plan_slots_never_alias_live_tensors():
    pl = P.build_score_plan(32, 64, 16)
    last = {}
    for i, op in enumerate(pl.ops):
        for t in op.inputs():
            last[id(t)] = i
    owner = {pl.x.slot: pl.x}
    for i, op in enumerate(pl.ops):
        for t in op.inputs():
            assert owner.get(t.slot) is t, (op.name, t.name)
        assert all(op.dst.slot != t.slot for t in op.inputs()), op.name
        prev = owner.get(op.dst.slot)
        assert prev is None or last.get(id(prev), -1) < i, (op.name, prev.name if prev else None)
        owner[op.dst.slot] = op.dst
    # far fewer physical buffers than logical tensors
    assert len(pl.slot_elems) < len(pl.tensors) // 3


def test_plan_semantics_match_reference_forward(weights64):
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    pl = P.build_score_plan(32, 64, 16)
    x = np.ascontiguousarray(g['x'][:2].transpose(0, 2, 3, 1))
    out = run_plan(pl, sd, x, np.full((2,), 1155))
    assert rel_err(out.transpose(0, 3, 1, 2), g['out'][1][:2]) < 2e-5


def test_folded_statistics_plan_matches_reference_forward(weights64):
    """``build_score_plan(fold_stats=True)``: the producers of 32- and 64-channel tensors of whole 128-pixel tiles (begin convolution, unpooled 3x3 convolutions) carry
    a tile-moments output and the statistics records of their tensors read those moments (PRO_NORM_MOMENTS) instead of the
    tensors.  Below 65 pixels per image (the 16x4 and 8x2 levels) there is no statistics record at all: the fourteen 3x3
    convolutions that consume those norms compute the statistics of their own input (PRO_NORM_SELF, `stats` -> the norm's
    parameters).  Interpreted on the CPU the plan still computes the reference forward, and the moments tensors take part in
    the storage sharing like any other tensor."""
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    pl = P.build_score_plan(32, 64, 16, fold_stats=True)
    kinds = [op.kind for op in pl.ops]
    assert kinds.count(P.INORM_STATS) == 25 - 14 and len(pl.ops) == 150 - 14
    selfn = [op for op in pl.ops if op.flags & P.PRO_NORM_SELF]
    assert len(selfn) == 14 and all(op.kind == P.CONV and op.ksize == 3 and op.flags & P.PRO_NORM and op.stats is None
                                    and op.norm_key == op.name.replace('conv', 'normalize') and op.src.h * op.src.w <= 64
                                    for op in selfn)
    assert not any(op.flags & P.PRO_NORM_SELF for op in P.build_score_plan(32, 64, 16).ops)
    prod = [op for op in pl.ops if op.moments is not None]
    fin = [op for op in pl.ops if op.kind == P.INORM_STATS and op.flags & P.PRO_NORM_MOMENTS]
    # seven tensors at full resolution (32 channels) and three at 32x8 (64 channels)
    assert len(prod) == 10 == len(fin) and all(op.flags & P.EPI_MOMENTS_OUT for op in prod)
    assert [f.src for f in fin] == [q.moments for q in prod] and all(f.geom is q.dst for f, q in zip(fin, prod))
    assert [m.elems for m in (q.moments for q in prod)] == [8 * 2 * 32] * 6 + [2 * 2 * 64] * 3 + [8 * 2 * 32]
    assert not any(op.flags & P.PRO_NORM_MOMENTS for op in pl.ops if op.kind != P.INORM_STATS)
    x = np.ascontiguousarray(g['x'][:2].transpose(0, 2, 3, 1))
    out = run_plan(pl, sd, x, np.full((2,), 1155))
    assert rel_err(out.transpose(0, 3, 1, 2), g['out'][1][:2]) < 2e-5
    # with fused pairs the normalizer's input comes out of a pair launch (no moments there): six of the seven fold
    plp = P.build_score_plan(32, 64, 16, fold_stats=True, fuse_pairs=True)
    assert sum(1 for op in plp.ops if op.kind == P.INORM_STATS and op.flags & P.PRO_NORM_MOMENTS) == 9
    # 256 x 64 arrays: 128 tiles per sample
    big = P.build_score_plan(32, 256, 64, fold_stats=True)
    mom = [op.moments for op in big.ops if op.moments is not None]
    assert len(mom) == 15
    assert sorted({(m.h, m.w) for m in mom}) == [(2, 64), (8, 64), (32, 64), (128, 32)]
    # nothing changes when the fold is not requested
    assert all(op.moments is None for op in P.build_score_plan(32, 64, 16).ops)


def test_fused_pair_plan_matches_reference_forward(weights64):
    """``build_score_plan(fuse_pairs=True)``: every 32-channel RCU block is ONE record (SBC_OP_CONV_PAIR) whose intermediate
    tensor does not exist; interpreted on the CPU the plan still computes the reference forward."""
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    pl = P.build_score_plan(32, 64, 16, fuse_pairs=True)
    kinds = [op.kind for op in pl.ops]
    # refine5: adapt_convs.0 (2 blocks) + output_convs (3 blocks) at 64x16
    # ... and refine5's CRP block there: two fused stages (SBC_OP_CONV_POOL) instead of 2 x (max pool + convolution)
    assert kinds.count(P.CONV_PAIR) == 5 and kinds.count(P.CONV_POOL) == 2 and kinds.count(P.CONV) == 111 - 10 - 2
    assert kinds.count(P.MAXPOOL5) == 10 and len(pl.ops) == 150 - 5 - 2
    pools = [op for op in pl.ops if op.kind == P.CONV_POOL]
    assert pools[0].flags == P.PRO_ELU and pools[0].res1 is None and pools[1].flags == P.EPI_RES1_ELU
    assert pools[1].src is pools[0].dst and pools[1].res2 is pools[0].dst and pools[1].res1 is pools[0].src
    assert P.count_conv_flops(pl) == 820772864
    assert sorted((op.src.h, op.src.w) for op in pl.ops if op.kind == P.CONV_PAIR) == [(64, 16)] * 5
    assert all(op.src.c == 32 and op.weight2 for op in pl.ops if op.kind == P.CONV_PAIR)
    x = np.ascontiguousarray(g['x'][:2].transpose(0, 2, 3, 1))
    out = run_plan(pl, sd, x, np.full((2,), 1155))
    assert rel_err(out.transpose(0, 3, 1, 2), g['out'][1][:2]) < 2e-5
    # wider arrays (config 5): 32- / 64-pixel rows and the 64-channel levels are fused in the fp16-weight mode only (scorenet
    # passes plan.PAIR_SHAPES_F16W)
    assert not any(op.kind == P.CONV_PAIR for op in P.build_score_plan(32, 256, 64, fuse_pairs=True).ops)
    big = P.build_score_plan(32, 256, 64, fuse_pairs=P.PAIR_SHAPES_F16W)
    shapes = sorted((op.src.c, op.src.h, op.src.w) for op in big.ops if op.kind == P.CONV_PAIR)
    assert shapes == [(32, 128, 32)] * 3 + [(32, 256, 64)] * 5 + [(64, 64, 16)] * 5 + [(64, 128, 32)] * 2
    assert P.count_conv_flops(big) == 13132365824
    small = P.build_score_plan(32, 64, 16, fuse_pairs=P.PAIR_SHAPES_F16W)                   # config 2 geometry in f16w mode
    assert sorted((op.src.c, op.src.h, op.src.w) for op in small.ops if op.kind == P.CONV_PAIR) == [(32, 64, 16)] * 5 + [(64, 16, 4)] * 0


def test_chain_plan_matches_reference_forward(weights64):
    """``build_score_plan(fuse_chain=True)``: the RCU / CRP runs and the ResidualBlocks without resampling or channel change of the
    8 x 2 and 16 x 4 levels are CHAIN records (csrc/conv_chain.hip), adjacent ones merged -- res5.0 + res5.1 + the whole of refine1
    is ONE record of six blocks: 68 records fewer than the plan it replaces, the same FLOPs, and interpreted on the CPU the same forward."""
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    base = P.build_score_plan(32, 64, 16, fold_stats=True, fuse_pairs=P.PAIR_SHAPES, fuse_res=True)
    pl = P.build_score_plan(32, 64, 16, fold_stats=True, fuse_pairs=P.PAIR_SHAPES, fuse_res=True, fuse_chain=True)
    chains = [op for op in pl.ops if op.kind == P.CHAIN]
    low, mid, top = [op for op in chains if op.src.h == 8], [op for op in chains if op.src.h == 16], [op for op in chains if op.src.h == 32]
    assert len(base.ops) == 125 and len(pl.ops) == 57 and len(low) == 8 and len(mid) == 3 and len(top) == 2
    # 32 x 8: refine4's adapt convolutions at 64 channels; its CRP + output RCU + refine5's second adapt pair at 32 (res2.1 stays on
    # its own launches: plan.chain_fusable)
    assert [(op.src.c, [b[0] for b in op.blocks]) for op in top] == [(64, [P.CHAIN_RCU] * 2), (32, [P.CHAIN_CRP] + [P.CHAIN_RCU] * 3)]
    assert [len(op.blocks) for op in low] == [1, 1, 6, 2, 2, 4, 2, 4] and [op.src.c for op in low] == [64, 128, 128, 128, 128, 64, 64, 64]
    assert [[b[0] for b in op.blocks] for op in low[:3]] == [[P.CHAIN_RES], [P.CHAIN_RES], [P.CHAIN_RES, P.CHAIN_RES, P.CHAIN_RCU, P.CHAIN_RCU, P.CHAIN_CRP, P.CHAIN_RCU]]
    assert [b[3]['dil'] for op in low[:3] for b in op.blocks if b[3]] == [1, 2, 4, 4] and low[2].blocks[0][3]['w3'] == 'res5.0.shortcut.weight'
    assert all(op.src.w == 2 and op.dst.c == op.src.c for op in low) and all(op.src.w == 4 and op.src.c == 64 for op in mid)
    assert [len(op.blocks) for op in mid] == [1, 2, 4]
    # of the 8 x 2 level's 53 convolutions only res4.0 (64 -> 128 channels: three launches) and the five MSF convolutions stay on their own
    assert sum(3 if (b[3] and b[3]['w3']) else 2 for op in low for b in op.blocks) == 45
    assert not any(op.kind == P.MAXPOOL5 for op in pl.ops)                 # (refine5's CRP: CONV_POOL records)
    assert not any(op.kind == P.INORM_STATS and (op.geom or op.src).h * (op.geom or op.src).w <= 64 for op in pl.ops)
    assert P.count_conv_flops(pl) == 820772864
    # a 256 x 64 array reaches 32 x 8 only at its lowest level: the RCU / CRP runs there (64 channels) are the only chains
    big = [op for op in P.build_score_plan(32, 256, 64, fuse_chain=True).ops if op.kind == P.CHAIN]
    assert big and all(op.src.h == 32 and op.src.w == 8 and op.src.c == 64 and all(b[0] != P.CHAIN_RES for b in op.blocks) for op in big)
    # ... and without folded statistics (statistics records elsewhere) the chained blocks still need none
    nf = P.build_score_plan(32, 64, 16, fuse_pairs=P.PAIR_SHAPES, fuse_chain=True)
    assert sum(op.kind == P.INORM_STATS for op in nf.ops) == 25 - 2 * 5
    x = np.ascontiguousarray(g['x'][:2].transpose(0, 2, 3, 1))
    for plan in (pl, nf):
        out = run_plan(plan, sd, x, np.full((2,), 1155))
        assert rel_err(out.transpose(0, 3, 1, 2), g['out'][1][:2]) < 2e-5


def test_end_statistics_in_the_end_convolution_plan_matches_reference_forward(weights64):
    """``build_score_plan(fuse_end=True)``: no statistics record for the normalizer -- the END_CONV record carries PRO_NORM_SELF and the
    norm's key (csrc/ops.hip: end_conv_self_kernel forms the statistics from the sample it holds); arrays whose sample does not fit a
    CU's LDS keep the record.  Interpreted on the CPU: the same forward."""
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    kw = dict(fold_stats=True, fuse_pairs=P.PAIR_SHAPES, fuse_res=True, fuse_chain=True, fuse_down=True)
    base, pl = P.build_score_plan(32, 64, 16, **kw), P.build_score_plan(32, 64, 16, fuse_end=True, **kw)
    assert len(base.ops) == 55 and len(pl.ops) == 54 and [op.name for op in base.ops[-2:]] == ['normalizer', 'end_conv']
    end = pl.ops[-1]
    assert end.kind == P.END_CONV and end.flags == P.PRO_NORM_SELF and end.norm_key == 'normalizer' and end.stats is None
    assert pl.ops[-2].kind == P.CONV_PAIR and P.count_conv_flops(pl) == 820772864
    big = P.build_score_plan(32, 256, 64, fuse_end=True)
    assert big.ops[-1].stats is not None and big.ops[-2].name == 'normalizer'
    assert P.build_score_plan(32, 32, 32, fuse_end=True).ops[-1].flags == P.PRO_NORM_SELF and P.build_score_plan(32, 32, 16, fuse_end=True).ops[-1].stats is not None
    x = np.ascontiguousarray(g['x'][:2].transpose(0, 2, 3, 1))
    out = run_plan(pl, sd, x, np.full((2,), 1155))
    assert rel_err(out.transpose(0, 3, 1, 2), g['out'][1][:2]) < 2e-5


def test_conv_down_plan_matches_reference_forward(weights64):
    """``build_score_plan(fuse_down=True)``: the pooled conv2 and the pooled 1x1 shortcut of res2.0 and res3.0 are ONE CONV_DOWN
    record each (csrc/conv_down.hip); res31.0 -- whose norm the consumer computes itself at 16x4 -- stays as it is.  Same FLOPs as
    the reference counts them, and interpreted on the CPU the same forward."""
    _, sd = weights64
    g = load_golden('forward_64x16.npz')
    base = P.build_score_plan(32, 64, 16, fold_stats=True, fuse_pairs=P.PAIR_SHAPES, fuse_res=True, fuse_chain=True)
    pl = P.build_score_plan(32, 64, 16, fold_stats=True, fuse_pairs=P.PAIR_SHAPES, fuse_res=True, fuse_chain=True, fuse_down=True)
    down = [op for op in pl.ops if op.kind == P.CONV_DOWN]
    assert len(base.ops) == 57 and len(pl.ops) == 55 and [op.name for op in down] == ['res2.0.down', 'res3.0.down']
    assert [(op.src.w, op.src.c, op.dst.c, op.dst.h) for op in down] == [(16, 32, 64, 32), (8, 64, 64, 16)]
    assert all(op.res1 is not None and op.stats is not None and op.weight.endswith('conv2.conv.weight') and op.weight2.endswith('shortcut.conv.weight') for op in down)
    assert P.count_conv_flops(pl) == 820772864
    x = np.ascontiguousarray(g['x'][:2].transpose(0, 2, 3, 1))
    out = run_plan(pl, sd, x, np.full((2,), 1155))
    assert rel_err(out.transpose(0, 3, 1, 2), g['out'][1][:2]) < 2e-5


def test_skip_branches_on_lanes_keep_the_dataflow_and_never_share_live_storage():
    """plan.hoist_skip_branches (round 6): the decoder's skip branches move behind their anchors onto launch lanes.  The list stays a
    valid sequential order (every input is produced earlier), every wait names an event an EARLIER record signals, and no record
    writes a slot that a concurrently running lane still reads or writes: for a lane record issued at i and joined at j, its input and
    output slots are touched by no other record in (i, j)."""
    kw = dict(fold_stats=True, fuse_pairs=P.PAIR_SHAPES, fuse_res=True, fuse_chain=True, fuse_down=True, fuse_end=True)
    seq = P.build_score_plan(32, 64, 16, **kw)
    pl = P.build_score_plan(32, 64, 16, skip_overlap=True, **kw)
    assert sorted(o.name for o in seq.ops) == sorted(o.name for o in pl.ops)
    assert P.count_conv_flops(pl) == 820772864
    made, signalled = {id(pl.x)}, set()
    for op in pl.ops:
        assert all(id(t) in made for t in op.inputs()), op.name
        assert all(e in signalled for e in op.wait), op.name
        made.update(id(t) for t in op.outputs())
        if op.signal:
            signalled.add(op.signal)
    lanes = [i for i, o in enumerate(pl.ops) if o.lane]
    assert len(lanes) == 6 and max(o.lane for o in pl.ops) < P.MAX_LANES
    for i in lanes:
        op = pl.ops[i]
        # the join: the first run-stream record waiting for an event this lane signals at or behind record i
        j = min(m for k in range(i, len(pl.ops)) if pl.ops[k].lane == op.lane and pl.ops[k].signal
                for m in range(k + 1, len(pl.ops)) if pl.ops[m].lane == 0 and pl.ops[k].signal in pl.ops[m].wait)
        mine = {t.slot for t in op.inputs() + op.outputs()}
        for m in range(i + 1, j):
            other = pl.ops[m]
            if other.lane == op.lane:
                continue                      # same lane: list order
            assert not mine & {t.slot for t in other.outputs()}, (op.name, other.name)
            assert not {t.slot for t in op.outputs()} & {t.slot for t in other.inputs()}, (op.name, other.name)
