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
