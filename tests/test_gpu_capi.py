"""The self-sufficient C boundary (``sbc_score_*``, include/sbc_hip.h): a host that is not Python gets the whole score
network from the checkpoint tensors in one call.  Held to the Python host (plan.py / scorenet.py): identical operator
records, bit-identical outputs, and a full Langevin-step plan composed from ``sbc_score_ops``.  ``pytest -m gpu``."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

MODES = {'bf16x3': 0, 'f32': 1, 'f16w': 2, 'f16x2': 3}


def _create(sd, cfg, B, nt, nr, mode, pairs=False, fold=False, res=False, chain=False, down=False, end=False, lanes=False):
    from score_based_channels_amd import _lib
    keep = {k: np.ascontiguousarray(v, np.float32) for k, v in sd.items() if k != 'sigmas'}
    refs = (_lib.sbc_tensor_ref * len(keep))(*[
        _lib.sbc_tensor_ref(name=k.encode(), data=v.ctypes.data, numel=v.size) for k, v in keep.items()])
    sig = np.ascontiguousarray(sd['sigmas'], np.float32)
    desc = _lib.sbc_score_desc(ngf=32, channels=2, nt=nt, nr=nr, batch=B, conv_mode=MODES[mode], sigmas=sig.ctypes.data,
                               num_classes=sig.size, flags=(1 if pairs else 0) | (2 if fold else 0) | (4 if res else 0) | (8 if chain else 0) | (16 if down else 0) | (32 if end else 0) | (64 if lanes else 0))
    h = C.c_void_p()
    _lib.check(_lib.lib().sbc_score_create(C.byref(desc), refs, len(keep), C.byref(h)))
    return h


@pytest.mark.parametrize('mode', ['bf16x3', 'f32', 'f16w', 'f16x2', 'f16x2+pairs', 'f16w+pairs', 'f16x2+pairs+fold', 'bf16x3+fold',
                                  'f16x2+pairs+fold+res', 'f16x2+pairs+fold+res+chain', 'f16x2+pairs+fold+res+chain+down', 'f16x2+pairs+fold+res+chain+down+end',
                                  'bf16x3+end', 'f16x2+pairs+fold+res+chain+down+end+lanes', 'bf16x3+lanes'])
def test_c_built_score_network_equals_python_host(weights64, mode):
    import torch
    from score_based_channels_amd import _lib
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    L = _lib.lib()
    B, nt, nr = 3, 64, 16
    mode, *opts = mode.split('+')
    pairs, fold, res, chain, down, end = 'pairs' in opts, 'fold' in opts, 'res' in opts, 'chain' in opts, 'down' in opts, 'end' in opts
    lanes = 'lanes' in opts                 # SBC_SCORE_SKIP_LANES: the skip branches on launch lane 1 (plan.hoist_skip_branches)
    h = _create(sd, cfg, B, nt, nr, mode, pairs, fold, res, chain, down, end, lanes)
    try:
        ops_p, n = C.POINTER(_lib.sbc_op)(), C.c_int32()
        _lib.check(L.sbc_score_ops(h, C.byref(ops_p), C.byref(n)))
        net = ScoreNet(cfg, conv_mode=mode, fuse_pairs=pairs, fold_stats=fold, fuse_res=res, fuse_chain=chain, fuse_down=down, fuse_end=end).cuda().load_state_dict(sd)
        # (record for record against the plan the flags ask for: sequential, or -- SBC_SCORE_SKIP_LANES -- with the skip branches on a launch
        # lane; the forward below runs whatever the Python host picks for a batch this small and must agree bit for bit either way)
        bound = net.bind(B, nt, nr, lanes=lanes)
        assert lanes == any(o.lane for o in bound.ops)
        assert n.value == len(bound.ops)
        assert chain == any(o.kind == 24 for o in bound.ops) and down == any(o.kind == 25 for o in bound.ops)
        # identical records: every scalar field, and the same storage-sharing pattern (pointers renamed by first use)
        ids_c, ids_p = {}, {}
        for i, ref in enumerate(bound.ops):
            got = ops_p[i]
            for f in ('kind', 'flags', 'B', 'H', 'W', 'cin', 'cout', 'ksize', 'dil', 'up_h', 'up_w', 'tag', 'lane', 'signal'):
                assert getattr(got, f) == getattr(ref, f), (i, f)
            assert list(got.wait) == list(ref.wait), (i, 'wait')
            for f in ('in_', 'out', 'stats', 'res1', 'res2', 'up', 'aux'):
                a, b = getattr(got, f), getattr(ref, f)
                assert (a is None) == (b is None), (i, f)
                if a is not None:
                    assert ids_c.setdefault(a, len(ids_c)) == ids_p.setdefault(b, len(ids_p)), (i, f)
            for f in ('weight', 'bias', 'weight_wino', 'weight_split', 'weight2_split', 'bias2', 'norm2'):
                if ref.kind == 25 and f in ('weight', 'weight_wino'):
                    # SBC_OP_CONV_DOWN: calibration-only pointers at the layers' UNPOOLED forms, which the Python host packs for other
                    # array sizes and a C handle (one size, only the forms it runs) does not have (include/sbc_hip.h)
                    assert getattr(got, f) is None
                    continue
                assert (getattr(got, f) is None) == (getattr(ref, f) is None), (i, f)
            if ref.kind == 24:                      # SBC_OP_CHAIN: the same blocks in the same order
                cg, cr = C.cast(got.ext, C.POINTER(_lib.sbc_chain)).contents, C.cast(ref.ext, C.POINTER(_lib.sbc_chain)).contents
                assert cg.n_blocks == cr.n_blocks and list(cg.type)[:cg.n_blocks] == list(cr.type)[:cr.n_blocks]
                assert all(cg.w1[k] and cg.w2[k] and cr.w1[k] and cr.w2[k] for k in range(cg.n_blocks))
            elif ref.kind == 3:                     # (fused records carry the Winograd forms for the calibration only where the host packed them)
                assert (got.weight_wino_split is None) == (ref.weight_wino_split is None), (i, 'weight_wino_split')
        # bit-identical forward
        g = load_golden('forward_64x16.npz')
        x = torch.from_numpy(g['x'][:B]).cuda()
        labels = torch.tensor([0, 1155, 2310], dtype=torch.long, device='cuda')
        want = net(x, labels)
        px, po, pl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(L.sbc_score_buffers(h, C.byref(px), C.byref(po), C.byref(pl)))
        xin = x.permute(0, 2, 3, 1).contiguous()
        hip = torch.cuda.current_stream().cuda_stream
        out = torch.empty(B, nt, nr, 2, device='cuda')
        lib_rt = C.CDLL('libamdhip64.so')
        lib_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        torch.cuda.synchronize()
        assert lib_rt.hipMemcpy(px, C.c_void_p(xin.data_ptr()), xin.numel() * 4, 3) == 0            # device to device
        assert lib_rt.hipMemcpy(pl, C.c_void_p(labels.data_ptr()), B * 8, 3) == 0
        _lib.check(L.sbc_score_forward(h, C.c_void_p(hip)))
        torch.cuda.synchronize()
        assert lib_rt.hipMemcpy(C.c_void_p(out.data_ptr()), po, out.numel() * 4, 3) == 0
        assert torch.equal(out.permute(0, 3, 1, 2), want)
    finally:
        L.sbc_score_destroy(h)


@pytest.mark.parametrize('mode,pairs,fold,chain', [('bf16x3', False, False, False), ('f16x2', True, True, False), ('f16x2', True, True, True)])
def test_langevin_plan_composed_from_c_records(weights64, mode, pairs, fold, chain):
    """sbc_score_level_source + sbc_score_ops + SBC_OP_LANGEVIN + SBC_OP_STEP_INC = the plan AldBatch builds: same NMSE log --
    in the exact mode and in the shipped default (conv_mode 3 | SBC_SCORE_FUSE_PAIRS | SBC_SCORE_FOLD_STATS, calibrated scales)."""
    import torch
    from score_based_channels_amd import _lib, plan as P
    from score_based_channels_amd.ald import AldBatch, schedule_tables, snr_to_noise
    from score_based_channels_amd.noise import HostNoise
    from score_based_channels_amd.scorenet import ScoreNet
    cfg, sd = weights64
    L = _lib.lib()
    g = load_golden('ald_plumbing_3levels.npz')
    H, Pm = g['H'], g['P']
    B, nt, nr = H.shape
    npil = Pm.shape[1]
    levels, n_steps = [0, 1, 2], 9
    noise = HostNoise(int(g['seed']))
    ln = float(snr_to_noise(g['snr_db'], nt)[0])
    steps = noise.step_block(0, H.shape, n_steps)
    # reference run through the Python host
    net = ScoreNet(cfg, conv_mode=mode, fold_stats=fold, fuse_pairs=pairs, fuse_res=False, fuse_chain=chain, fuse_down=False, fuse_end=False).cuda().load_state_dict(sd)
    ald = AldBatch(net, H, Pm, np.arange(B), np.arange(B), ln, levels=levels, step_noise=torch.from_numpy(steps))
    ald.set_init(torch.from_numpy(noise.init(H.shape)))
    Y = ald.synthesize_measurements(torch.from_numpy(noise.measurement(0, (B, npil, nr)))).clone()
    ald.run()
    torch.cuda.synchronize()
    want = ald.nmse_log().clone()
    # the same plan from C records
    h = _create(sd, cfg, B, nt, nr, mode, pairs, fold, False, chain)
    try:
        sched, sig = schedule_tables(sd['sigmas'], cfg.model.sigma_end, levels, 3, [3e-11], [0.01], [ln])
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in dict(
            sched=sched, sig=sig, H=H.view(np.float32), P=Pm.view(np.float32), nz=steps.view(np.float32)).items()}
        step = torch.zeros(1, dtype=torch.int32, device='cuda')
        nm = torch.zeros(n_steps, B, device='cuda')
        _lib.check(L.sbc_score_level_source(h, C.c_void_p(dev['sig'].data_ptr()), C.c_void_p(step.data_ptr())))
        ops_p, n = C.POINTER(_lib.sbc_op)(), C.c_int32()
        _lib.check(L.sbc_score_ops(h, C.byref(ops_p), C.byref(n)))
        px, po = C.c_void_p(), C.c_void_p()
        _lib.check(L.sbc_score_buffers(h, C.byref(px), C.byref(po), None))
        ext = _lib.sbc_langevin(X=px, score=po, P=dev['P'].data_ptr(), Y=torch.view_as_real(Y).data_ptr(),
                                Htrue=dev['H'].data_ptr(), sched=dev['sched'].data_ptr(), noise=dev['nz'].data_ptr(),
                                nmse=nm.data_ptr(), step=step.data_ptr(), n_steps=n_steps, Nt=nt, Nr=nr, Np=npil)
        recs = [ops_p[i] for i in range(n.value)]
        recs.append(_lib.sbc_op(kind=P.LANGEVIN, B=B, ext=C.cast(C.pointer(ext), C.c_void_p)))
        recs.append(_lib.sbc_op(kind=P.STEP_INC, B=1, out=step.data_ptr()))
        plan = _lib.Plan(recs)
        x0 = torch.view_as_real(torch.from_numpy(noise.init(H.shape)).cuda()).contiguous()
        lib_rt = C.CDLL('libamdhip64.so')
        lib_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        torch.cuda.synchronize()
        assert lib_rt.hipMemcpy(px, C.c_void_p(x0.data_ptr()), x0.numel() * 4, 3) == 0
        plan.run(torch.cuda.current_stream().cuda_stream, n_steps)
        torch.cuda.synchronize()
        assert torch.equal(nm, want)
        assert np.max(np.abs(nm.cpu().numpy() / g['nmse_log'][0] - 1)) < 1e-5
        plan.close()
    finally:
        L.sbc_score_destroy(h)


def test_score_create_reports_missing_tensors(weights64):
    from score_based_channels_amd import _lib
    cfg, sd = weights64
    bad = {k: v for k, v in sd.items() if k != 'refine3.msf.convs.1.bias'}
    with pytest.raises(_lib.SbcError, match='refine3.msf.convs.1.bias'):
        _create(bad, cfg, 2, 64, 16, 'bf16x3')


def test_plan_refuses_inconsistent_lanes():
    """sbc_plan_create validates the launch-lane fields of its records (ABI 14): a lane beyond SBC_MAX_LANES, an event id beyond
    SBC_MAX_EVENTS, a wait for an event no EARLIER record signals, lanes mixed with the SBC_OP_SIDE / SBC_OP_JOIN flags -- all refused
    with a message, none of them a hang at run time."""
    import torch
    from score_based_channels_amd import _lib, plan as P
    x = torch.zeros(2, 8, 8, 2, device='cuda')
    y = torch.zeros(1, dtype=torch.int32, device='cuda')

    def inc(**kw):
        return _lib.sbc_op(kind=P.STEP_INC, B=1, out=C.c_void_p(y.data_ptr()), **kw)
    ok = _lib.Plan([inc(signal=1), inc(lane=1, wait=(C.c_int32 * 2)(1, 0), signal=2), inc(wait=(C.c_int32 * 2)(2, 0))])
    ok.run(torch.cuda.current_stream().cuda_stream, 3)
    torch.cuda.synchronize()
    assert int(y.item()) == 9
    ok.close()
    for bad, what in (([inc(lane=4)], 'lane out of range'), ([inc(signal=65)], 'signal id out of range'),
                      ([inc(wait=(C.c_int32 * 2)(1, 0)), inc(signal=1)], 'no earlier record signals'),
                      ([inc(signal=1), inc(lane=1, flags=P.OP_SIDE, wait=(C.c_int32 * 2)(1, 0))], 'do not mix')):
        with pytest.raises(_lib.SbcError, match=what):
            _lib.Plan(bad)
    del x
