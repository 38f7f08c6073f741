"""Shared driver of the two inference scripts: shard a list of lock-step trajectories over the ranks, run
them through ``AldBatch`` (in chunks that fit the device), gather the NMSE logs.

Replaces the triple loop of ``test_score.py:118-171`` / ``tune_hparams_score.py:100-148`` (SNR point -> noise
level -> inner step, one host synchronisation per step) by one asynchronous ``plan.run`` per chunk.
"""
import os

import numpy as np
import torch

from . import _lib, shard
from .ald import AldBatch
from .config import DEFAULT_STREAMS        # two concurrent sub-batch streams fill the gaps of the low-resolution launches


# Launch mode of a Langevin step the CLIs and bench.py default to: False = eager launches, True = hipGraph replay of the
# ~150-launch step.  Both give bit-identical results (tests/test_gpu_parity.py::test_ald_graph_replay_equals_eager); the
# faster one on MI355X is the default -- see DESIGN.md section 5 for the measured pair.
DEFAULT_USE_GRAPH = False
# how long a host thread of run_concurrently waits for its partner thread to have QUEUED its launches (not for the GPU): a leader
# that hangs inside a runtime call after a device fault must surface as an error of the follower, not as a join() that never returns
HOST_WAIT_S = 600


def resolve_launch_mode(args):
    """``--graph`` / ``--no_graph`` of the CLIs -> use_graph (default ``DEFAULT_USE_GRAPH``)."""
    if getattr(args, 'graph', False) and getattr(args, 'no_graph', False):
        raise SystemExit('--graph and --no_graph are mutually exclusive')
    if getattr(args, 'graph', False):
        return True
    if getattr(args, 'no_graph', False):
        return False
    return DEFAULT_USE_GRAPH


def level_subset(num_classes, stride=1, num_levels=None):
    """Noise levels to walk: all of them (reference behaviour), every ``stride``-th plus the last one, or the
    first ``num_levels``.  Truncation is an addition of this build for quick runs; it is never applied silently."""
    levels = list(range(0, num_classes, max(1, int(stride))))
    if levels[-1] != num_classes - 1:
        levels.append(num_classes - 1)
    if num_levels is not None:
        levels = levels[:int(num_levels)]
    return levels


def shared_init(n_channels, nt, nr, seed, combo):
    """``init_val_H = torch.randn_like(val_H)`` (test_score.py:115): one CN(0,1) draw per channel, shared by all
    SNR points of a combination.  CPU generator keyed by (seed, combo) => independent of sharding."""
    g = torch.Generator().manual_seed((int(seed) * 1000003 + int(combo)) % (2 ** 63 - 1))
    return torch.randn(n_channels, nt, nr, dtype=torch.complex64, generator=g)


def batch_limit(net, nt, nr, requested, reserve=0.25):
    """Largest lock-step batch for an ``nt x nr`` array: bounded by the caller's ``requested`` size, by the 32-bit element
    index of the convolution kernels (``B * Nt * Nr * ngf`` elements in the largest tensor) and by the free device memory
    (the activation slots of ``plan.assign_slots`` dominate: ~1 MB per trajectory at 64x16, 16 MB at 256x64), keeping
    ``reserve`` of it for the rest of the process.  In ``conv_mode f16x2`` a flagged chunk is re-run with the UNFUSED ``bf16x3``
    plan (``net.fallback_plan_slot_elems``: ~15 % more slot memory at 64x16): the chunk is sized for the larger of the two, so the
    re-run cannot run out of memory exactly when it is needed."""
    slot_elems = sum(net.score_plan(nt, nr).slot_elems)
    if getattr(net, 'conv_mode', None) == 'f16x2' and hasattr(net, 'fallback_plan_slot_elems'):
        slot_elems = max(slot_elems, net.fallback_plan_slot_elems(nt, nr))
    per_traj = 4 * slot_elems + 64 * nt * nr
    free, _ = torch.cuda.mem_get_info(net.device)
    by_mem = int(free * (1.0 - reserve)) // per_traj
    by_index = 0x7fffffff // (nt * nr * net.ngf)                       # wider tensors exist only at lower resolution
    return max(1, min(int(requested), by_mem, by_index))


def stream_count(net, T, nt, nr, requested=None):
    """Sub-batch streams for a lock-step chunk of ``T`` trajectories: the requested count (default ``config.DEFAULT_STREAMS``), except
    that a chunk small enough for the plan with launch lanes (``ScoreNet.skip_overlap_for``: a rank's share of a sharded test_score
    run) stays ONE batch -- its low-resolution launches are latency-bound, halving the batch does not shorten them, and the skip
    branches already fill the idle CUs from a lane of the same plan (213 trajectories: 1.06 ms per step sequential, 1.15 as two
    sub-batches, 0.98 with lanes; 425: 1.60 / 1.55 / 1.47)."""
    n = DEFAULT_STREAMS if requested is None else max(1, int(requested))
    if n > 1 and getattr(net, 'skip_overlap_for', None) is not None and net.skip_overlap_for(int(T), nt, nr):
        return 1
    return n


def run_concurrently(batches, streams, n_steps, use_graph=False):
    """Advance several ``AldBatch`` objects by ``n_steps`` each, every one on its own HIP stream and fed by its own host
    thread.  A Langevin step is ~150 dependent launches, many of them (the 8x2 / 16x4 levels) with fewer workgroups than the
    chip has CUs; two independent sub-batches in flight let the GPU fill one's gaps with the other's kernels.  One thread
    per stream matters: ``plan.run`` returns only when its launches are queued, so sequential calls would queue one
    stream's whole schedule ahead of the other's (ctypes releases the GIL during the call, the threads really overlap)."""
    import threading
    if len(batches) == 1:
        with torch.cuda.stream(streams[0]):
            batches[0].run(n_steps, use_graph=use_graph)
        return
    errors = []
    # two (or more) streams: the persistent kernels take half the CUs each, so that both streams' launches are resident side by side
    # (include/sbc_hip.h: sbc_set_persistent_cus; -0.7 % per step, identical results)
    dev = batches[0].net.device
    # (eager launches only: replayed graphs of two host threads do not overlap at kernel level, DESIGN.md section 13, and would just run
    # at half width)
    # (... and small sub-batches only: at 10 200 trajectories per stream every launch fills the chip for a millisecond, the streams
    # gain nothing from each other and half-width grids cost 20 % -- 74.6 against 62.1 ms per step of the 20 400-trajectory workload;
    # sustained gain +2.3 % at 1275 trajectories per stream, +0.9 % at 1700, +0.5 % at 2550 where a 20-step segment already loses 1 %)
    small = max(b.T * b.nt * b.nr for b in batches) <= int(os.environ.get('SBC_STREAM_SMALL_PX', 1 << 21))   # (the variable: A/B aid)
    # (the width is a field of each batch's plans -- sbc_plan_set_persistent_cus -- not process state: concurrent calls on other host
    # threads or devices do not see it, and it is reset in the `finally` below whatever happens in between)
    half = torch.cuda.get_device_properties(dev).multi_processor_count // 2 if (not use_graph and small) else 0

    # ... and every second stream walks the schedule ~0.45 of a step behind its neighbour (AldBatch.run_leading / run_following: the
    # lag is the head of the leader's first step, run alone; ~4 % of every step after it)
    lag = (not use_graph and small and n_steps >= int(os.environ.get('SBC_STREAM_LAG_MIN_STEPS', '2'))       # (the variable: tests)
           and not os.environ.get('SBC_NO_STREAM_LAG') and not any(b.net.overlap or b.uses_lanes for b in batches))
    # pair (2j, 2j + 1): two device events (leader's head done; leader done) and the host flags that say they have been RECORDED --
    # a stream that waits for an event nobody has recorded yet does not wait at all
    class _Pair:
        def __init__(self):
            self.head_evt, self.done_evt = torch.cuda.Event(), torch.cuda.Event()
            self.head_flag, self.done_flag = threading.Event(), threading.Event()
    pairs = [_Pair() for _ in range(len(batches) // 2)] if lag else []

    def work(b, st, k):
        pr = pairs[k // 2] if k // 2 < len(pairs) else None
        try:
            torch.cuda.set_device(b.net.device)
            with torch.cuda.stream(st):
                if pr is None:
                    b.run(n_steps, use_graph=use_graph)
                elif k % 2 == 0:
                    def head_done():
                        pr.head_evt.record(st)
                        pr.head_flag.set()
                    b.run_leading(n_steps, head_done, half)
                    pr.done_evt.record(st)
                else:
                    def wait_head():
                        if not pr.head_flag.wait(HOST_WAIT_S):
                            raise RuntimeError('the leading stream did not queue the head of its first step within %d s' % HOST_WAIT_S)
                        st.wait_event(pr.head_evt)

                    def wait_leader():
                        if not pr.done_flag.wait(HOST_WAIT_S):
                            raise RuntimeError('the leading stream did not queue its steps within %d s' % HOST_WAIT_S)
                        st.wait_event(pr.done_evt)
                    b.run_following(n_steps, wait_head, wait_leader, half)
        except BaseException as e:                        # surfaced in the caller's thread
            errors.append(e)
        finally:
            # whatever happened to the leader (even before its first launch), its follower must not wait on the host for ever: a
            # stream that waits for an event nobody recorded does not wait
            if pr is not None and k % 2 == 0:
                pr.head_flag.set()
                pr.done_flag.set()
    threads = [threading.Thread(target=work, args=(b, st, k)) for k, (b, st) in enumerate(zip(batches, streams))]
    started = []
    try:
        for b in batches:
            b.set_persistent_cus(half)
        for t in threads:
            t.start()
            started.append(t)
    finally:
        for t in started:
            t.join()
        for b in batches:
            b.set_persistent_cus(0)
    if errors:
        raise errors[0]


def host_noise_streams(seed, combo, shape, n_snr, n_steps, meas_shape):
    """The Gaussian draws of one (spacing, pilot_alpha) combination from the keyed host streams of ``noise.HostNoise``
    -- the reference's draw order: one initial estimate shared by all SNR points, then per SNR point
    one measurement-noise draw and one draw per Langevin step -- laid out for the lock-step batch ``t = snr * B + b``.
    Returns (init ``[B, Nt, Nr]`` torch, meas ``[S*B, Np, Nr]``, steps ``[n_steps, S*B, Nt, Nr]``)."""
    from .noise import HostNoise
    noise = HostNoise(seed, combo)
    init = torch.from_numpy(noise.init(shape))
    meas = np.concatenate([noise.measurement(s, meas_shape) for s in range(n_snr)], axis=0)
    steps = np.concatenate([noise.step_block(s, shape, n_steps) for s in range(n_snr)], axis=1)
    return init, meas, steps


def run_trajectories(net, Htrue, P, h_index, p_index, local_noise, alpha_step, beta_noise, levels, steps_each,
                     seed, init, traj_base=0, max_batch=4096, use_graph=None, rank=0, world=1, n_streams=None,
                     return_final=False, n_steps=None, dc_boost=1.0, init_index=None, Y=None, y_index=None,
                     step_noise=None, meas_noise=None, info=None, traj_id=None):
    """Run ``T = len(h_index)`` trajectories, sharded over ``world`` ranks; returns the full NMSE log
    ``[n_steps, T]`` (float32 numpy, identical on every rank).  ``init``: ``[nH, Nt, Nr]`` complex64 initial
    estimates indexed by ``h_index``.  Trajectory ``t`` draws its noise from Philox stream ``traj_base + t`` (or ``traj_id[t]``
    when the array is given: several test profiles in one list reuse the ids of their single-profile runs, ``test_score --test A B``).

    ``n_streams`` > 1 runs a chunk as independent sub-batches on concurrent HIP streams, one host thread each
    (``run_concurrently``; results do not depend on the split: per-trajectory noise keys, per-sample normalisation).

    ``init_index`` (default ``h_index``) selects the initial estimate of each trajectory; ``Y`` ``[nY, Np, Nr]`` with
    ``y_index`` supplies measurements shared by several trajectories instead of synthesising one per trajectory
    (the 50 chains per sample of ``test_mmse.py:185-193``).  ``return_final``: also return the final estimates ``[T, Nt, Nr]`` complex64 (``--save_channels``); ``n_steps``: stop
    after that many Langevin steps (early stop of ``test_mmse.py:246-250``).

    ``step_noise`` ``[n_steps, T, Nt, Nr]`` / ``meas_noise`` ``[T, Np, Nr]`` complex64 (host arrays) replay externally
    drawn CN(0,1) noise instead of the in-kernel Philox streams (``--noise host`` parity runs against the reference).

    ``conv_mode f16x2``: a chunk whose launches raised the device's range flag (an activation outside the window in which the
    two-term fp16 split is fp32-class, ``sbc_range_flag``) is run again with ``net.fallback_net()`` (``bf16x3``) before its
    results are used -- a warning is printed, and ``info`` (a dict) receives one ``'f16x2_fallback'`` record per such chunk, gathered
    from EVERY rank (each record carries its ``rank``).  (A re-run replaces the whole chunk, so with a fallback the results of the
    chunk's other trajectories are the ``bf16x3`` ones: fp32-class either way, but no longer independent of ``max_batch`` bit for bit.)

    Multi-rank failure protocol (``shard.check_peers``): the ranks agree that nobody failed before they enter the final gather; a rank
    whose run raised makes the others raise ``shard.PeerFailure`` instead of waiting for the collective timeout."""
    from . import _lib
    use_graph = DEFAULT_USE_GRAPH if use_graph is None else bool(use_graph)
    n_streams = DEFAULT_STREAMS if n_streams is None else int(n_streams)
    h_index = np.asarray(h_index, np.int64)
    T = len(h_index)
    bc = lambda a: np.broadcast_to(np.asarray(a), (T,))            # noqa: E731
    p_index, local_noise, alpha_step, beta_noise = bc(p_index), bc(local_noise), bc(alpha_step), bc(beta_noise)
    init_index = h_index if init_index is None else np.asarray(init_index, np.int64)
    traj_id = traj_base + np.arange(T, dtype=np.int64) if traj_id is None else np.asarray(traj_id, np.int64)
    if traj_id.shape != (T,):
        raise ValueError('traj_id must have one entry per trajectory')
    lo, hi = shard.my_block(T, rank, world)
    n_all = len(levels) * steps_each
    n_steps = n_all if n_steps is None else min(int(n_steps), n_all)
    local = torch.zeros(n_steps, hi - lo, dtype=torch.float32, device=net.device)
    nt, nr = init.shape[-2], init.shape[-1]
    final = torch.zeros(hi - lo, nt, nr, dtype=torch.complex64, device=net.device) if return_final else None
    cur = torch.cuda.current_stream(net.device)
    max_batch = batch_limit(net, nt, nr, max_batch)
    # (a chunk small enough for the launch-lane plan runs as one batch: stream_count)
    streams = [torch.cuda.Stream(net.device) for _ in range(stream_count(net, min(max_batch, hi - lo), nt, nr, n_streams))]

    def run_chunk(use_net, c0, c1):
        running = []
        for part, st in zip(np.array_split(np.arange(c0, c1), len(streams)), streams):
            if len(part) == 0:
                continue
            st.wait_stream(cur)
            with torch.cuda.stream(st):                     # set-up on the sub-batch's stream; the walk itself below
                sn = None if step_noise is None else torch.from_numpy(np.ascontiguousarray(step_noise[:n_steps, part]))
                ald = AldBatch(use_net, Htrue, P, h_index[part], p_index[part], local_noise[part], alpha_step[part],
                               beta_noise[part], levels=levels, steps_each=steps_each, seed=seed,
                               traj_id=traj_id[part], dc_boost=dc_boost, step_noise=sn, lanes=None if len(streams) == 1 else False)
                ald.set_init(init[torch.from_numpy(init_index[part])])
                if Y is None:
                    ald.synthesize_measurements(None if meas_noise is None else torch.from_numpy(meas_noise[part]))
                else:
                    ald.set_measurements(Y[torch.from_numpy(np.asarray(y_index)[part])])
            running.append((part, ald, st))
        run_concurrently([r[1] for r in running], [r[2] for r in running], n_steps, use_graph)
        for part, ald, st in running:
            cur.wait_stream(st)
            local[:, part[0] - lo:part[-1] + 1 - lo] = ald.nmse_log()[:n_steps]
            if return_final:
                final[part[0] - lo:part[-1] + 1 - lo] = ald.X
        torch.cuda.synchronize(net.device)              # the copies above read the chunk's buffers
        for _, ald, _ in running:
            ald.close()
        del running

    f16x2 = getattr(net, 'conv_mode', None) == 'f16x2'
    fallbacks = []
    if f16x2:
        _lib.range_flag(True, net.device)               # whatever an earlier, unrelated run left behind
    for c0 in range(lo, hi, max_batch):
        c1 = min(hi, c0 + max_batch)
        run_chunk(net, c0, c1)
        if f16x2:
            bits = _lib.range_flag(True, net.device)
            if bits:
                # an activation of this chunk left the window in which the two-term fp16 split is fp32-class (sbc_range_flag):
                # the same trajectories again in bf16x3 -- same noise keys, same measurements -- in this process
                rec = {'rank': int(rank), 'trajectories': [int(c0), int(c1)], 'traj_id_first': int(traj_id[c0]),
                       'range_flag': int(bits), 'reason': _lib.describe_range(bits), 'rerun_in': 'bf16x3'}
                import sys
                print('WARNING [rank %d]: f16x2 range flag %d on trajectories [%d, %d) -- %s; re-running the chunk in bf16x3'
                      % (rank, bits, c0, c1, rec['reason']), file=sys.stderr, flush=True)
                run_chunk(net.fallback_net(), c0, c1)
                fallbacks.append(rec)
    torch.cuda.synchronize(net.device)
    shard.check_peers(world, 'the gather of the NMSE logs')        # a rank that raised above never gets here: see shard.report_failure
    if info is not None:
        for recs in shard.gather_objects(fallbacks, world):
            info.setdefault('f16x2_fallback', []).extend(recs)
    full = shard.gather_trajectory_logs(local, T, rank, world)
    if not return_final:
        return full.cpu().numpy()
    flat = torch.view_as_real(final).reshape(hi - lo, -1).t().contiguous()          # [2 Nt Nr, T_local]
    est = shard.gather_trajectory_logs(flat, T, rank, world).t().contiguous().view(T, nt, nr, 2)
    return full.cpu().numpy(), torch.view_as_complex(est).cpu().numpy()
