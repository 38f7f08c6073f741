"""World-size-2 test of the multi-GPU path on CPU (gloo): block sharding + the one all_gather of NMSE logs, and the SUM
all-reduce that data-parallel training uses for its flat gradient buffer."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_items, n_steps, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from score_based_channels_amd import shard
    r, w, _ = shard.init_distributed('gloo')
    assert (r, w) == (rank, world)
    lo, hi = shard.my_block(n_items, r, w)
    full = torch.arange(n_steps * n_items, dtype=torch.float32).view(n_steps, n_items)   # "NMSE of trajectory t at step k"
    got = shard.gather_trajectory_logs(full[:, lo:hi].clone(), n_items, r, w)
    # the gradient all-reduce of data-parallel training (train.TrainNet.step with world > 1)
    g = torch.full((1000,), float(rank + 1))
    shard.all_reduce_sum_(g, w)
    ok_sum = bool(torch.equal(g, torch.full((1000,), float(sum(range(1, world + 1))))))
    q.put((rank, bool(torch.equal(got, full)) and ok_sum, (lo, hi)))
    dist.destroy_process_group()


def test_sharded_logs_gather_to_the_full_log():
    ctx = mp.get_context('spawn')
    for n_items in (17, 34):                      # uneven and even split
        q = ctx.Queue()
        port = 29500 + os.getpid() % 500 + n_items
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, 6, q)) for r in range(2)]
        [p.start() for p in procs]
        res = sorted(q.get(timeout=120) for _ in procs)
        [p.join(60) for p in procs]
        assert [ok for _, ok, _ in res] == [True, True]
        assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == n_items


def _failing_worker(rank, world, port, q):
    """Rank 1 raises before the gather; rank 0 must learn of it at its agreement point (shard.check_peers) instead of sitting in the
    all_gather until the process-group timeout.  Also: the fallback records of every rank reach every rank (gather_objects)."""
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from score_based_channels_amd import shard
    r, w, _ = shard.init_distributed('gloo')
    t0 = time.time()

    def body():
        shard.check_peers(w, 'round 1')                     # everybody healthy: passes
        recs = shard.gather_objects([{'rank': r, 'chunk': [r * 10, r * 10 + 5]}] if r == 1 else [], w)
        assert [len(x) for x in recs] == [0, 1] and recs[1][0]['rank'] == 1
        if r == 1:
            raise ValueError('rank 1 broke before the gather')
        shard.check_peers(w, 'the gather of the NMSE logs')
        shard.gather_trajectory_logs(torch.zeros(2, 3), 6, r, w)      # never reached on the healthy rank either
        return 'gathered'

    try:
        out = shard.run_guarded(w, body)
    except shard.PeerFailure as e:
        out = 'peer failure: %s' % e
    except ValueError as e:
        out = 'own failure: %s' % e
    q.put((rank, out, time.time() - t0, dist.is_initialized()))


def test_a_failing_rank_stops_the_others_at_the_agreement_point():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29100 + os.getpid() % 500
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert res[0][1].startswith('peer failure') and 'the gather of the NMSE logs' in res[0][1]
    assert res[1][1].startswith('own failure')
    assert max(r[2] for r in res) < 60                       # seconds, not the 300 s collective timeout
    assert not res[0][3] and not res[1][3]                   # both tore their process group down


def _late_failing_worker(rank, world, port, q):
    """rank 0 fails AFTER the last collective (say, while it writes its result file).  Its failure report must not be a
    collective nobody matches: it returns at once, and the healthy rank -- which has no agreement point left -- finishes normally."""
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SBC_DIST_TIMEOUT_S='60')
    from score_based_channels_amd import shard
    r, w, _ = shard.init_distributed('gloo')
    t0 = time.time()

    def body():
        shard.check_peers(w, 'the gather of the NMSE logs')
        got = shard.gather_trajectory_logs(torch.full((2, 3), float(r)), 6, r, w)
        assert got.shape == (2, 6)
        if r == 0:
            raise OSError('rank 0 could not write results.pt')
        return 'done'

    try:
        out = shard.run_guarded(w, body)
    except OSError as e:
        out = 'own failure: %s' % e
    q.put((rank, out, time.time() - t0))


def test_a_rank_failing_after_the_last_collective_hangs_nobody():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 28600 + os.getpid() % 500
    procs = [ctx.Process(target=_late_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert res[0][1].startswith('own failure') and res[1][1] == 'done'
    assert max(r[2] for r in res) < 30


def _forced_single_worker(port, q):
    """``--force_dist``: ONE rank, but a real process group -- every agreement point and gather of shard.py goes
    through it instead of being short-circuited."""
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from score_based_channels_amd import shard
    r, w, _ = shard.init_distributed('gloo', force=True)
    assert (r, w) == (0, 1) and dist.is_initialized() and dist.get_world_size() == 1

    def body():
        assert shard.broadcast_int(41, 0) == 41
        shard.check_peers(w, 'the gather')
        assert shard.gather_objects([{'rank': 0}], w) == [[{'rank': 0}]]
        full = torch.arange(12, dtype=torch.float32).view(3, 4)
        return bool(torch.equal(shard.gather_trajectory_logs(full.clone(), 4, r, w), full))
    ok = shard.run_guarded(w, body)
    q.put((ok, dist.is_initialized()))


def test_forced_one_rank_group_routes_everything_through_the_collectives():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_forced_single_worker, args=(28100 + os.getpid() % 500, q))
    p.start()
    ok, still = q.get(timeout=120)
    p.join(60)
    assert ok and not still
