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
