"""Sharding of independent annealed-Langevin trajectories over the GPUs of one node.

The reference is single-GPU (``CUDA_VISIBLE_DEVICES``, test_score.py:29-30).  Trajectories -- (channel, SNR
point, (alpha, beta) cell, test profile) combinations -- never interact, so each rank runs a
contiguous block of the flattened trajectory list with a full weight replica, and the only exchange is one
``all_gather`` of the NMSE logs at the end (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU
box, ``gloo`` in the CPU tests).  Noise streams are keyed by global trajectory id, so results do not depend on
the world size.
"""
import os

import numpy as np


def dist_info():
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


# ``--force_dist`` of the CLIs: a ONE-rank process group is created and every agreement point / gather of this module goes through it
# (world = 1 normally short-circuits them) -- the RCCL path end to end on a one-GPU box
_force = [False]


def _single(world):
    """Is this a one-rank run whose collectives are skipped?"""
    return world <= 1 and not _force[0]


def init_distributed(backend=None, force=False):
    """Initialise ``torch.distributed`` when launched with WORLD_SIZE > 1 (or ``force``: also for one rank); returns
    (rank, world, local_rank)."""
    rank, world, local = dist_info()
    if force and world == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        _force[0] = True
    if world > 1 or force:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                # SBC_DIST_BACKEND=gloo: more ranks than GPUs (smoke runs and tests that share one device; RCCL refuses that)
                backend = os.environ.get('SBC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
            kw = {'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}
            dist.init_process_group(backend, **kw)
            _agreement[0] = 0                          # a fresh group, a fresh store: every rank starts the sequence of agreement points anew
        if torch.cuda.is_available() and local >= torch.cuda.device_count():
            if os.environ.get('SBC_DIST_BACKEND') != 'gloo':
                raise RuntimeError('LOCAL_RANK %d but only %d visible device(s): launch one rank per GPU (or set '
                                   'SBC_DIST_BACKEND=gloo to share a device in a smoke test)' % (local, torch.cuda.device_count()))
            local %= torch.cuda.device_count()
    return rank, world, local


class PeerFailure(RuntimeError):
    """Another rank reported a failure at an agreement point (``check_peers``): this rank stops too instead of waiting in the next
    collective until the process-group timeout."""


# Agreement points without a collective: the process group's key-value store (the TCPStore every torchrun job has)
# carries one arrival counter per agreement point and one failure key.  A failing rank only WRITES -- it never waits for anybody,
# wherever it failed (before the first agreement point, between the last one and the gathers, while rank 0 writes its result file)
# -- and a healthy rank polls: everybody arrived -> go on; failure key present -> PeerFailure.  No all-reduce that a dead peer
# leaves unmatched, nothing for the RCCL watchdog to time out on.
_POLL_S = 0.005
_agreement = [0]          # index of this rank's next agreement point (every rank walks the same sequence)


def _store():
    from torch.distributed import distributed_c10d as c10d
    return c10d._get_default_store()


def _peer_timeout_s():
    return float(os.environ.get('SBC_DIST_TIMEOUT_S', '300'))


def check_peers(world, where=''):
    """Agreement point in front of every collective of the CLIs.  A rank that failed since the last agreement point does not arrive
    here -- its handler (``report_failure``) sets the failure key instead -- and every healthy rank raises ``PeerFailure`` rather than
    entering a gather the failed rank will never join.  A peer that neither arrives nor reports within ``SBC_DIST_TIMEOUT_S``
    (default 300 s) counts as failed."""
    if _single(world):
        return
    import time
    st = _store()
    k = _agreement[0]
    _agreement[0] += 1
    key = 'sbc/arrived/%d' % k
    st.add(key, 1)
    deadline = time.monotonic() + _peer_timeout_s()
    while True:
        if st.check(['sbc/failed']):
            raise PeerFailure('another rank failed before %s; stopping this rank too' % (where or 'the next collective'))
        if int(st.add(key, 0)) >= world:
            return
        if time.monotonic() > deadline:
            raise PeerFailure('a peer did not reach the agreement point before %s within %.0f s' % (where or 'the next collective', _peer_timeout_s()))
        time.sleep(_POLL_S)


def report_failure(world):
    """The failing rank's half of ``check_peers``: sets the failure key its peers poll at their next agreement point.  Never blocks,
    so it is safe wherever the failure happened (also after the last collective).  Call once, from the handler of whatever
    exception ended this rank's work; never for a ``PeerFailure``."""
    if not _single(world):
        try:
            _store().set('sbc/failed', '1')
        except Exception:                                  # the store itself is gone: nothing more to tell anybody
            pass


def run_guarded(world, body):
    """``body()`` with the failure protocol around it: on an exception of THIS rank tell the peers (``report_failure``), on a
    ``PeerFailure`` just leave; either way the process group is torn down and the exception propagates (non-zero exit on every
    rank within seconds, not after the collective timeout)."""
    try:
        return body()
    except PeerFailure:
        raise
    except BaseException:
        report_failure(world)
        raise
    finally:
        if not _single(world):
            import torch.distributed as dist
            if dist.is_initialized():
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass


def gather_objects(obj, world):
    """Every rank's picklable ``obj`` as a list indexed by rank (``all_gather_object``; one rank: ``[obj]``)."""
    if _single(world):
        return [obj]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def collective_device(t):
    """Where a collective's payload must live: on the tensor's device for RCCL, in host memory for a gloo group."""
    import torch.distributed as dist
    return t.device if dist.get_backend() == 'nccl' else 'cpu'


def broadcast_int(value, src=0, device=None):
    """Rank ``src``'s integer on every rank (seeds)."""
    import torch
    import torch.distributed as dist
    check_peers(dist.get_world_size(), 'the seed broadcast')
    dev = device if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src)
    return int(t.item())


def block_bounds(n_items, world):
    """Boundaries of ``world`` contiguous, near-equal blocks of ``range(n_items)``: rank r owns
    ``[b[r], b[r+1])``; the first ``n_items % world`` ranks get one extra item."""
    q, r = divmod(int(n_items), int(world))
    sizes = np.array([q + (1 if i < r else 0) for i in range(world)], np.int64)
    return np.concatenate(([0], np.cumsum(sizes)))


def my_block(n_items, rank, world):
    b = block_bounds(n_items, world)
    return int(b[rank]), int(b[rank + 1])


def gather_trajectory_logs(local_log, n_items, rank, world):
    """All-gather per-rank NMSE logs ``[n_steps, T_local]`` into the full ``[n_steps, n_items]`` log (every rank
    gets it).  Blocks are padded to equal width because the collective needs equal shapes."""
    import torch
    if _single(world):
        return local_log
    import torch.distributed as dist
    b = block_bounds(n_items, world)
    width = int(np.max(np.diff(b)))
    n_steps = local_log.shape[0]
    pad = torch.zeros(n_steps, width, dtype=local_log.dtype, device=collective_device(local_log))
    pad[:, :local_log.shape[1]] = local_log
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([parts[r][:, :int(b[r + 1] - b[r])] for r in range(world)], dim=1).to(local_log.device)


def all_reduce_sum_(t, world):
    """In-place SUM all-reduce of a device tensor (the flat gradient buffer of data-parallel training, train.py).  RCCL
    reduces it where it lies; a gloo group (CPU tests, more ranks than GPUs) takes it through host memory."""
    if _single(world):
        return t
    import torch.distributed as dist
    if dist.get_backend() == 'nccl' or not t.is_cuda:
        dist.all_reduce(t)
    else:
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    return t
