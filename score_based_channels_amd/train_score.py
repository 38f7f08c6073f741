#!/usr/bin/env python3
"""Train the score network by denoising score matching -- counterpart of ``src/score_based_channels/train_score.py``
with the whole optimiser step (perturbation, forward, loss, backward, Adam, EMA) on the MI355X HIP path (``train.py``).

    python -m score_based_channels_amd.train_score --train CDL-C

Same arguments (``--gpu``, ``--train``, train_score.py:20-23), configuration (:34-67,98-115), data files
(``./data/<profile>_Nt64_Nr16_ULA0.50_seed{1234,4321}.mat``), loop (:145-207: running loss, validation loss of the EMA copy
every 100 steps) and output file ``./models/score/<train>/final_model.pt`` with the keys ``model_state, optim_state,
config, train_loss, val_loss`` (:211-216; read by ``test_score`` here; ``model_state`` and ``optim_state`` load into the
reference's ``NCSNv2Deepest`` / ``torch.optim.Adam`` unchanged, ``config`` is a ``dotmap.DotMap`` only where dotmap is installed -- a
plain dict otherwise, which the reference's scripts would have to wrap).  Additions, all optional:
``--seed`` (the reference never seeds), ``--n_epochs / --max_steps / --batch_size / --val_every`` (short runs),
``--synthetic`` (stand-in for the undistributed data), ``--init`` (start from a checkpoint), ``--graph`` (replay the step
as a hipGraph).
"""
import argparse
import os

import numpy as np


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpu', type=int, default=0)
    p.add_argument('--train', type=str, default='CDL-C')
    # additions of this build
    p.add_argument('--seed', type=int, default=None, help='seed of the initial weights, the batch order, labels and noise')
    p.add_argument('--n_epochs', type=int, default=400, help='train_score.py:54')
    p.add_argument('--max_steps', type=int, default=None, help='stop after this many optimiser steps')
    p.add_argument('--batch_size', type=int, default=32, help='train_score.py:52')
    p.add_argument('--val_every', type=int, default=100, help='train_score.py:169')
    p.add_argument('--synthetic', action='store_true', help='generate CDL-like channels instead of reading ./data')
    p.add_argument('--num_synthetic', type=int, default=200)
    p.add_argument('--init', type=str, default=None, help='checkpoint to start from instead of a fresh initialisation')
    p.add_argument('--graph', action='store_true', help='replay the optimiser step as a hipGraph')
    p.add_argument('--out_dir', type=str, default=None, help='default ./models/score/<train> (train_score.py:140)')
    return p.parse_args(argv)


def training_config(channel):
    """train_score.py:34-67,98-115."""
    from .config import default_config
    c = default_config(channel)
    c.optim.weight_decay = 0.000
    c.optim.optimizer = 'Adam'
    c.optim.lr = 0.0001
    c.optim.beta1 = 0.9
    c.optim.amsgrad = False
    c.optim.eps = 0.001
    c.training.batch_size = 32
    c.training.num_workers = 0
    c.training.n_epochs = 400
    c.training.anneal_power = 2
    c.training.log_all_sigmas = False
    # inference step size according to [Song '20] (train_score.py:103-115)
    m = c.model
    candidate_steps = np.logspace(-13, -8, 1000)
    gamma_rate = 1 / m.sigma_rate
    s2 = m.sigma_end ** 2
    crit = np.zeros(len(candidate_steps))
    for i, step in enumerate(candidate_steps):
        inner = 2 * step / (s2 - s2 * (1 - step / s2) ** 2)
        crit[i] = (1 - step / s2) ** (2 * m.num_classes) * (gamma_rate ** 2 - inner) + inner
    m.step_size = float(candidate_steps[np.argmin(np.abs(crit - 1.))])
    return c


def herm_real_view(dataset):
    """``sample['H_herm']`` of every item at once: float32 ``[N, 2, Nt, Nr]`` (loaders.py:69,88-91)."""
    hn = (dataset.channels - dataset.mean) / dataset.std
    h = np.conj(np.transpose(hn, (0, 2, 1)))
    return np.stack((h.real, h.imag), axis=1).astype(np.float32)


def fresh_state_dict(config, seed):
    """Initial parameters with the distributions the reference modules start from: nn.Conv2d defaults, InstanceNorm++
    alpha, gamma ~ N(1, 0.02), beta = 0 (normalization.py:154-161)."""
    from .weights import seeded_state_dict
    sd = seeded_state_dict(config, seed)
    for k in sd:
        if k.endswith('.beta'):
            sd[k] = np.zeros_like(sd[k])
    return sd


def torch_optim_state(net):
    """The optimiser state in ``torch.optim.Adam.state_dict()`` form (parameters in ``named_parameters`` order)."""
    import torch
    st = net.optimizer_state()
    names = [n for n, _ in net.spec]
    state = {i: {'step': torch.tensor(float(st['step'])), 'exp_avg': torch.from_numpy(st['exp_avg'][n]),
                 'exp_avg_sq': torch.from_numpy(st['exp_avg_sq'][n])} for i, n in enumerate(names)}
    group = {'lr': net.lr, 'betas': (net.beta1, 0.999), 'eps': net.eps, 'weight_decay': 0.0, 'amsgrad': False,
             'params': list(range(len(names)))}
    return {'state': state if st['step'] else {}, 'param_groups': [group]}


def main(argv=None):
    args = parse_args(argv)
    import torch
    from .checkpoint import load_checkpoint
    from .loaders import Channels
    from .train import TrainNet

    from . import shard
    rank, world, local = shard.init_distributed(os.environ.get('SBC_DIST_BACKEND'))
    n_dev = max(1, torch.cuda.device_count())
    device = 'cuda:%d' % ((local % n_dev) if world > 1 else args.gpu)
    torch.cuda.set_device(device)
    seed = int.from_bytes(os.urandom(4), 'little') if args.seed is None else args.seed
    if world > 1:                                         # every rank must use rank 0's seed
        import torch.distributed as dist
        t = torch.tensor([seed], dtype=torch.int64)
        t = t.to(device) if dist.get_backend() == 'nccl' else t
        dist.broadcast(t, 0)
        seed = int(t.item())
    np.random.seed(seed % (2 ** 32))                      # pilots of the datasets (loaders.py:52-55)
    gen = torch.Generator().manual_seed(seed)             # batch order and noise levels

    config = training_config(args.train)
    config.training.batch_size, config.training.n_epochs = args.batch_size, args.n_epochs
    train_seed, val_seed = 1234, 4321
    kw = dict(synthetic=args.synthetic, num_synthetic=args.num_synthetic)
    dataset = Channels(train_seed, config, norm=config.data.norm_channels, **kw)
    val_dataset = Channels(val_seed, config, norm=[dataset.mean, dataset.std], **kw)
    train_x, val_x = herm_real_view(dataset), herm_real_view(val_dataset)
    # data parallel (torch.distributed.run): --batch_size stays the GLOBAL batch of train_score.py:52, every rank takes a
    # contiguous share of it; gradients are summed by one all-reduce per step (RCCL), so the run is the same optimisation
    # as on one GPU
    Bg = args.batch_size
    if Bg % world:
        raise ValueError('batch size %d is not a multiple of the world size %d' % (Bg, world))
    B = Bg // world
    if len(train_x) < Bg:
        raise ValueError('%d training channels < batch size %d' % (len(train_x), Bg))

    net = TrainNet(config, batch=B, device=device, seed=seed, rank=rank, world=world)
    if args.init:
        net.load_state_dict(load_checkpoint(args.init)['model_state'])
    else:
        net.load_state_dict(fresh_state_dict(config, seed))
    L = config.model.num_classes

    def global_mean(per):
        """Mean of the per-sample losses over all ranks (every rank has the same count)."""
        t = per.double().sum().reshape(1)
        if world > 1:
            import torch.distributed as dist
            t = t if dist.get_backend() == 'nccl' else t.cpu()
            dist.all_reduce(t)
        return float(t.item()) / (per.numel() * world)

    def val_loss_ema():
        """train_score.py:172-185: DSM loss of the EMA copy on the validation channels with fresh labels and fresh noise
        (``TrainNet.loss`` draws from its own Philox stream, advanced per call).  The reference evaluates ONE fixed first
        validation batch (``val_sample``, :120-121,172); here every full chunk of the validation set is evaluated and the
        mean over all of them is logged: the same estimator with less variance, not the same number."""
        tot, cnt = 0.0, 0
        for s in range(0, len(val_x) - Bg + 1, Bg):
            labels = torch.randint(0, L, (Bg,), generator=gen)[rank * B:(rank + 1) * B]
            per = net.loss(val_x[s + rank * B:s + (rank + 1) * B], labels, ema=bool(config.model.ema))
            tot, cnt = tot + global_mean(per), cnt + 1
        return tot / max(cnt, 1)

    train_loss, val_loss = [], []
    step, running = 0, 0.0
    done = False
    steps_per_epoch = len(train_x) // Bg                   # drop_last=True (:75)
    torch.cuda.synchronize(device)
    torch.cuda.set_stream(torch.cuda.Stream(device))       # hipGraph replay needs a stream of its own
    for epoch in range(config.training.n_epochs):
        order = torch.randperm(len(train_x), generator=gen).numpy()          # shuffle=True (:74)
        for i in range(steps_per_epoch):
            step += 1
            mine = slice(i * Bg + rank * B, i * Bg + (rank + 1) * B)       # this rank's share of the global batch
            batch = train_x[order[mine]]
            labels = torch.randint(0, L, (Bg,), generator=gen)[rank * B:(rank + 1) * B]      # dsm.py:9-12
            loss = global_mean(net.step(batch, labels, use_graph=args.graph))       # :151-167 (.item() as :156-159)
            running = loss if step == 1 else 0.99 * running + 0.01 * loss
            train_loss.append(loss)
            if step % args.val_every == 0:
                val_loss.append([val_loss_ema()])
                if rank == 0:
                    print('Epoch %d, Step %d, Train Loss (EMA) %.3f, Val. Loss %.3f' % (epoch, step, running, val_loss[-1][0]))
            if args.max_steps is not None and step >= args.max_steps:
                done = True
                break
        if done:
            break

    torch.cuda.synchronize(device)
    torch.cuda.set_stream(torch.cuda.default_stream(device))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if rank != 0:                                      # replicas are identical: rank 0 writes the checkpoint
            dist.destroy_process_group()
            return train_loss, val_loss
    out_dir = args.out_dir or './models/score/%s' % args.train
    os.makedirs(out_dir, exist_ok=True)
    config.log_path = out_dir
    to_t = lambda sd: {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}     # noqa: E731
    # `config`: the reference pickles its dotmap.DotMap and reads it back with attribute access (test_score.py:56,59).  With
    # dotmap installed the same object type is written; without it (this image) a plain nested dict, which this package's
    # loader re-wraps (checkpoint.load_checkpoint) and the reference's scripts would have to wrap in DotMap(...) themselves.
    try:
        from dotmap import DotMap
        cfg_obj = DotMap(config.toDict())
    except ImportError:
        cfg_obj = config.toDict()
    torch.save({'model_state': to_t(net.state_dict()), 'optim_state': torch_optim_state(net), 'config': cfg_obj,
                'train_loss': train_loss, 'val_loss': val_loss,
                'ema_state': to_t(net.ema_state_dict()), 'seed': seed},           # two additions of this build
               os.path.join(out_dir, 'final_model.pt'))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return train_loss, val_loss


if __name__ == '__main__':
    main()
