"""Denoising-score-matching training step of the score network on the HIP kernels.

Reference: ``anneal_dsm_score_estimation`` (``ncsnv2/losses/dsm.py:6-32``) inside the loop of
``train_score.py:145-173`` -- perturb the batch with per-sample noise levels, evaluate the network, weighted squared
error against ``-noise / sigma^2``, ``loss.backward()``, ``optimizer.step()`` (Adam, ``losses/__init__.py:3-7``),
``ema_helper.update`` (``models/ema.py:17-22``).

``TrainNet`` holds every parameter of ``NCSNv2Deepest`` in ONE flat float32 device buffer (torch layouts, names as in
``state_dict()``), a gradient buffer of the same layout and the optimiser state, and turns one training step into one
``sbc_plan``:

    pack weights (SBC_OP_PACK_WEIGHT, forward + adjoint forms)  ->  SBC_OP_DSM_PERTURB  ->  the forward records of
    plan.build_score_plan(share_slots=False)  ->  SBC_OP_DSM_LOSS  ->  the reverse records built here, one group per
    forward record in reverse order  ->  SBC_OP_ADAM_EMA  ->  SBC_OP_STEP_INC

torch is device memory and streams only; no torch operator (and no autograd) computes anything here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import plan as P
from .weights import get_sigmas, state_dict_spec


def _ptr(t, offset_elems=0):
    return C.c_void_p(t.data_ptr() + 4 * offset_elems)


class TrainNet:
    """``TrainNet(config, batch, nt, nr)``: a trainable NCSNv2Deepest for a fixed batch shape.

    ``load_state_dict`` / ``state_dict`` / ``ema_state_dict`` speak the reference checkpoint grammar
    (``train_score.py:211-216``).  ``step(samples, labels=None, noise=None)`` runs one optimiser step and returns the
    device tensor of per-sample losses (their mean is ``anneal_dsm_score_estimation``'s return value);
    ``loss(samples, labels, noise, ema=False)`` evaluates the loss only (validation, ``train_score.py:178-185``);
    ``backward(samples, labels, noise)`` stops after the gradients (parity tests)."""

    def __init__(self, config, batch, nt=None, nr=None, device=None, seed=0, rank=0, world=1):
        m, d, o, tr = config.model, config.data, config.optim, config.training
        if str(m.normalization) != 'InstanceNorm++' or str(m.nonlinearity).lower() != 'elu':
            raise NotImplementedError('the HIP path implements InstanceNorm++ / ELU (train_score.py:39-40)')
        if int(m.ngf) != 32 or int(d.channels) != 2:
            raise NotImplementedError('kernels are instantiated for ngf = 32, 2 input channels')
        if o and str(o.optimizer or 'Adam') != 'Adam':
            raise NotImplementedError('only Adam (train_score.py:44) is implemented')
        self.config = config
        self.device = torch.device(device if device is not None else 'cuda:0')
        self.B = int(batch)
        self.nt = int(nt if nt is not None else d.image_size[1])
        self.nr = int(nr if nr is not None else d.image_size[0])
        self.ngf, self.channels, self.num_classes = int(m.ngf), int(d.channels), int(m.num_classes)
        self.lr = float(o.lr) if o and o.lr else 1e-4
        self.beta1 = float(o.beta1) if o and o.beta1 else 0.9
        self.eps = float(o.eps) if o and o.eps else 1e-3
        self.ema_mu = float(m.ema_rate) if m.ema and m.ema_rate else -1.0
        self.anneal_power = float(tr.anneal_power) if tr and tr.anneal_power else 2.0
        self.seed = int(seed)
        # data parallel: `batch` samples per rank; the loss gradient is scaled by 1 / world so that the SUM all-reduce of the
        # flat gradient buffer is the gradient of the mean over the global batch; Philox streams are keyed by the GLOBAL
        # sample index, so a step does not depend on how the batch is split
        self.rank, self.world = int(rank), int(world)
        self.plan = P.build_score_plan(self.ngf, self.nt, self.nr, self.channels, share_slots=False)
        self._layout()
        self._alloc()
        self._plans = {}
        self._last_stream = None

    # --- parameter layout -------------------------------------------------------------------------------------------
    def _layout(self):
        """Offsets (float32 elements, 16-byte aligned) of every parameter in the flat buffer; the alpha | gamma | beta of a
        norm are adjacent so that the statistics / backward kernels see one [3][C] array."""
        self.spec = [(n, s) for n, s in state_dict_spec(self.ngf, self.channels, self.num_classes) if n != 'sigmas']
        self.off, cur = {}, 0
        for name, shape in self.spec:
            if name.endswith('.gamma') or name.endswith('.beta'):
                continue
            cur += (-cur) % 4
            if name.endswith('.alpha'):
                pre = name[:-len('alpha')]
                c = shape[0]
                self.off[pre + 'alpha'], self.off[pre + 'gamma'], self.off[pre + 'beta'] = cur, cur + c, cur + 2 * c
                cur += 3 * c
            else:
                self.off[name] = cur
                cur += int(np.prod(shape))
        cur += (-cur) % 4
        self.n_params = cur
        self.shape = dict(self.spec)

    def _alloc(self):
        dev, B = self.device, self.B
        f32 = dict(dtype=torch.float32, device=dev)
        self.params = torch.zeros(self.n_params, **f32)
        self.grads = torch.zeros(self.n_params, **f32)
        self.state = torch.zeros(3, self.n_params, **f32)            # exp_avg | exp_avg_sq | EMA shadow
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.eval_count = torch.zeros(1, dtype=torch.int32, device=dev)     # forward-only (validation) calls made so far
        self.sigmas = torch.from_numpy(get_sigmas(self.config)).to(dev)
        pl = self.plan
        self.slots = [torch.zeros(B * e, **f32) for e in pl.slot_elems]
        n = self.nt * self.nr * self.channels
        self.samples = torch.zeros(B, n, **f32)
        self.noise = torch.zeros(B, n, **f32)                         # sigma_b * z (kept for the loss)
        self.replay = torch.zeros(B, n, **f32)                        # standard-normal draws to replay (parity runs)
        self.labels = torch.zeros(B, dtype=torch.int64, device=dev)
        self.sample_id = (torch.arange(B, dtype=torch.int64) + self.rank * B).to(dev)
        self.loss_per_sample = torch.zeros(B, **f32)
        self._make_pack_table()
        big = max(t.elems for t in pl.tensors)
        self.tmp_a = torch.zeros(B * big, **f32)
        self.inorm_aux = torch.zeros(B * 6 * 4 * self.ngf, **f32)
        self.pool_aux = torch.zeros(B * big, dtype=torch.uint8, device=dev)
        lib = _lib.lib()
        scr = 0
        for op in pl.ops:
            if op.kind in (P.CONV, P.BEGIN_CONV, P.END_CONV):
                scr = max(scr, int(lib.sbc_wgrad_scratch_floats(B, op.src.h, op.src.w, op.src.c, op.dst.c, op.ksize)))
        self.scratch = torch.zeros(scr, **f32)                        # weight-gradient partials (side stream, in order)
        self.scratch_main = torch.zeros(int(lib.sbc_wgrad_scratch_floats(B, self.nt, self.nr, self.ngf, self.channels, 3)), **f32)
        self.gbuf, self.pool_grad = {}, {}

    # --- checkpoint grammar -----------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        sd = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in state_dict.items()}
        flat = np.zeros(self.n_params, np.float32)
        for name, shape in self.spec:
            if name not in sd:
                if strict:
                    raise KeyError('missing tensor %r' % name)
                continue
            a = np.asarray(sd[name], np.float32)
            if tuple(a.shape) != tuple(shape):
                raise ValueError('%s: shape %s, expected %s' % (name, a.shape, shape))
            flat[self.off[name]:self.off[name] + a.size] = a.ravel()
        if 'sigmas' in sd:
            self.sigmas.copy_(torch.from_numpy(np.asarray(sd['sigmas'], np.float32)))
        self.params.copy_(torch.from_numpy(flat))
        self.state.zero_()
        self.state[2].copy_(self.params)                               # EMAHelper.register: shadow = param.clone()
        self.step_count.zero_()
        # these copies ran on the caller's current stream: the next _run (possibly on another stream) must wait for them
        if self.params.is_cuda:                                        # (a CPU-resident TrainNet only exists in host-logic tests)
            self._last_stream = torch.cuda.current_stream(self.device)
        return self

    def _export(self, flat):
        flat = flat.detach().cpu().numpy()
        out = {'sigmas': self.sigmas.cpu().numpy().copy()}
        for name, shape in self.spec:
            out[name] = flat[self.off[name]:self.off[name] + int(np.prod(shape))].reshape(shape).copy()
        return out

    def state_dict(self):
        return self._export(self.params)

    def ema_state_dict(self):
        return self._export(self.state[2])

    def grad_dict(self):
        g = self._export(self.grads)
        g.pop('sigmas')
        return g

    def optimizer_state(self):
        return {'step': int(self.step_count.item()), 'exp_avg': self._export(self.state[0]),
                'exp_avg_sq': self._export(self.state[1])}

    # --- op records ---------------------------------------------------------------------------------------------------
    def _par(self, base, name):
        return _ptr(base, self.off[name])

    def _grad_of(self, t):
        if id(t) not in self.gbuf:
            self.gbuf[id(t)] = torch.zeros(self.B * t.elems, dtype=torch.float32, device=self.device)
        return self.gbuf[id(t)]

    def _pool_grad(self, op):
        if op.name not in self.pool_grad:
            self.pool_grad[op.name] = torch.zeros(self.B * op.src.h * op.src.w * op.dst.c, dtype=torch.float32, device=self.device)
        return self.pool_grad[op.name]

    def _make_pack_table(self):
        """Device table of SBC_OP_PACK_WEIGHT's batched form: per convolution weight the direct split-bf16 form and its
        adjoint, and for undilated 3x3 layers also the Winograd F(2x2,3x3) form and its adjoint (what conv_wx3 consumes)."""
        rows, cur, self._pack_off = [], 0, {}
        for op in self.plan.ops:
            if op.kind != P.CONV:
                continue
            cout, cin, k, _ = self.shape[op.weight]
            forms = [('fwd', k * k, 0), ('adj', k * k, 1)]
            if k == 3 and op.dil == 1:
                forms += [('wfwd', 16, 0), ('wadj', 16, 1)]
            for form, taps, adj in forms:
                self._pack_off[(op.weight, form)] = cur
                rows.append((self.off[op.weight], cur, cout, cin, taps, adj))
                cur += 3 * cout * cin * taps                               # uint16 elements
        self._pack_table = torch.tensor(rows, dtype=torch.int32, device=self.device)
        self._pack_buf = torch.zeros(cur // 2, dtype=torch.float32, device=self.device)
        self._pack_max = (max(r[2] for r in rows), max(r[3] for r in rows))

    def _pack_ops(self, base):
        """ONE launch that re-packs every convolution weight (all forms) from the flat parameter buffer."""
        return [_lib.sbc_op(kind=P.PACK_WEIGHT, B=self._pack_table.shape[0], cout=self._pack_max[0], cin=self._pack_max[1],
                            ksize=3, in_=_ptr(base), out=_ptr(self._pack_buf), aux=C.c_void_p(self._pack_table.data_ptr()))]

    def _packed(self, wkey, form):
        """Device pointer of a packed form ('fwd', 'adj', 'wfwd', 'wadj'), or None if the layer has no such form."""
        off = self._pack_off.get((wkey, form))
        return None if off is None else C.c_void_p(self._pack_buf.data_ptr() + 2 * off)

    def _forward_ops(self, base, keep):
        """The forward records of plan.py bound to private activation slots; conv weights come from the packed copies."""
        B, sl = self.B, self.slots
        ext = _lib.sbc_endconv(sigmas=_ptr(self.sigmas), labels=_ptr(self.labels))
        keep.append(ext)
        ops = []
        for op in self.plan.ops:
            o = _lib.sbc_op(kind=op.kind, flags=op.flags, B=B, H=op.src.h, W=op.src.w, cin=op.src.c, cout=op.dst.c,
                            ksize=op.ksize, dil=op.dil, tag=op.tag, in_=_ptr(sl[op.src.slot]), out=_ptr(sl[op.dst.slot]))
            if op.kind == P.CONV:
                o.weight_split = self._packed(op.weight, 'fwd')
                o.weight_wino_split = self._packed(op.weight, 'wfwd')      # the inference kernels (conv_wx3) where they apply
            elif op.kind == P.INORM_STATS:
                o.weight = self._par(base, op.weight + '.alpha')
            elif op.weight is not None:
                o.weight = self._par(base, op.weight)
            if op.bias is not None:
                o.bias = self._par(base, op.bias)
            for f in ('stats', 'res1', 'res2'):
                t = getattr(op, f)
                if t is not None:
                    setattr(o, f, _ptr(sl[t.slot]))
            if op.up is not None:
                o.up, o.up_h, o.up_w = _ptr(sl[op.up.slot]), op.up.h, op.up.w
            if op.kind == P.END_CONV:
                o.ext = C.cast(C.pointer(ext), C.c_void_p)
            ops.append(o)
        return ops

    def _backward_ops(self, keep):
        """Reverse-mode records: walk the forward list backwards.  Per tensor the gradient is in one of three states:
        nothing yet; ``alias`` -- exactly one plain term so far, which is simply another tensor's finished gradient buffer
        (e.g. the ``+ x`` of a residual connection): nothing is copied until a second term arrives; ``done`` -- materialised
        in the tensor's own buffer, further terms are added (SBC_BWD_ACCUM, or the adjoint convolution's own epilogue)."""
        B, sl, pl = self.B, self.slots, self.plan
        base, gr = self.params, self.grads
        done, alias = set(), {}
        ops = []
        ext = _lib.sbc_endconv(sigmas=_ptr(self.sigmas), labels=_ptr(self.labels))
        keep.append(ext)
        done.add(id(pl.out))                                   # written by SBC_OP_DSM_LOSS

        def grad_add(t, src_buf, flags=0, x=None):
            ops.append(_lib.sbc_op(kind=P.GRAD_ADD, flags=flags, B=B, H=t.h, W=t.w, cin=t.c,
                                   in_=_ptr(sl[x.slot]) if x is not None else None, grad=_ptr(src_buf),
                                   out=_ptr(self._grad_of(t))))

        def materialise(t):
            """Make the tensor's own buffer hold what has been collected so far; returns SBC_BWD_ACCUM if there is anything."""
            if id(t) in alias:
                grad_add(t, alias.pop(id(t)))
                done.add(id(t))
            return P.BWD_ACCUM if id(t) in done else 0

        def acc(t):
            """Flag for a kernel that writes (first term) or adds to (later terms) the tensor's own buffer."""
            f = materialise(t)
            done.add(id(t))
            return f

        def add_plain(t, buf):
            """g(t) += buf, where buf is a finished gradient buffer that nobody writes again."""
            if id(t) not in done and id(t) not in alias:
                alias[id(t)] = buf
            else:
                grad_add(t, buf, acc(t))

        def current(t):
            """The buffer that holds the complete gradient of t (call when every consumer of t has been reversed)."""
            if id(t) in alias:
                return alias[id(t)]
            if id(t) not in done:
                raise RuntimeError('no gradient reaches %s' % t.name)
            return self._grad_of(t)

        def norm_params(stats_tensor):
            """state_dict prefix of the norm that produced a statistics tensor."""
            for op in pl.ops:
                if op.kind == P.INORM_STATS and op.dst is stats_tensor:
                    return op.weight
            raise KeyError(stats_tensor.name)

        def inorm_bwd(src, stats, grad_buf):
            pre = norm_params(stats)
            ops.append(_lib.sbc_op(kind=P.INORM_BWD, flags=P.PRO_ELU | acc(src), B=B, H=src.h, W=src.w, cin=src.c,
                                   in_=_ptr(sl[src.slot]), stats=_ptr(sl[stats.slot]), weight=self._par(base, pre + '.alpha'),
                                   grad=_ptr(grad_buf), out=_ptr(self._grad_of(src)), aux=_ptr(self.inorm_aux),
                                   wgrad=self._par(gr, pre + '.alpha')))

        for op in reversed(pl.ops):
            if op.kind == P.INORM_STATS:
                continue                                    # reversed together with the consumer of the statistics
            dy = current(op.dst)
            src = op.src
            if op.kind == P.END_CONV:
                ops.append(_lib.sbc_op(kind=P.END_CONV_BWD, B=B, H=src.h, W=src.w, cin=src.c, cout=op.dst.c, ksize=3, dil=1,
                                       in_=_ptr(sl[src.slot]), stats=_ptr(sl[op.stats.slot]), weight=self._par(base, op.weight),
                                       grad=_ptr(dy), out=_ptr(self.tmp_a), aux=_ptr(self.scratch_main),
                                       wgrad=self._par(gr, op.weight), bgrad=self._par(gr, op.bias),
                                       ext=C.cast(C.pointer(ext), C.c_void_p)))
                inorm_bwd(src, op.stats, self.tmp_a)
            elif op.kind == P.MAXPOOL5:
                ops.append(_lib.sbc_op(kind=P.MAXPOOL5_BWD, flags=(op.flags & P.PRO_ELU) | acc(src), B=B, H=src.h, W=src.w,
                                       cin=src.c, in_=_ptr(sl[src.slot]), grad=_ptr(dy), out=_ptr(self._grad_of(src)),
                                       aux=C.c_void_p(self.pool_aux.data_ptr())))
            elif op.kind == P.BEGIN_CONV:
                ops.append(_lib.sbc_op(kind=P.BEGIN_CONV_BWD, flags=P.OP_SIDE, B=B, H=src.h, W=src.w, cin=src.c, cout=op.dst.c,
                                       ksize=3, dil=1, in_=_ptr(sl[src.slot]), grad=_ptr(dy), aux=_ptr(self.scratch),
                                       wgrad=self._par(gr, op.weight), bgrad=self._par(gr, op.bias)))
            elif op.kind == P.CONV:
                dst = op.dst
                if op.res2 is not None:
                    add_plain(op.res2, dy)
                if op.res1 is not None:
                    if op.flags & P.EPI_RES1_ELU:
                        grad_add(op.res1, dy, P.PRO_ELU | acc(op.res1), x=op.res1)
                    else:
                        add_plain(op.res1, dy)
                if op.up is not None:
                    ops.append(_lib.sbc_op(kind=P.UPSAMPLE_BWD, flags=acc(op.up), B=B, H=dst.h, W=dst.w, cin=dst.c,
                                           up_h=op.up.h, up_w=op.up.w, grad=_ptr(dy), out=_ptr(self._grad_of(op.up))))
                dc = dy
                if op.flags & P.EPI_POOL:
                    dc = self._pool_grad(op)              # private buffer: the weight gradient reads it from the side stream
                    ops.append(_lib.sbc_op(kind=P.POOL_BWD, B=B, H=src.h, W=src.w, cin=dst.c, grad=_ptr(dy), out=_ptr(dc)))
                pro = op.flags & (P.PRO_NORM | P.PRO_ELU)
                # weight gradients are off the critical path (nothing before the optimiser reads them): side stream
                ops.append(_lib.sbc_op(kind=P.CONV_WGRAD, flags=pro | P.OP_SIDE, B=B, H=src.h, W=src.w, cin=src.c, cout=dst.c,
                                       ksize=op.ksize, dil=op.dil, in_=_ptr(sl[src.slot]),
                                       stats=_ptr(sl[op.stats.slot]) if op.stats is not None else None, grad=_ptr(dc),
                                       aux=_ptr(self.scratch), wgrad=self._par(gr, op.weight),
                                       bgrad=self._par(gr, op.bias) if op.bias is not None else None))
                # input gradient: the adjoint convolution dst.c -> src.c of dC, then back through the prologue.  Without a
                # norm in the prologue the convolution's own epilogue multiplies by ELU'(src) and adds what was collected
                # before (an aliased buffer, or the tensor's own buffer in place)
                adj = _lib.sbc_op(kind=P.CONV, B=B, H=src.h, W=src.w, cin=dst.c, cout=src.c, ksize=op.ksize, dil=op.dil,
                                  in_=_ptr(dc), weight_split=self._packed(op.weight, 'adj'),
                                  weight_wino_split=self._packed(op.weight, 'wadj'))
                if pro & P.PRO_NORM:
                    adj.out = _ptr(self.tmp_a)
                    ops.append(adj)
                    inorm_bwd(src, op.stats, self.tmp_a)
                else:
                    if pro & P.PRO_ELU:
                        adj.flags |= P.EPI_ELUGRAD
                        adj.res2 = _ptr(sl[src.slot])
                    if id(src) in alias:
                        adj.res1 = _ptr(alias.pop(id(src)))
                    elif id(src) in done:
                        adj.res1 = _ptr(self._grad_of(src))
                    adj.out = _ptr(self._grad_of(src))
                    done.add(id(src))
                    ops.append(adj)
            else:
                raise NotImplementedError('no reverse rule for op kind %d' % op.kind)
        return ops

    def _build(self, mode):
        """mode: 'step' (full optimiser step), 'backward' (gradients only), 'apply' (optimiser only), 'loss' / 'loss_ema'
        (forward only)."""
        keep = []
        base = self.state[2] if mode == 'loss_ema' else self.params
        pl = self.plan
        # Philox counter word 1 of the perturbation noise: the optimiser step for the training plans; forward-only (validation)
        # plans draw from a stream of their own -- offset 2^30 + the number of forward-only calls so far -- so that every
        # loss() call sees fresh noise (the reference draws a fresh randn per call, dsm.py:14) that no training step reuses
        forward_only = mode in ('loss', 'loss_ema')
        counter = self.eval_count if forward_only else self.step_count
        common = dict(sigmas=_ptr(self.sigmas), labels=_ptr(self.labels), sample_id=_ptr(self.sample_id), seed=self.seed,
                      offset=(1 << 30) if forward_only else 0, anneal_power=self.anneal_power, step=_ptr(counter),
                      grad_scale=1.0 / self.world)
        dsm = _lib.sbc_dsm(noise=None, **common)
        dsm_replay = _lib.sbc_dsm(noise=_ptr(self.replay), **common)
        keep += [dsm, dsm_replay]
        plans = {}
        for replay in (False, True):
            e = dsm_replay if replay else dsm
            ep = C.cast(C.pointer(e), C.c_void_p)
            want_grad = mode in ('step', 'backward')
            ops = []
            if mode != 'apply':                                  # 'apply' = optimiser only (after the gradient all-reduce)
                ops += self._pack_ops(base)
                ops.append(_lib.sbc_op(kind=P.DSM_PERTURB, B=self.B, H=self.nt, W=self.nr, cin=self.channels,
                                       in_=_ptr(self.samples), out=_ptr(self.slots[pl.x.slot]), aux=_ptr(self.noise), ext=ep))
                ops += self._forward_ops(base, keep)
                ops.append(_lib.sbc_op(kind=P.DSM_LOSS, B=self.B, H=self.nt, W=self.nr, cin=self.channels,
                                       in_=_ptr(self.slots[pl.out.slot]), grad=_ptr(self.noise), out=_ptr(self.loss_per_sample),
                                       aux=_ptr(self._grad_of(pl.out)) if want_grad else None, ext=ep))
            n_fwd = len(ops)
            if want_grad:
                ops += self._backward_ops(keep)
            if mode in ('step', 'apply'):
                adam = _lib.sbc_adam(n=self.n_params, lr=self.lr, beta1=self.beta1, beta2=0.999, eps=self.eps,
                                     ema_mu=self.ema_mu, step=_ptr(self.step_count))
                keep.append(adam)
                ops.append(_lib.sbc_op(kind=P.ADAM_EMA, flags=P.OP_JOIN, in_=_ptr(self.grads), out=_ptr(self.params), aux=_ptr(self.state),
                                       ext=C.cast(C.pointer(adam), C.c_void_p)))
                ops.append(_lib.sbc_op(kind=P.STEP_INC, out=C.c_void_p(self.step_count.data_ptr())))
            if forward_only:
                ops.append(_lib.sbc_op(kind=P.STEP_INC, out=C.c_void_p(self.eval_count.data_ptr())))
            # profiling tags (sbc_plan_profile): 100 + kind for the packing / forward part, 200 + kind for the rest
            for i, o in enumerate(ops):
                o.tag = (100 if i < n_fwd else 200) + o.kind
            plans[replay] = _lib.Plan(ops, keepalive=keep)
        return plans

    def _run(self, mode, samples, labels, noise, use_graph=False):
        if mode not in self._plans:
            self._plans[mode] = self._build(mode)
        cur = torch.cuda.current_stream(self.device)
        if self._last_stream is not None and self._last_stream != cur:
            cur.wait_stream(self._last_stream)                           # the buffers are shared between calls
        self._last_stream = cur
        x = torch.as_tensor(samples)
        if x.dim() == 4 and x.shape[1] == self.channels:                 # reference layout [B, 2, Nt, Nr]
            x = x.permute(0, 2, 3, 1)
        if tuple(x.shape[:1]) != (self.B,) or x.numel() != self.samples.numel():
            raise ValueError('expected %d samples of %dx%dx%d, got %s' % (self.B, self.nt, self.nr, self.channels, tuple(x.shape)))
        self.samples.copy_(x.to(self.device, torch.float32).reshape(self.B, -1))
        if labels is None:
            raise ValueError('labels must be given (draw them with torch.randint as dsm.py:9-12 does)')
        self.labels.copy_(torch.as_tensor(labels).to(self.device).long())
        replay = noise is not None
        if replay:
            z = torch.as_tensor(noise)
            if z.dim() == 4 and z.shape[1] == self.channels:
                z = z.permute(0, 2, 3, 1)
            self.replay.copy_(z.to(self.device, torch.float32).reshape(self.B, -1))
        self._plans[mode][replay].run(cur.cuda_stream, 1, use_graph)
        return self.loss_per_sample

    def profile_step(self, samples, labels, repeats=5):
        """Per-operator-class GPU time of one optimiser step: {tag: (ms per step, launches per step)} by hipEvents around
        every launch carrying the tag (eager runs; tags: 100 + kind forward / packing, 200 + kind reverse)."""
        self._run('step', samples, labels, None)
        plan = self._plans['step'][False]
        tags = sorted({(100 if forward else 200) + k for forward in (True, False) for k in range(1, 21)})
        out = {}
        for tag in tags:
            plan.profile(tag)
            for _ in range(repeats):
                self._run('step', samples, labels, None)
            ms, n = plan.profile_read()
            if n:
                out[tag] = (ms / repeats, n // repeats)
        plan.profile(-1)
        return out

    def step(self, samples, labels, noise=None, use_graph=False):
        """One optimiser step (train_score.py:145-173); returns the per-sample losses (device tensor) of this rank's
        samples.  With ``world`` > 1: backward, SUM all-reduce of the flat gradient buffer (RCCL), optimiser."""
        if self.world == 1:
            return self._run('step', samples, labels, noise, use_graph)
        from . import shard
        per = self._run('backward', samples, labels, noise, use_graph)
        shard.all_reduce_sum_(self.grads, self.world)
        if 'apply' not in self._plans:
            self._plans['apply'] = self._build('apply')
        self._plans['apply'][False].run(torch.cuda.current_stream(self.device).cuda_stream, 1, use_graph)
        return per

    def backward(self, samples, labels, noise=None):
        """Loss and gradients only (``loss.backward()`` without ``optimizer.step()``)."""
        return self._run('backward', samples, labels, noise)

    def loss(self, samples, labels, noise=None, ema=False):
        """Forward-only loss with the current or the EMA parameters (validation, train_score.py:172-185)."""
        return self._run('loss_ema' if ema else 'loss', samples, labels, noise)
