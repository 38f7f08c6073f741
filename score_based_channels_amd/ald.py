"""Annealed-Langevin channel estimation on the HIP path: the counterpart of the sampling loop in
``src/score_based_channels/test_score.py:118-171`` (= ``tune_hparams_score.py:100-148``).

The reference walks the 17 SNR points (and, in the tuner, the (alpha, beta) cells) one after another with a
batch of 100 channels and synchronises with the host after every step.  All of those trajectories follow the
same noise-level schedule in lock-step, so here they are ONE batch of ``T`` trajectories with per-trajectory
scalars: trajectory ``t`` estimates channel ``h_index[t]`` observed through pilots ``p_index[t]`` at noise
``local_noise[t]`` with hyper-parameters ``(alpha_step[t], beta_noise[t])``.  One ``sbc_plan`` holds a complete
Langevin step (score network + data-consistency/update/NMSE kernel + step counter), and a whole schedule is a
single ``plan.run(n_steps)``: per-step scalars come from device tables indexed by the device step counter, the
NMSE log stays on the device until it is read.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from . import plan as P
from .scorenet import _ptr


def snr_to_noise(snr_db, nt):
    """``noise_range = 10 ** (-snr_range / 10.) * config.data.image_size[1]`` (test_score.py:72-75)."""
    return 10 ** (-np.asarray(snr_db, np.float64) / 10.) * nt


def schedule_tables(sigmas_f32, sigma_end, levels, steps_each, alpha_step, beta_noise, local_noise, dc_boost=1.0):
    """Per-step scalars of the loop, computed in float64 exactly like the python code of
    test_score.py:137-165 and rounded to float32 where they meet a complex64 tensor.

    ``alpha_step``, ``beta_noise``, ``local_noise``: float64 arrays of equal length G (one row per distinct
    scalar group).  Returns (``sched`` float32 ``[G, n_steps, 4]`` = (alpha, dc_div, noise_scale, dc_boost),
    ``sigma_of_step`` float32 ``[n_steps]``).  ``dc_boost`` (test_mmse.py:231-233) stays a separate factor: the reference
    evaluates ``dc_boost * meas_grad / (...)`` left to right, so the kernel multiplies the gradient first."""
    levels = np.asarray(levels, np.int64)
    sig = np.asarray(sigmas_f32, np.float32)[levels].astype(np.float64)        # .item() of a float32 tensor
    alpha = np.asarray(alpha_step, np.float64)[:, None] * (sig[None, :] / float(sigma_end)) ** 2     # :143-144
    nscale = np.sqrt(2 * alpha * np.asarray(beta_noise, np.float64)[:, None])                        # :160
    dc_div = np.asarray(local_noise, np.float64)[:, None] / 2. + sig[None, :] ** 2                   # :165
    per_level = np.stack((alpha, dc_div, nscale, np.full_like(alpha, float(dc_boost))), axis=-1).astype(np.float32)
    sched = np.repeat(per_level, steps_each, axis=1)
    sigma_of_step = np.repeat(np.asarray(sigmas_f32, np.float32)[levels], steps_each)
    return np.ascontiguousarray(sched), np.ascontiguousarray(sigma_of_step)


def _as_c64(t, device):
    t = torch.as_tensor(t)
    if not t.is_complex():
        raise TypeError('expected a complex tensor')
    return t.to(device=device, dtype=torch.complex64).contiguous()


class AldBatch:
    """``T`` lock-step annealed-Langevin trajectories on one GPU.

    Htrue ``[nH, Nt, Nr]`` complex64 (normalised Hermitian channels, ``val_H`` of test_score.py:112-113),
    P ``[nP, Np, Nt]`` complex64 (conj-transposed pilots, ``val_P`` of :109-111).  Per-trajectory arrays
    (length T): ``h_index``, ``p_index``, ``local_noise``, ``alpha_step``, ``beta_noise`` (scalars broadcast).
    ``levels``: noise-level indices to walk (default: the full schedule), ``steps_each``: Langevin steps per
    level (test_score.py:56).  Noise: ``seed`` keys the in-kernel Philox stream of trajectory ``traj_id[t]``
    (independent of batching / world size); pass ``step_noise`` ``[n_steps, T, Nt, Nr]`` complex64 to replay
    externally drawn noise instead (parity runs; fewer rows than ``n_steps`` are allowed when the run stops early).
    ``lanes``: the score plan with launch lanes (small batches; ``ScoreNet.bind``), default by batch size.
    """

    def __init__(self, net, Htrue, P_pilots, h_index, p_index, local_noise, alpha_step=3e-11, beta_noise=0.01,
                 levels=None, steps_each=3, seed=0, traj_id=None, step_noise=None, dc_boost=1.0, lanes=None):
        dev = net.device
        self.net = net
        self.H = _as_c64(Htrue, dev)
        self.P = _as_c64(P_pilots, dev)
        _, self.nt, self.nr = self.H.shape
        self.np_ = self.P.shape[1]
        if self.P.shape[2] != self.nt:
            raise ValueError('P must be [nP, Np, Nt=%d], got %s' % (self.nt, tuple(self.P.shape)))
        h_index = np.array(h_index, np.int32)
        T = self.T = int(h_index.shape[0])
        p_index = np.array(np.broadcast_to(np.asarray(p_index, np.int32), (T,)))
        if h_index.min() < 0 or h_index.max() >= self.H.shape[0] or p_index.min() < 0 or p_index.max() >= self.P.shape[0]:
            raise IndexError('h_index / p_index out of range')
        ln = np.broadcast_to(np.asarray(local_noise, np.float64), (T,))
        a0 = np.broadcast_to(np.asarray(alpha_step, np.float64), (T,))
        be = np.broadcast_to(np.asarray(beta_noise, np.float64), (T,))
        self.local_noise = ln
        self.levels = np.arange(net.num_classes) if levels is None else np.asarray(levels, np.int64)
        self.steps_each = int(steps_each)
        self.n_steps = len(self.levels) * self.steps_each
        rows, group = np.unique(np.stack((a0, be, ln), axis=1), axis=0, return_inverse=True)
        sched, sig_step = schedule_tables(net._sigmas_np, net.config.model.sigma_end, self.levels, self.steps_each,
                                          rows[:, 0], rows[:, 1], rows[:, 2], dc_boost)
        i32 = dict(dtype=torch.int32, device=dev)
        self.d_sched = torch.from_numpy(sched).to(dev)
        self.d_sigma_of_step = torch.from_numpy(sig_step).to(dev)
        self.d_group = torch.from_numpy(group.reshape(-1).astype(np.int32)).to(dev)
        self.d_hidx = torch.from_numpy(np.ascontiguousarray(h_index)).to(dev)
        self.d_pidx = torch.from_numpy(np.ascontiguousarray(p_index)).to(dev)
        tid = np.arange(T, dtype=np.int64) if traj_id is None else np.asarray(traj_id, np.int64)
        self.d_traj = torch.from_numpy(np.ascontiguousarray(tid)).to(dev)
        self.d_meas_scale = torch.from_numpy(np.sqrt(ln).astype(np.float32)).to(dev)     # :124
        self.d_step = torch.zeros(1, **i32)
        self.d_nmse = torch.zeros(self.n_steps, T, dtype=torch.float32, device=dev)
        self.Y = torch.zeros(T, self.np_, self.nr, dtype=torch.complex64, device=dev)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._done = 0
        self._gstream = None
        self.step_noise = None
        if step_noise is not None:
            self.step_noise = _as_c64(step_noise, dev)
            if tuple(self.step_noise.shape[1:]) != (T, self.nt, self.nr) or not 0 < self.step_noise.shape[0] <= self.n_steps:
                raise ValueError('step_noise must be [k <= n_steps=%d, T=%d, Nt, Nr] (rows = steps that will be run)'
                                 % (self.n_steps, T))
        # score network bound to this batch: its input buffer IS the current estimate X (complex64 view)
        # (lanes: None = by batch size; a driver that splits a chunk into concurrent sub-batches passes False for them)
        self.bound = net.bind(T, self.nt, self.nr, step=self.d_step, sigma_of_step=self.d_sigma_of_step,
                              use_labels=False, lanes=lanes)
        self.X = torch.view_as_complex(self.bound.x)                    # [T, Nt, Nr] complex64, in place
        self.uses_lanes = any(op.lane or op.signal for op in self.bound.plan.ops)     # small batches: skip branches on launch lanes
        self._lang = _lib.sbc_langevin(
            X=_ptr(self.bound.x), score=_ptr(self.bound.out), P=_ptr(torch.view_as_real(self.P)),
            p_index=_ptr(self.d_pidx), Y=_ptr(torch.view_as_real(self.Y)), Htrue=_ptr(torch.view_as_real(self.H)),
            h_index=_ptr(self.d_hidx), sched=_ptr(self.d_sched), group=_ptr(self.d_group),
            noise=_ptr(torch.view_as_real(self.step_noise)) if self.step_noise is not None else None,
            nmse=_ptr(self.d_nmse), step=_ptr(self.d_step), traj_id=_ptr(self.d_traj),
            meas_scale=_ptr(self.d_meas_scale), seed=self.seed, n_steps=self.n_steps, Nt=self.nt, Nr=self.nr,
            Np=self.np_)
        lang = _lib.sbc_op(kind=P.LANGEVIN, B=T, ext=C.cast(C.pointer(self._lang), C.c_void_p))
        inc = _lib.sbc_op(kind=P.STEP_INC, B=1, out=_ptr(self.d_step))
        self._step_ops = list(self.bound.ops) + [lang, inc]
        self.plan = _lib.Plan(self._step_ops, keepalive=self)
        self.score_plan = _lib.Plan(list(self.bound.ops), keepalive=self)
        # run_leading / run_following: ONE Langevin step's records cut at ~45 % of a score evaluation -- [:k] and [k:]
        self._lag_plan = self._rest_plan = None

    def _stream(self):
        return torch.cuda.current_stream(self.net.device).cuda_stream

    def close(self):
        """Destroy the two plans and drop their back-references: ``AldBatch`` <-> ``Plan(keepalive=self)`` is a reference
        cycle, so without this the slot buffers of a finished chunk (GBs) live until the cyclic collector runs.  The
        tensors already handed out (``X``, ``nmse_log()``) stay valid; the batch cannot run again."""
        for pl in (self.plan, self.score_plan, self._lag_plan, self._rest_plan):
            if pl is not None:
                pl.close()
                pl._keep = None

    # --- inputs -------------------------------------------------------------------------------------
    def set_init(self, X0):
        """``current = init_val_H.clone()`` (test_score.py:115,126); ``X0`` ``[T, Nt, Nr]`` complex64."""
        self.X.copy_(_as_c64(X0, self.net.device))
        self.d_step.zero_()
        self._done = 0

    def set_measurements(self, Y):
        self.Y.copy_(_as_c64(Y, self.net.device))

    def synthesize_measurements(self, noise=None):
        """``val_Y = P H + sqrt(local_noise) * randn`` (test_score.py:122-124) on the device.  ``noise``
        ``[T, Np, Nr]`` complex64 replays an external draw; otherwise Philox keyed by (seed, traj_id)."""
        nz = _as_c64(noise, self.net.device) if noise is not None else None
        ext = _lib.sbc_langevin(**{f: getattr(self._lang, f) for f, _ in _lib.sbc_langevin._fields_})
        ext.noise = _ptr(torch.view_as_real(nz)) if nz is not None else None
        op = _lib.sbc_op(kind=P.MEASURE, B=self.T, ext=C.cast(C.pointer(ext), C.c_void_p))
        _lib.check(_lib.lib().sbc_op_launch(C.byref(op), C.c_void_p(self._stream())))
        torch.cuda.current_stream(self.net.device).synchronize()       # ext / nz go out of scope
        return self.Y

    # --- execution ----------------------------------------------------------------------------------
    def run(self, n_steps=None, use_graph=False):
        """Advance every trajectory by ``n_steps`` Langevin steps (default: the rest of the schedule).
        Asynchronous: returns once the launches are queued on the current stream."""
        done = self._done
        n = self.n_steps - done if n_steps is None else int(n_steps)
        if n < 0 or done + n > self.n_steps:
            raise ValueError('schedule has %d steps, %d already done, %d requested' % (self.n_steps, done, n))
        if self.step_noise is not None and done + n > self.step_noise.shape[0]:
            raise ValueError('replayed noise covers %d steps, %d requested' % (self.step_noise.shape[0], done + n))
        if use_graph:
            # hipGraph capture is not allowed on the legacy default stream: replay on a private stream that is
            # ordered after / before the caller's current stream
            cur = torch.cuda.current_stream(self.net.device)
            if self._gstream is None:
                self._gstream = torch.cuda.Stream(self.net.device)
            self._gstream.wait_stream(cur)
            self.plan.run(self._gstream.cuda_stream, n, True)
            cur.wait_stream(self._gstream)
        else:
            self.plan.run(self._stream(), n, False)
        self._done = done + n

    # --- two concurrent sub-batch streams, the second ~0.45 of a step behind the first (driver.run_concurrently) ------------------
    # One stream is in the high-resolution, issue-bound part of a step while the other is in the dozens of small low-resolution
    # launches: ~4 % per step (sustained two-stream step 5.47 -> 5.38 ms for lags of 0.38-0.52 of a step in round 4, nothing outside).
    # The lag is made of the work itself, whatever the array size: the records of one step are cut at the first record of refine31
    # (the middle of the low-resolution stretch) into a head [:k] and a rest [k:]; the LEADING stream runs the head of its first step
    # alone on the chip (full-width grids), then lets the FOLLOWING stream start; the following stream ends with the rest of its last
    # step alone (full width again) once the leader is done.  Every stream runs exactly its own n steps, record for record -- no
    # throw-away evaluation (round 4 ran 0.45 of a score evaluation for nothing per call, 2 % of a 20-step call).
    def _cut_step(self):
        if self._lag_plan is None:
            if self.uses_lanes:
                raise RuntimeError('a plan with launch lanes is not cut into a leading and a following part (events would cross the cut)')
            names = [op.name for op in self.bound.plan.ops]
            k = next((i for i, nm in enumerate(names) if nm.startswith('refine31.')), len(names) // 2)
            k = int(os.environ.get('SBC_LAG_RECORDS', k))        # (A/B aid: where the cut is)
            k = min(max(k, 1), len(self._step_ops) - 1)
            self._lag_plan = _lib.Plan(self._step_ops[:k], keepalive=self)
            self._rest_plan = _lib.Plan(self._step_ops[k:], keepalive=self)
        return self._lag_plan, self._rest_plan

    def _check_steps(self, n_steps):
        done = self._done
        n = self.n_steps - done if n_steps is None else int(n_steps)
        if n < 1 or done + n > self.n_steps:
            raise ValueError('schedule has %d steps, %d already done, %d requested' % (self.n_steps, done, n))
        if self.step_noise is not None and done + n > self.step_noise.shape[0]:
            raise ValueError('replayed noise covers %d steps, %d requested' % (self.step_noise.shape[0], done + n))
        return n

    def run_leading(self, n_steps, head_done, width=0):
        """``run(n_steps)`` for the FIRST of two concurrent sub-batch streams: the head of the first step at the process-default grid width
        (``set_persistent_cus(0)``: every CU unless ``sbc_set_persistent_cus`` / ``SBC_PERSIST_CUS`` say otherwise), then
        ``head_done()`` (the caller records the event the following stream waits for), then everything else at ``width`` CUs."""
        n = self._check_steps(n_steps)
        head, rest = self._cut_step()
        st = self._stream()
        try:
            head.set_persistent_cus(0)
            head.run(st, 1, False)
        finally:
            head_done()
        rest.set_persistent_cus(width)
        rest.run(st, 1, False)
        if n > 1:
            self.plan.run(st, n - 1, False)
        self._done += n

    def run_following(self, n_steps, wait_head, wait_leader, width=0):
        """``run(n_steps)`` for the SECOND stream: starts behind ``wait_head()`` (the caller makes this stream wait for the leader's
        event), runs its steps at ``width`` CUs, and the rest of its last step at full width behind ``wait_leader()``."""
        n = self._check_steps(n_steps)
        head, rest = self._cut_step()
        st = self._stream()
        wait_head()
        if n > 1:
            self.plan.run(st, n - 1, False)
        head.set_persistent_cus(width)
        head.run(st, 1, False)
        wait_leader()
        rest.set_persistent_cus(0)
        rest.run(st, 1, False)
        self._done += n

    def set_persistent_cus(self, n):
        """Grid width (CUs) of the persistent kernels of THIS batch's launches, 0 = the process default -- all CUs unless
        ``sbc_set_persistent_cus`` / ``SBC_PERSIST_CUS`` narrowed it (``sbc_plan_set_persistent_cus``: a
        field of the batch's plans, not process state -- batches on other threads / streams / devices are unaffected)."""
        self._persist = int(n)
        for pl in (self.plan, self.score_plan, self._lag_plan, self._rest_plan):
            if pl is not None:
                pl.set_persistent_cus(n)

    def steps_done(self):
        return self._done

    def rewind(self):
        """Reset the step counter without touching X (benchmark loops re-walk the first steps)."""
        self.d_step.zero_()
        self._done = 0

    def score_only(self):
        """One evaluation of the score network on the current estimate (no update); benchmark helper."""
        self.score_plan.run(self._stream(), 1, False)
        return torch.view_as_complex(self.bound.out)

    def nmse_log(self):
        """``[n_steps, T]`` float32 device tensor: row k = NMSE after Langevin step k (test_score.py:168-170)."""
        return self.d_nmse


class AldPair:
    """Two sub-batches of one lock-step chunk as ONE plan with two launch lanes ("one captured graph with two
    branches"): sub-batch A on the run stream, sub-batch B on lane 1, ``k_steps`` Langevin steps unrolled into one record list in which B
    starts behind the head of A's first step (the record ``driver.run_concurrently`` cuts at: A is in its issue-bound full-resolution
    launches while B is in the small low-resolution ones) and keeps that lag for the whole list; the lanes join at the end of the list.
    One host thread issues both lanes -- or one hipGraph holds both branches (``run(use_graph=True)``).  Every record is a record of A's
    or B's own step plan with its own buffers: results are those of the two batches run alone, bit for bit."""

    def __init__(self, a, b, k_steps=20):
        if a.uses_lanes or b.uses_lanes or a.net is not b.net:
            raise ValueError('AldPair takes two sequential-plan batches of one network')
        self.a, self.b, self.k = a, b, int(k_steps)
        names = [op.name for op in a.bound.plan.ops]
        cut = next((i for i, nm in enumerate(names) if nm.startswith('refine31.')), len(names) // 2)
        self.cut = cut
        self._plans = {}
        self._keep = []

    def _plan(self, k):
        if k not in self._plans:
            import copy
            A, B = self.a._step_ops, self.b._step_ops
            n = len(A)
            seq_a = [(s, i) for s in range(k) for i in range(n)]
            seq_b = list(seq_a)
            ops = []

            def rec(src, lane, signal=0, wait=0):
                o = _lib.sbc_op()
                C.memmove(C.byref(o), C.byref(src), C.sizeof(o))
                o.lane, o.signal = lane, signal
                o.wait[0], o.wait[1] = wait, 0
                self._keep.append(o)
                return o
            # A's head alone, then A and B record by record (B behind by the head), then B's tail
            for j, (s, i) in enumerate(seq_a[:self.cut + 1]):
                ops.append(rec(A[i], 0, signal=1 if j == self.cut else 0))
            rest_a = seq_a[self.cut + 1:]
            for j in range(max(len(rest_a), len(seq_b))):
                if j < len(rest_a):
                    ops.append(rec(A[rest_a[j][1]], 0))
                if j < len(seq_b):
                    ops.append(rec(B[seq_b[j][1]], 1, wait=1 if j == 0 else 0))
            self._plans[k] = _lib.Plan(ops, keepalive=(self.a, self.b))
        return self._plans[k]

    def set_persistent_cus(self, n):
        self._width = int(n)
        for pl in self._plans.values():
            pl.set_persistent_cus(n)

    def run(self, n_steps, use_graph=False, stream=None):
        """Advance both sub-batches by ``n_steps`` (whole lists of ``k_steps`` steps, then one shorter list)."""
        for x in (self.a, self.b):
            x._check_steps(n_steps)
        cur = torch.cuda.current_stream(self.a.net.device)
        if use_graph:                                  # (capture is not allowed on the legacy default stream: AldBatch.run)
            if getattr(self, '_gstream', None) is None:
                self._gstream = torch.cuda.Stream(self.a.net.device)
            self._gstream.wait_stream(cur)
            st = self._gstream.cuda_stream
        else:
            st = cur.cuda_stream if stream is None else stream
        done = 0
        while done < n_steps:
            k = min(self.k, n_steps - done)
            pl = self._plan(k)
            pl.set_persistent_cus(getattr(self, '_width', 0))
            pl.run(st, 1, 2 if use_graph else False)         # (2: the lanes stay parallel branches of the captured graph)
            done += k
        if use_graph:
            cur.wait_stream(self._gstream)
        self.a._done += n_steps
        self.b._done += n_steps

    def close(self):
        for pl in self._plans.values():
            pl.close()
            pl._keep = None
        self._plans = {}
