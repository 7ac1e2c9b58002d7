"""Host-side mirror of the reference's evaluator interface over the C ABI (include/cmax_hip.h).

Names follow the reference so the parity tests read like its own call sites:

  FrontendEvaluator  <->  cmax_slam::AngVelEstimator's hot-path members
      set_packet                     event_subset_ / time_packet_ hand-over   (ang_vel_estimator.cpp:137-147)
      computeImageOfWarpedEvents     local_image_warped_events.cpp:10-57
      contrast_f / contrast_df / contrast_fdf   local_contrast_{f,df,fdf}   (local_optim_contrast_gsl.cpp:20-70)

  BackendEvaluator   <->  PoseGraphOptimizer + EventWarper hot-path members
      set_window                     processTimeWindow hand-over             (pose_graph_optimizer.cpp:283-293)
      computeImageOfWarpedEvents     event_pano_warper.cpp:167-231
      contrast_f / contrast_df / contrast_fdf   global_contrast_{f,df,fdf}  (global_optim_contrast_gsl_analytical.cpp:17-81)

contrast_* return the GSL-side values: f = -contrast, df = -gradient.  eval() returns the un-negated pair.
All compute happens in libcmaxhip.so's HIP kernels; numpy only carries host buffers across the ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (GRAD_ADJOINT, GRAD_PLANES, MEAN_SQUARE, VARIANCE, CmaxHipError, c_dp, c_fp, c_i64p, c_u16p,
                   check)

__all__ = ["FrontendEvaluator", "BackendEvaluator", "EventStore", "CmaxHipError", "VARIANCE", "MEAN_SQUARE", "GRAD_PLANES",
           "GRAD_ADJOINT"]


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _dp(a):
    return a.ctypes.data_as(c_dp)


class _Evaluator:
    def __init__(self):
        self._L = _lib.lib()
        self._ctx = _lib.ctx_p()

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.cmx_destroy(self._ctx)
            self._ctx = _lib.ctx_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, status):
        check(self._ctx, status)

    def _io_buffers(self, n):
        """Persistent ctypes in/out buffers for the eval calls (+ numpy views on them): `ndarray.ctypes.data_as` and
        fresh arrays cost ~5 us per call -- a tenth of a front-end evaluation -- and a C++ host pays none of it."""
        n = max(int(n), 1)
        if getattr(self, "_io_n", 0) != n:
            self._xb, self._gb, self._cb = (C.c_double * n)(), (C.c_double * n)(), C.c_double()
            self._xin = np.frombuffer(self._xb, dtype=np.float64)
            self._gout = np.frombuffer(self._gb, dtype=np.float64)
            self._xp, self._gp = C.cast(self._xb, c_dp), C.cast(self._gb, c_dp)
            self._cref = C.pointer(self._cb)
            self._io_n = n

    def set_option(self, key, value):
        self._ck(self._L.cmx_set_option(self._ctx, int(key), int(value)))

    def set_grad_mode(self, mode):
        self.set_option(_lib.OPT_GRAD_MODE, mode)

    def set_splat_mode(self, mode):
        """0 = global atomics, 1 = LDS-privatised (events sorted by destination tile)."""
        self.set_option(_lib.OPT_SPLAT_MODE, mode)

    def stats(self):
        s = np.zeros(22)
        self._ck(self._L.cmx_get_stats(self._ctx, _dp(s), 22))
        return {"rebins": int(s[0]), "fallback_frac": float(s[1]), "chunks": int(s[2]), "events": int(s[3]),
                "reuse_hits": int(s[4]), "sharded_host_syncs": int(s[5]), "exchange_misses": int(s[6]), "exchange_tiles": int(s[7]), "comm_bytes": int(s[8]), "spec_images": int(s[9]), "spec_hits": int(s[10]),
                "gated_launches": int(s[11]), "gated_hits": int(s[12]), "chain_solves": int(s[13]), "chain_slots": int(s[14]),
                "chain_takeovers": int(s[15]), "chain_warm_starts": int(s[16]), "fused_evals": int(s[17]), "fused_redos": int(s[18]), "one_launch_evals": int(s[19]), "fused_timeouts": int(s[20]), "self_serve_evals": int(s[21])}

    def hint_next_df(self, threshold, mode):
        """cmx_hint_next_df: the next cost-only evaluation's value f = -contrast decides (mode 1: f < threshold, 2: f <= threshold,
        3: not f >= threshold, 4: always) whether the gradient pass is queued behind it."""
        self._ck(self._L.cmx_hint_next_df(self._ctx, float(threshold), int(mode)))

    def set_fast_path(self):
        """The production configuration and the library's default: adjoint gradient + LDS-privatised splat (+ image
        reuse, on by default)."""
        self.set_grad_mode(_lib.GRAD_ADJOINT)
        self.set_splat_mode(1)

    def set_reference_path(self):
        """The reference's own data flow: derivative planes, one global fp32 atomic per vote (14-50x slower; also
        what produces the derivative images)."""
        self.set_grad_mode(_lib.GRAD_PLANES)
        self.set_splat_mode(0)

    def set_deterministic(self, on=True):
        """Bitwise run-to-run reproducible evaluations (CMX_OPT_DETERMINISTIC): integer vote accumulation in global
        memory, fixed summation orders everywhere; ~10-20 % slower.  Applies to the production path."""
        self.set_option(_lib.OPT_DETERMINISTIC, 1 if on else 0)

    def set_stream_priority(self, level):
        """cmx_set_stream_priority: > 0 highest (the front end beside a back-end solve), 0 normal, < 0 lowest."""
        self._ck(self._L.cmx_set_stream_priority(self._ctx, int(level)))

    def set_sched_class(self, sched_class):
        """cmx_set_sched_class: +1 urgent (the front end), 0 normal, -1 background (the back end: holds its next evaluation while an
        urgent context of the same device is busy)."""
        self._ck(self._L.cmx_set_sched_class(self._ctx, int(sched_class)))

    def set_cu_mask(self, n_cus=None, first=0, mask_words=None):
        """cmx_set_cu_mask: run on `n_cus` compute units starting at bit `first` (on MI355X consecutive bits walk the eight
        XCDs, so any run of bits is spread over all of them), or on an explicit list of 32-bit words; None / 0 = all."""
        if mask_words is None:
            if not n_cus:
                self._ck(self._L.cmx_set_cu_mask(self._ctx, None, 0))
                return
            bits = ((1 << int(n_cus)) - 1) << int(first)
            nw = max(8, (bits.bit_length() + 31) // 32)
            mask_words = [(bits >> (32 * i)) & 0xffffffff for i in range(nw)]
        arr = (C.c_uint32 * len(mask_words))(*mask_words)
        self._ck(self._L.cmx_set_cu_mask(self._ctx, arr, len(mask_words)))

    def set_stream(self, hip_stream_handle):
        """Run on a caller-owned stream (an int / void* hipStream_t, e.g. torch.cuda.current_stream().cuda_stream)."""
        self._ck(self._L.cmx_set_stream(self._ctx, C.c_void_p(hip_stream_handle or None)))

    # split-phase plumbing (multi-GPU): the accumulation planes between splat and blur/reduce
    def accum_capacity(self):
        return int(self._L.cmx_accum_capacity(self._ctx))

    def set_accum_buffer(self, device_ptr, n_floats):
        self._ck(self._L.cmx_set_accum_buffer(self._ctx, C.c_void_p(device_ptr), int(n_floats)))

    def accum_ptr(self):
        return self._L.cmx_accum_ptr(self._ctx)

    def accum_count(self):
        return int(self._L.cmx_accum_count(self._ctx))

    # native RCCL exchange inside the evaluator (one process per GPU)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(None, _lib.lib().cmx_comm_unique_id(buf))
        return buf.raw

    def comm_attach(self, unique_id, rank, nranks):
        self._ck(self._L.cmx_comm_attach(self._ctx, C.c_char_p(unique_id), int(rank), int(nranks)))

    def comm_attach_custom(self, fn, rank, nranks):
        """Exchange through a caller-supplied all-reduce: fn(device_ptr, count, dtype, op, hip_stream) -> 0 on success
        (dtype / op: _lib.DT_* / _lib.OP_*), called at the evaluator's exchange points in place of RCCL."""
        def tramp(_user, buf, count, dt, op, stream):
            try:
                return int(fn(buf, count, dt, op, stream) or 0)
            except Exception:  # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return -1
        self._comm_cb = _lib.ALLREDUCE_FN(tramp)  # keep the trampoline alive as long as it is attached
        self._ck(self._L.cmx_comm_attach_custom(self._ctx, C.cast(self._comm_cb, C.c_void_p), None, int(rank), int(nranks)))

    def comm_detach(self):
        self._ck(self._L.cmx_comm_detach(self._ctx))

    def comm_info(self):
        """{'rank', 'nranks', 'transport'} as the attached communicator itself reports them (cmx_comm_info)."""
        r, n, t = C.c_int(), C.c_int(), C.c_int()
        self._ck(self._L.cmx_comm_info(self._ctx, C.byref(r), C.byref(n), C.byref(t)))
        return {"rank": r.value, "nranks": n.value, "transport": ("none", "rccl", "custom/direct")[t.value]}

    def set_grad_buffer(self, device_ptr, n_doubles):
        self._ck(self._L.cmx_set_grad_buffer(self._ctx, C.c_void_p(device_ptr), int(n_doubles)))

    def grad_count(self):
        return int(self._L.cmx_grad_count(self._ctx))

    def finish_begin(self, want_grad=True):
        fn = self._L.cmx_frontend_finish_begin if isinstance(self, FrontendEvaluator) else self._L.cmx_backend_finish_begin
        self._ck(fn(self._ctx, int(bool(want_grad))))

    def finish_end(self, want_grad=True):
        fe = isinstance(self, FrontendEvaluator)
        n = 3 if fe else self.num_params
        c = C.c_double()
        g = np.zeros(max(n, 1))
        fn = self._L.cmx_frontend_finish_end if fe else self._L.cmx_backend_finish_end
        self._ck(fn(self._ctx, C.byref(c), _dp(g) if want_grad else None))
        return c.value, (g[:n] if want_grad else None)

    def eval_many(self, xs, want_grad=True):
        """m independent evaluations in one call (cmx_*_eval_many): xs = m x n_params.  Returns (contrasts[m], grads[m, n] | None)."""
        fe = isinstance(self, FrontendEvaluator)
        n = 3 if fe else self.num_params
        xs = np.ascontiguousarray(np.asarray(xs, np.float64).reshape(-1, max(n, 1)))
        m = xs.shape[0]
        c = np.zeros(m)
        g = np.zeros((m, max(n, 1))) if want_grad else None
        fn = self._L.cmx_frontend_eval_many if fe else self._L.cmx_backend_eval_many
        self._ck(fn(self._ctx, m, _dp(xs), _dp(c), _dp(g) if want_grad else None))
        return c, (g[:, :n] if want_grad else None)

    def eval_each(self, xs, want_grad=True):
        """m evaluations one after the other inside ONE native call (cmx_*_eval_each): the same as [self.eval(x) for x in xs]
        without the interpreter between them.  Returns (contrasts[m], grads[m, n] | None)."""
        return self.prepare_eval_each(xs, want_grad)()

    def prepare_eval_each(self, xs, want_grad=True):
        """eval_each in two steps: everything the interpreter does around the native call (array conversion, output buffers, ctypes
        pointers: ~10 us) now, the call itself when the returned function is called -- for callers that time a short list."""
        fe = isinstance(self, FrontendEvaluator)
        n = 3 if fe else self.num_params
        xs = np.ascontiguousarray(np.asarray(xs, np.float64).reshape(-1, max(n, 1)))
        m = xs.shape[0]
        c = np.zeros(m)
        g = np.zeros((m, max(n, 1))) if want_grad else None
        fn = self._L.cmx_frontend_eval_each if fe else self._L.cmx_backend_eval_each
        ctx, px, pc, pg, ck = self._ctx, _dp(xs), _dp(c), (_dp(g) if want_grad else None), self._ck

        def call():
            ck(fn(ctx, m, px, pc, pg))
            return c, (g[:, :n] if want_grad else None)
        return call

    def timing_enable(self, on=True, every=1):
        """on: True = all kernel classes, False = off, or an iterable of class names (e.g. ["splat"]).
        every: sample every n-th evaluation only (the per-event kernels are timed through events attached to the
        kernel itself, which costs ~1.5 us per timed launch)."""
        if on is True:
            mask = (1 << _lib.T_COUNT) - 1
        elif not on:
            mask = 0
        else:
            mask = sum(1 << _lib.T_NAMES.index(n) for n in on)
        self._ck(self._L.cmx_timing_enable(self._ctx, mask | (int(every) << 8) if mask else 0))

    def timing_get(self):
        """{'splat': (ms, launches), ...} accumulated since the last call."""
        ms = np.zeros(_lib.T_COUNT)
        n = np.zeros(_lib.T_COUNT, np.int64)
        self._ck(self._L.cmx_timing_get(self._ctx, _dp(ms), n.ctypes.data_as(c_i64p)))
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(_lib.T_NAMES)}


class EventStore:
    """Device-resident copy of the event stream (the reference's AngVelEstimator::events_): push chunks as they
    arrive, cut packets / windows from it by global event index, drop the prefix like deleteOldEvents."""

    def __init__(self, W, H, capacity, device=0, devices=None):
        """devices = [d0, d1, ...] (a group's member list): one replica of the stream on every distinct device
        (cmx_events_create_group); a group handle over the same list cuts its windows from it member by member."""
        self._L = _lib.lib()
        self._h = C.c_void_p()
        if devices is None:
            check(None, self._L.cmx_events_create(C.byref(self._h), int(device), int(W), int(H), int(capacity)))
        else:
            dv = (C.c_int * len(devices))(*[int(d) for d in devices])
            check(None, self._L.cmx_events_create_group(C.byref(self._h), dv, len(devices), int(W), int(H), int(capacity)))

    @property
    def devices(self):
        dv = (C.c_int * 16)()
        n = self._L.cmx_events_devices(self._h, dv, 16)
        return list(dv[:n])

    def _ck(self, status):
        if status != _lib.OK:
            raise CmaxHipError(status, self._L.cmx_status_string(status).decode() + ": " +
                               self._L.cmx_events_last_error(self._h).decode())

    def push(self, x, y, t_ns):
        x, y, t = _c(x, np.uint16), _c(y, np.uint16), _c(t_ns, np.int64)
        self._ck(self._L.cmx_events_push(self._h, len(x), x.ctypes.data_as(c_u16p), y.ctypes.data_as(c_u16p),
                                         t.ctypes.data_as(c_i64p)))

    def push_aos(self, events):
        """cmx_events_push_aos: records with fields x, y, sec, nsec (msg->events as it lies in the host's memory)."""
        ev = np.ascontiguousarray(events)
        lay = _lib.aos_layout_of(ev)
        self._ck(self._L.cmx_events_push_aos(self._h, len(ev), ev.ctypes.data_as(C.c_void_p), C.byref(lay)))

    def drop_before(self, global_index):
        self._ck(self._L.cmx_events_drop_before(self._h, int(global_index)))

    @property
    def begin(self):
        return int(self._L.cmx_events_begin(self._h))

    @property
    def end(self):
        return int(self._L.cmx_events_end(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.cmx_events_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrontendEvaluator(_Evaluator):
    def __init__(self, W, H, lut, device=0):
        super().__init__()
        self.W, self.H = int(W), int(H)
        lut = _c(lut, np.float64).reshape(-1)
        if lut.size != self.W * self.H * 3:
            raise ValueError("lut must hold W*H*3 doubles")
        self._ck(self._L.cmx_frontend_create(C.byref(self._ctx), int(device), self.W, self.H, _dp(lut)))
        self.n_events = 0

    def set_packet(self, x, y, t_ns, t_ref_ns, fx, fy, cx, cy, event_batch_size=100, blur_sigma=1.0,
                   contrast_measure=VARIANCE):
        x, y, t = _c(x, np.uint16), _c(y, np.uint16), _c(t_ns, np.int64)
        if not (len(x) == len(y) == len(t)):
            raise ValueError("x, y, t_ns must have equal length")
        self._ck(self._L.cmx_frontend_set_packet(
            self._ctx, len(x), x.ctypes.data_as(c_u16p), y.ctypes.data_as(c_u16p), t.ctypes.data_as(c_i64p),
            int(t_ref_ns), float(fx), float(fy), float(cx), float(cy), int(event_batch_size), float(blur_sigma),
            int(contrast_measure)))
        self.n_events = len(x)

    def set_packet_aos(self, events, t_ref_ns, fx, fy, cx, cy, event_batch_size=100, blur_sigma=1.0, contrast_measure=VARIANCE):
        """cmx_frontend_set_packet_aos: `events` = structured array of records with fields x, y, sec, nsec (e.g. _lib.DVS_EVENT_DTYPE)."""
        ev = np.ascontiguousarray(events)
        lay = _lib.aos_layout_of(ev)
        self._ck(self._L.cmx_frontend_set_packet_aos(self._ctx, len(ev), ev.ctypes.data_as(C.c_void_p), C.byref(lay), int(t_ref_ns),
                                                     float(fx), float(fy), float(cx), float(cy), int(event_batch_size),
                                                     float(blur_sigma), int(contrast_measure)))
        self.n_events = len(ev)

    def set_packet_from(self, store, first, count, t_ref_ns, fx, fy, cx, cy, event_batch_size=100, blur_sigma=1.0,
                        contrast_measure=VARIANCE):
        """The packet events_[first, first+count) cut from a device-resident EventStore (no host copy)."""
        self._ck(self._L.cmx_frontend_set_packet_from(self._ctx, store._h, int(first), int(count), int(t_ref_ns), float(fx),
                                                      float(fy), float(cx), float(cy), int(event_batch_size),
                                                      float(blur_sigma), int(contrast_measure)))
        self.n_events = int(count)

    def prepare(self, ang_vel_hint=(0.0, 0.0, 0.0)):
        """cmx_frontend_prepare: queue the packet's tile sort / streams / chunk table now (non-blocking); results never depend on
        the hint."""
        h = np.ascontiguousarray(ang_vel_hint, dtype=np.float64)
        self._ck(self._L.cmx_frontend_prepare(self._ctx, _dp(h)))

    def eval(self, ang_vel, want_grad=True):
        """(contrast, gradient[3] | None) -- what computeContrast returns."""
        self._io_buffers(3)
        self._xin[:] = ang_vel
        rc = self._L.cmx_frontend_eval(self._ctx, self._xp, self._cref, self._gp if want_grad else None)
        if rc:
            self._ck(rc)
        return self._cb.value, (self._gout.copy() if want_grad else None)

    def accumulate(self, ang_vel, want_grad=True):
        om = _c(ang_vel, np.float64)
        self._ck(self._L.cmx_frontend_accumulate(self._ctx, _dp(om), int(bool(want_grad))))

    def finish(self, want_grad=True):
        c = C.c_double()
        g = np.zeros(3)
        self._ck(self._L.cmx_frontend_finish(self._ctx, C.byref(c), _dp(g) if want_grad else None))
        return c.value, (g if want_grad else None)

    # --- reference-named entry points
    def computeImageOfWarpedEvents(self, ang_vel, want_deriv=False, blur=True):
        """image_warped (H x W fp32) [, image_warped_deriv (H x W x 3 fp32)]."""
        om = _c(ang_vel, np.float64)
        iwe = np.empty((self.H, self.W), np.float32)
        d = np.empty((self.H, self.W, 3), np.float32) if want_deriv else None
        self._ck(self._L.cmx_frontend_get_iwe(self._ctx, _dp(om), int(bool(blur)), iwe.ctypes.data_as(c_fp),
                                              d.ctypes.data_as(c_fp) if want_deriv else None))
        return (iwe, d) if want_deriv else iwe

    def contrast_fdf(self, v):
        c, g = self.eval(v, True)
        return -c, -g

    def contrast_f(self, v):
        return -self.eval(v, False)[0]

    def contrast_df(self, v):
        return -self.eval(v, True)[1]

    def setupProblemAndOptimize(self, ang_vel):
        """FR-CG solve from the warm start `ang_vel` (local_optim_contrast_gsl.cpp:74-233).
        Returns (ang_vel_estimate, report dict)."""
        x = np.array(ang_vel, dtype=np.float64, order="C", copy=True)
        rep = _lib.SolveReport()
        self._ck(self._L.cmx_frontend_solve(self._ctx, _dp(x), C.byref(rep)))
        return x, {f: getattr(rep, f) for f, _ in rep._fields_}


class BackendEvaluator(_Evaluator):
    def __init__(self, W, H, lut, pano_width, pano_height, device=0, devices=None, transport=0):
        """devices = [d0, d1, ...]: a one-process multi-GPU GROUP (cmx_backend_create_group) behind the same interface --
        set_window shards the window, eval / setupProblemAndOptimize fan out and return one contrast / gradient.  The same
        device may be listed more than once (members sharing a GPU: how a one-GPU box exercises the group)."""
        super().__init__()
        self.W, self.H, self.Wp, self.Hp = int(W), int(H), int(pano_width), int(pano_height)
        lut = _c(lut, np.float64).reshape(-1)
        if lut.size != self.W * self.H * 3:
            raise ValueError("lut must hold W*H*3 doubles")
        if devices is None:
            self._ck(self._L.cmx_backend_create(C.byref(self._ctx), int(device), self.W, self.H, _dp(lut), self.Wp, self.Hp))
        else:
            dv = (C.c_int * len(devices))(*[int(d) for d in devices])
            self._ck(self._L.cmx_backend_create_group(C.byref(self._ctx), dv, len(devices), self.W, self.H, _dp(lut), self.Wp,
                                                      self.Hp, int(transport)))
        self.K = self.num_fixed = 0

    def group_info(self):
        """{'members', 'devices', 'transport', 'events_per_member', 'last_fanout_us'} of this handle (a plain context: 1 member)."""
        n, tr, us = C.c_int(), C.c_int(), C.c_double()
        dev = (C.c_int * 16)()
        ev = np.zeros(16, np.int64)
        self._ck(self._L.cmx_group_info(self._ctx, C.byref(n), dev, 16, C.byref(tr), ev.ctypes.data_as(c_i64p), C.byref(us)))
        return {"members": n.value, "devices": list(dev[:n.value]), "transport": tr.value,
                "events_per_member": [int(v) for v in ev[:n.value]], "last_fanout_us": us.value}

    def group_transport_info(self):
        """{'chosen', 'measured', 'us_direct', 'us_rccl'}: cmx_group_transport_info (CMX_GROUP_AUTO times the candidates at creation)."""
        ch, me, ud, ur = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        self._ck(self._L.cmx_group_transport_info(self._ctx, C.byref(ch), C.byref(me), C.byref(ud), C.byref(ur)))
        return {"chosen": ch.value, "measured": bool(me.value), "us_direct": ud.value, "us_rccl": ur.value}

    def get_pose_table(self):
        """cmx_backend_get_pose_table: (R[nb,3,3] fp64, Jcp[nb,3,3*order] fp32, idx[nb], t_batch_ns[nb]) at the last evaluation's
        parameters -- what Trajectory::evaluate returns per batch (value, ddrot_ddrot_cp, idx_cp_beg)."""
        nb = C.c_int()
        self._ck(self._L.cmx_backend_get_pose_table(self._ctx, 0, None, None, None, None, C.byref(nb)))
        n = nb.value
        R = np.zeros((max(n, 1), 9))
        J = np.zeros((max(n, 1), 36), np.float32)
        idx = np.zeros(max(n, 1), np.int32)
        t = np.zeros(max(n, 1), np.int64)
        self._ck(self._L.cmx_backend_get_pose_table(self._ctx, n, _dp(R), J.ctypes.data_as(c_fp), idx.ctypes.data_as(C.POINTER(C.c_int)),
                                                    t.ctypes.data_as(c_i64p), C.byref(nb)))
        order = self._order
        return (R[:n].reshape(n, 3, 3), J[:n, :9 * order].reshape(n, 3, 3 * order), idx[:n].copy(), t[:n].copy())

    @property
    def num_params(self):
        return 3 * (self.K - self.num_fixed)

    def set_window(self, x, y, t_ns, order, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns,
                   event_batch_size=100, event_sample_rate=1, blur_sigma=1.0, contrast_measure=VARIANCE, IG=None):
        x, y, t = _c(x, np.uint16), _c(y, np.uint16), _c(t_ns, np.int64)
        if not (len(x) == len(y) == len(t)):
            raise ValueError("x, y, t_ns must have equal length")
        k = _c(knots_xyzw, np.float64).reshape(-1, 4)
        ig = None
        keep = isinstance(IG, str) and IG == "resident"  # CMX_KEEP_MAP: use the device-resident global map
        if IG is not None and not keep:
            ig = _c(IG, np.float32)
            if ig.size != self.Wp * self.Hp:
                raise ValueError("IG must be Hp x Wp")
        self._ck(self._L.cmx_backend_set_window(
            self._ctx, len(x), x.ctypes.data_as(c_u16p), y.ctypes.data_as(c_u16p), t.ctypes.data_as(c_i64p),
            int(order), k.shape[0], _dp(k), int(start_ns), int(dt_ns), int(num_fixed), int(t_next_win_beg_ns),
            int(event_batch_size), int(event_sample_rate), float(blur_sigma), int(contrast_measure),
            C.cast(C.c_void_p(1), c_fp) if keep else (ig.ctypes.data_as(c_fp) if ig is not None else None)))
        self.K, self.num_fixed, self._order = k.shape[0], int(num_fixed), int(order)

    def set_window_aos(self, events, order, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns, event_batch_size=100,
                       event_sample_rate=1, blur_sigma=1.0, contrast_measure=VARIANCE, IG=None):
        """cmx_backend_set_window_aos: `events` = structured array of records with fields x, y, sec, nsec."""
        ev = np.ascontiguousarray(events)
        lay = _lib.aos_layout_of(ev)
        k = _c(knots_xyzw, np.float64).reshape(-1, 4)
        keep = isinstance(IG, str) and IG == "resident"
        ig = _c(IG, np.float32) if (IG is not None and not keep) else None
        self._ck(self._L.cmx_backend_set_window_aos(
            self._ctx, len(ev), ev.ctypes.data_as(C.c_void_p), C.byref(lay), int(order), k.shape[0], _dp(k), int(start_ns), int(dt_ns),
            int(num_fixed), int(t_next_win_beg_ns), int(event_batch_size), int(event_sample_rate), float(blur_sigma), int(contrast_measure),
            C.cast(C.c_void_p(1), c_fp) if keep else (ig.ctypes.data_as(c_fp) if ig is not None else None)))
        self.K, self.num_fixed, self._order = k.shape[0], int(num_fixed), int(order)

    def set_window_from(self, store, first, count, order, knots_xyzw, start_ns, dt_ns, num_fixed, t_next_win_beg_ns,
                        event_batch_size=100, event_sample_rate=1, blur_sigma=1.0, contrast_measure=VARIANCE, IG=None):
        """The window events_[first, first+count) cut from a device-resident EventStore."""
        k = _c(knots_xyzw, np.float64).reshape(-1, 4)
        keep = isinstance(IG, str) and IG == "resident"
        ig = _c(IG, np.float32) if (IG is not None and not keep) else None
        self._ck(self._L.cmx_backend_set_window_from(
            self._ctx, store._h, int(first), int(count), int(order), k.shape[0], _dp(k), int(start_ns), int(dt_ns),
            int(num_fixed), int(t_next_win_beg_ns), int(event_batch_size), int(event_sample_rate), float(blur_sigma),
            int(contrast_measure),
            C.cast(C.c_void_p(1), c_fp) if keep else (ig.ctypes.data_as(c_fp) if ig is not None else None)))
        self.K, self.num_fixed, self._order = k.shape[0], int(num_fixed), int(order)

    def prepare(self, drotv_hint=None):
        """cmx_backend_prepare: pose table at the hint, tile sort, chunk table and bearing streams of the window, queued now."""
        h = None if drotv_hint is None else np.ascontiguousarray(drotv_hint, dtype=np.float64)
        self._ck(self._L.cmx_backend_prepare(self._ctx, _dp(h) if h is not None else None))

    def eval(self, drotv, want_grad=True):
        P = self.num_params
        if np.size(drotv) != P:
            raise ValueError("drotv must hold 3*(K-num_fixed) doubles")
        self._io_buffers(P)
        if P:
            self._xin[:P] = np.reshape(drotv, -1)
        rc = self._L.cmx_backend_eval(self._ctx, self._xp, self._cref, self._gp if want_grad else None)
        if rc:
            self._ck(rc)
        return self._cb.value, (self._gout[:P].copy() if want_grad else None)

    def accumulate(self, drotv, want_grad=True):
        d = _c(drotv, np.float64).reshape(-1)
        self._ck(self._L.cmx_backend_accumulate(self._ctx, _dp(d), int(bool(want_grad))))

    def finish(self, want_grad=True):
        c = C.c_double()
        g = np.zeros(max(self.num_params, 1))
        self._ck(self._L.cmx_backend_finish(self._ctx, C.byref(c), _dp(g) if want_grad else None))
        return c.value, (g[:self.num_params] if want_grad else None)

    def get_plane(self, which):
        out = np.empty((self.Hp, self.Wp), np.float32)
        self._ck(self._L.cmx_backend_get_plane(self._ctx, int(which), out.ctypes.data_as(c_fp)))
        return out

    @property
    def alpha(self):
        a = C.c_double()
        self._ck(self._L.cmx_backend_get_alpha(self._ctx, C.byref(a)))
        return a.value

    # --- global-map upkeep (EventWarper::updateIG / setUpdateTimesIG / resetIG), device-resident
    def updateIG(self, max_update_times):
        self._ck(self._L.cmx_backend_update_map(self._ctx, int(max_update_times)))

    def setUpdateTimesIG(self, quat_xyzw, radius=3):
        q = _c(quat_xyzw, np.float64)
        self._ck(self._L.cmx_backend_mark_visited(self._ctx, _dp(q), int(radius)))

    def resetIG(self):
        self._ck(self._L.cmx_backend_reset_map(self._ctx))

    def getIG(self, with_visits=False):
        ig = np.empty((self.Hp, self.Wp), np.float32)
        v = np.empty((self.Hp, self.Wp), np.uint8) if with_visits else None
        self._ck(self._L.cmx_backend_get_map(self._ctx, ig.ctypes.data_as(c_fp),
                                             v.ctypes.data_as(C.POINTER(C.c_uint8)) if with_visits else None))
        return (ig, v) if with_visits else ig

    def setIG(self, IG=None, visits=None):
        ig = _c(IG, np.float32) if IG is not None else None
        v = _c(visits, np.uint8) if visits is not None else None
        self._ck(self._L.cmx_backend_set_map(self._ctx, ig.ctypes.data_as(c_fp) if ig is not None else None,
                                             v.ctypes.data_as(C.POINTER(C.c_uint8)) if v is not None else None))

    # --- reference-named entry points
    def computeImageOfWarpedEvents(self, drotv, want_deriv=False):
        """iwe (blurred I = IL + alpha*IGp) [, list of blurred derivative planes] at the updated trajectory."""
        self.accumulate(drotv, want_deriv)
        iwe = self.get_plane(_lib.PLANE_IWE)
        if not want_deriv:
            return iwe
        return iwe, np.stack([self.get_plane(_lib.PLANE_DERIV0 + j) for j in range(self.num_params)])

    def contrast_fdf(self, v):
        c, g = self.eval(v, True)
        return -c, -g

    def contrast_f(self, v):
        return -self.eval(v, False)[0]

    def contrast_df(self, v):
        return -self.eval(v, True)[1]

    def setupProblemAndOptimize(self, drotv0=None):
        """FR-CG solve of the window (global_optim_contrast_gsl.cpp:15-145); the reference starts at 0.
        Returns (optimal incremental rotation vectors, report dict)."""
        x = np.zeros(self.num_params) if drotv0 is None else np.array(drotv0, dtype=np.float64, order="C", copy=True)
        rep = _lib.SolveReport()
        self._ck(self._L.cmx_backend_solve(self._ctx, int(self.num_params), _dp(x), C.byref(rep)))
        return x, {f: getattr(rep, f) for f, _ in rep._fields_}
