"""Thin Python owner of one ``scg_handle`` (C ABI, ``include/scg_rollout.h``).

PyTorch-ROCm tensors are only the device buffers here: their ``data_ptr()`` and the current HIP
stream cross the ABI; all arithmetic happens in the HIP kernels.
"""
import ctypes as C

import numpy as np
import torch

from safe_control_gym_amd import _capi


class Engine:
    def __init__(self, cfg, x_goal_table=None, constraint_list=None, device=None, disturbances=None):
        """cfg: a filled ``_capi.ScgConfig``; x_goal_table: (T, nx) float64 or None."""
        self.lib = _capi.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _capi.ScgError('the rollout engine runs on a ROCm device only (got %r); there is no CPU path'
                                 % (device,))
        if not torch.cuda.is_available():
            raise _capi.ScgError('no ROCm device is visible to this process: the rollout engine cannot run '
                                 '(there is no CPU path)')
        self.cfg = cfg
        self.nx, self.nu = {_capi.SCG_QUAD_1D: (2, 1), _capi.SCG_QUAD_2D: (6, 2), _capi.SCG_QUAD_3D: (12, 4),
                            _capi.SCG_CARTPOLE: (4, 1)}[cfg.system]
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_create(C.byref(cfg), C.byref(handle)), 'scg_create')
            self.handle = handle
            if x_goal_table is not None:
                tab = np.ascontiguousarray(x_goal_table, dtype=np.float64)
                _capi.check(self.lib.scg_set_reference(handle, tab.ctypes.data_as(C.POINTER(C.c_double)),
                                                       tab.shape[0]), 'scg_set_reference')
            self.num_device_rows = 0
            if constraint_list is not None and constraint_list.num_device_rows > 0:
                rows, n_rows, pmats, n_p = constraint_list.compile_rows()
                _capi.check(self.lib.scg_set_constraints(handle, rows, n_rows, pmats, n_p), 'scg_set_constraints')
                self.num_device_rows = n_rows
            if disturbances:
                arr = (_capi.ScgDisturbance * len(disturbances))(*disturbances)
                _capi.check(self.lib.scg_set_disturbances(handle, arr, len(disturbances)), 'scg_set_disturbances')
        self.obs_dim = self.lib.scg_obs_dim(handle)
        self._io = _capi.ScgStepIO()
        self._io.struct_size = C.sizeof(_capi.ScgStepIO)
        self._io_ref = C.byref(self._io)

    def set_serial_word(self, word):
        """``scg_step_io.serial_word`` of every later step / rollout of this engine: a one-element int32 device tensor whose value
        the kernels add to the step's serial (a caller's own HIP graph of the step loop), or None."""
        self._serial_word = word
        self._io.serial_word = None if word is None else word.data_ptr()

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.scg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)

    def _stream(self):
        """The current HIP stream of this engine's device as a raw handle.  torch.cuda.current_stream() costs ~4 us of
        Python per call (device-index normalisation, lazy-init checks); the raw accessor is what it ends up calling."""
        if Engine._raw_stream is not None:
            idx = self.device.index
            return C.c_void_p(Engine._raw_stream(torch.cuda.current_device() if idx is None else idx))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def reset(self, bufs, mask, reset_state, serial, n, ld):
        p = _capi.ptr
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_reset(self.handle, p(bufs['state']), p(bufs.get('state_lo')),
                                           p(bufs['step_count']), p(bufs.get('ep_return')),
                                           p(bufs.get('ep_length')), p(bufs.get('obs')), p(mask),
                                           p(reset_state), p(bufs.get('inertial')), p(bufs.get('episode_id')),
                                           serial & 0xffffffff, ld, n, self._stream()),
                        'scg_reset')

    _IO_FIELDS = ('action', 'action_disturbance', 'dynamics_force', 'obs_noise', 'reset_state', 'inertial', 'state',
                  'state_lo', 'state_out', 'step_count', 'ep_return', 'ep_length', 'episode_id', 'obs', 'reward', 'done',
                  'constraint_values', 'constraint_violation', 'truncated', 'goal_reached', 'mse',
                  'terminal_record', 'stats')

    def _bind(self, bufs, ld):
        io = self._io
        io.ld = ld
        rec = bufs.get('terminal_record')
        # row-major records [ld, stride], or SoA planes [obs_dim + 2, ld] (terminal_stride 0)
        io.terminal_stride = rec.shape[1] if (rec is not None and rec.shape[0] == ld) else 0
        lo = bufs.get('state_lo')
        # [6, ld]: attitude-only residuals, as float32 (kind 1) or int16 (kind 2) planes
        io.state_lo_kind = 0 if (lo is None or lo.shape[0] == self.nx) else (2 if lo.dtype == torch.int16 else 1)
        for name in self._IO_FIELDS:
            t = bufs.get(name)
            setattr(io, name, None if t is None else t.data_ptr())
        return io

    def step(self, bufs, n, ld, serial, action_rows=False, asynchronous=False):
        io = self._bind(bufs, ld)
        io.reset_serial = serial & 0xffffffff
        io.action_rows = int(action_rows)
        fn = self.lib.scg_step_async if asynchronous else self.lib.scg_step
        with torch.cuda.device(self.device):
            _capi.check(fn(self.handle, C.byref(io), n, self._stream()), 'scg_step_async' if asynchronous else 'scg_step')

    def wait(self):
        """``scg_step_wait``: the current stream waits (on the device) for this handle's pending asynchronous step."""
        if Engine._raw_stream is not None and torch.cuda.current_device() == getattr(self, '_dev_index', -1):
            status = self.lib.scg_step_wait(self.handle, Engine._raw_stream(self._dev_index))
        else:
            with torch.cuda.device(self.device):
                status = self.lib.scg_step_wait(self.handle, self._stream())
        if status != 0:
            _capi.check(status, 'scg_step_wait')

    def queue_stream(self):
        """The handle's own launch queue (``scg_step_queue``) as a torch stream: run a shard's policy on it and its
        ``step_async`` / ``step_wait`` need no cross-stream event."""
        if getattr(self, '_queue_stream', None) is None:
            q = C.c_void_p()
            with torch.cuda.device(self.device):
                _capi.check(self.lib.scg_step_queue(self.handle, C.byref(q)), 'scg_step_queue')
            self._queue_stream = torch.cuda.ExternalStream(q.value, device=self.device)
        return self._queue_stream

    def bind_static(self, bufs, ld):
        """Bind the long-lived buffers once; ``step_bound`` then only swaps the action pointer and the serial."""
        self._bind(bufs, ld)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

    def step_bound(self, action, n, serial, action_rows=False, outs=None, state_out=None, asynchronous=False):
        """Hot path of env.step(): ~4 ctypes stores and one foreign call.  ``outs`` = (obs planes or None, reward, done)
        re-points this step's outputs at tensors of the caller's (copy_outputs: fresh ones every step, written by the kernel
        itself); ``state_out`` = the buffer the new state goes to, which becomes the bound ``state`` of the next call.
        ``asynchronous``: ``scg_step_async`` -- the kernel goes to the handle's own queue behind an event of the current stream."""
        io = self._io
        if outs is not None:
            io.obs = None if outs[0] is None else outs[0].data_ptr()
            io.reward, io.done = outs[1].data_ptr(), outs[2].data_ptr()
        if state_out is not None:
            io.state_out = state_out.data_ptr()
        io.action = action.data_ptr()
        io.action_rows = int(action_rows)
        io.reset_serial = serial & 0xffffffff
        fn = self.lib.scg_step_async if asynchronous else self.lib.scg_step
        if Engine._raw_stream is not None and torch.cuda.current_device() == self._dev_index:
            status = fn(self.handle, self._io_ref, n, Engine._raw_stream(self._dev_index))
        else:
            with torch.cuda.device(self.device):
                status = fn(self.handle, self._io_ref, n, self._stream())
        if status != 0:
            io.state_out = None
            _capi.check(status, 'scg_step_async' if asynchronous else 'scg_step')
        if state_out is not None:
            io.state, io.state_out = io.state_out, None

    def restore_outputs(self, reward, done):
        """After a ``step_bound(outs=...)`` that pointed this step's reward / done at a caller's tensors: bind the env's own planes
        again (two stores instead of a full re-bind)."""
        self._io.reward, self._io.done = reward.data_ptr(), done.data_ptr()

    def rollout(self, bufs, n, ld, serial, steps, actions, fused=False, sink=None):
        """``steps`` launches (or one fused launch); ``actions`` is a contiguous [period, nu, ld] float32 tensor
        used cyclically.  ``sink``: {name: time-major tensor} -- step k writes that output into slab k of the tensor
        (``scg_rollout_trajectory``)."""
        io = self._bind(bufs, ld)
        io.action = actions.data_ptr()
        io.action_rows = 0
        io.reset_serial = serial & 0xffffffff
        period, stride = actions.shape[0], actions.shape[1] * actions.shape[2]
        with torch.cuda.device(self.device):
            if sink is not None:
                sk = _capi.ScgTrajectorySink()
                sk.struct_size = C.sizeof(_capi.ScgTrajectorySink)
                for name, t in sink.items():
                    setattr(sk, name, t.data_ptr())
                    setattr(sk, name + '_step_stride', t.stride(0))
                _capi.check(self.lib.scg_rollout_trajectory(self.handle, C.byref(io), C.byref(sk), n, steps, stride, period,
                                                            int(bool(fused)), self._stream()), 'scg_rollout_trajectory')
                return
            fn = self.lib.scg_rollout_fused if fused else self.lib.scg_rollout
            _capi.check(fn(self.handle, C.byref(io), n, steps, stride, period, self._stream()), 'scg_rollout')

    def eval_constraints(self, env, action_soa):
        """[N, num_device_rows] values of this engine's constraint table on env's state."""
        n, ld = env.num_envs, env._ld
        out = torch.empty((max(self.num_device_rows, 1), ld), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_eval_constraints(self.handle, _capi.ptr(env._bufs['state']),
                                                      _capi.ptr(action_soa), _capi.ptr(out), None, ld, n,
                                                      self._stream()), 'scg_eval_constraints')
        return out[:self.num_device_rows, :n].T

    def model_eval(self, state_soa, input_soa, inertial_soa, n, ld, want_xdot, want_xnext):
        """Batched fc_func / fd_func (symbolic_systems.py:68-73): ([nx, ld] xdot or None, [nx, ld] xnext or None)."""
        mk = lambda: torch.empty((self.nx, ld), dtype=torch.float32, device=self.device)
        xdot, xnext = (mk() if want_xdot else None), (mk() if want_xnext else None)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_model_eval(self.handle, _capi.ptr(state_soa), _capi.ptr(input_soa),
                                                _capi.ptr(inertial_soa), _capi.ptr(xdot), _capi.ptr(xnext), ld, n,
                                                self._stream()), 'scg_model_eval')
        return xdot, xnext

    def model_flow(self, state_soa, input_soa, inertial_soa, n, ld, dt, rtol, atol, want_sens=False, max_doublings=12):
        """fd_func as the reference defines it (symbolic_systems.py:70-73), on float64 SoA planes: the flow over [0, dt] to the
        stated tolerance -> (xf [nx, ld], dxf_dx0 [n, nx, nx] | None, dxf_dp [n, nx, nu] | None, intervals [n] int32)."""
        f64 = torch.float64
        for t in (state_soa, input_soa, inertial_soa):
            if t is not None and t.dtype != f64:
                raise TypeError('[ERROR] model_flow works on float64 planes.')
        xf = torch.empty((self.nx, ld), dtype=f64, device=self.device)
        sx = torch.empty((n, self.nx, self.nx), dtype=f64, device=self.device) if want_sens else None
        su = torch.empty((n, self.nx, self.nu), dtype=f64, device=self.device) if want_sens else None
        intervals = torch.empty(n, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_model_flow(self.handle, _capi.ptr(state_soa), _capi.ptr(input_soa), _capi.ptr(inertial_soa),
                                                float(dt), float(rtol), float(atol), int(max_doublings), _capi.ptr(xf),
                                                _capi.ptr(sx), _capi.ptr(su), _capi.ptr(intervals), ld, n, self._stream()),
                        'scg_model_flow')
        return xf, sx, su, intervals

    def linearize(self, state_soa, input_soa, inertial_soa, n, ld, disc_dt=0.0):
        """Batched df_func (symbolic_systems.py:77-92): ([n, nx, nx], [n, nx, nu]) for SoA planes [.][ld]."""
        A = torch.empty((n, self.nx, self.nx), dtype=torch.float32, device=self.device)
        B = torch.empty((n, self.nx, self.nu), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_linearize(self.handle, _capi.ptr(state_soa), _capi.ptr(input_soa),
                                               _capi.ptr(inertial_soa), _capi.ptr(A), _capi.ptr(B), float(disc_dt),
                                               ld, n, self._stream()), 'scg_linearize')
        return A, B

    def fc_linear(self, x_eval_soa, u_eval_soa, state_soa, input_soa, inertial_soa, n, ld):
        """Batched fc_linear_func (symbolic_systems.py:93-101) on SoA planes -> [nx, ld] planes; x_eval_soa None = at 0."""
        out = torch.empty((self.nx, ld), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_fc_linear(self.handle, _capi.ptr(x_eval_soa), _capi.ptr(u_eval_soa), _capi.ptr(state_soa),
                                               _capi.ptr(input_soa), _capi.ptr(inertial_soa), _capi.ptr(out), ld, n,
                                               self._stream()), 'scg_fc_linear')
        return out

    def quadratic_loss(self, state_soa, input_soa, xr, ur, Q, R, n, ld, want_grad=True):
        """Batched loss / l_x / l_u (symbolic_systems.py:94-123).  xr: [nx] vector or [nx, ld'] planes; ur likewise."""
        Qh = (C.c_double * (self.nx * self.nx))(*[float(v) for v in Q.reshape(-1)])
        Rh = (C.c_double * (self.nu * self.nu))(*[float(v) for v in R.reshape(-1)])
        l = torch.empty(ld, dtype=torch.float32, device=self.device)
        l_x = torch.empty((self.nx, ld), dtype=torch.float32, device=self.device) if want_grad else None
        l_u = torch.empty((self.nu, ld), dtype=torch.float32, device=self.device) if want_grad else None
        xr_ld = xr.shape[1] if xr.ndim == 2 else 0
        ur_ld = ur.shape[1] if ur.ndim == 2 else 0
        with torch.cuda.device(self.device):
            _capi.check(self.lib.scg_quadratic_loss(self.handle, Qh, Rh, _capi.ptr(state_soa), _capi.ptr(input_soa),
                                                    _capi.ptr(xr), xr_ld, _capi.ptr(ur), ur_ld, _capi.ptr(l),
                                                    _capi.ptr(l_x), _capi.ptr(l_u), ld, n, self._stream()),
                        'scg_quadratic_loss')
        return l, l_x, l_u

    def rollout_graph_info(self):
        """(signatures held as instantiated hipGraphs [-1: capture failed, plain launches], graph launches so far)."""
        c, r = C.c_int32(), C.c_int64()
        _capi.check(self.lib.scg_rollout_graph_info(self.handle, C.byref(c), C.byref(r)), 'scg_rollout_graph_info')
        return c.value, r.value

    def kernel_info(self, n):
        b, g, l = C.c_int32(), C.c_int32(), C.c_int32()
        _capi.check(self.lib.scg_kernel_info(self.handle, C.byref(b), C.byref(g), C.byref(l), n), 'scg_kernel_info')
        return dict(block=b.value, grid=g.value, lds_bytes=l.value)
