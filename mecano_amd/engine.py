"""Thin object wrapper over the C-ABI: a device-resident model plus batched rnea / aba / crba calls.

Device inputs are torch tensors on a HIP device (torch is used for device memory and streams only); numpy
inputs go through the host-pointer entry points.  Every compute goes through libmecano_hip.so: nothing here
computes dynamics in Python.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .multibody import ModelDesc


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class HipModel:
    """mh_model_t: the flattened MultiBodySystemReadOnly, uploaded once (replaces the calculators' constructors)."""

    def __init__(self, desc: ModelDesc):
        lib = _lib.load()
        self.desc = desc
        self._arrays = dict(
            parent=_np(desc.parent, np.int32), joint_type=_np(desc.joint_type, np.int32), axis=_np(desc.axis, np.float64),
            X_before=_np(desc.X_before, np.float64), X_com=_np(desc.X_com, np.float64), inertia_J=_np(desc.inertia_J, np.float64),
            inertia_mass=_np(desc.inertia_mass, np.float64), inertia_com=_np(desc.inertia_com, np.float64),
            dof_indices=_np(desc.dof_indices, np.int32), cfg_indices=_np(desc.cfg_indices, np.int32))
        d = _lib.MhModelDesc()
        d.n_joints, d.nq, d.nv = int(desc.n_joints), int(desc.nq), int(desc.nv)
        for k, a in self._arrays.items():
            setattr(d, k, a.ctypes.data_as(ctypes.c_void_p))
        handle = ctypes.c_void_p()
        _lib.check(lib.mh_model_create(ctypes.byref(d), ctypes.byref(handle)))
        self._h = handle
        self._ctx = None       # mh_context_t of a view made by context(); None: the model's default context
        self._parent = None    # the HipModel a context view was made from (kept alive: it owns the handle)
        self.nq, self.nv, self.n_joints = desc.nq, desc.nv, desc.n_joints

    def context(self) -> "HipModel":
        """A view of this model with a context of its own (mh_context_create): the handle is read-only and shared, everything compute
        calls write besides their outputs -- workspace, scratch, staging buffers, hand-off flags, error word -- belongs to the view.  One
        view per host thread / per stream; views and the model itself may then be used at the same time (include/mecano_hip.h, Threading)."""
        import copy
        ctx = ctypes.c_void_p()
        _lib.check(_lib.load().mh_context_create(self._h, ctypes.byref(ctx)))
        view = copy.copy(self)
        view._ctx, view._parent = ctx, (self._parent or self)
        return view

    def check(self, stream=None):
        """mh_model_check: synchronises `stream` and raises what this model's (view's) asynchronous calls left behind on the device."""
        _lib.check(_lib.load().mh_model_check(self._h, self._ctx, stream))

    def close(self):
        if getattr(self, "_ctx", None):
            _lib.load().mh_context_destroy(self._ctx)
            self._ctx = None
            self._h = None  # (a view: the handle belongs to the model it was made from)
            return
        if getattr(self, "_parent", None) is not None:
            return
        if getattr(self, "_h", None):
            _lib.load().mh_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def kernel_variant(self) -> str:
        return _lib.load().mh_model_kernel_variant(self._h).decode()

    @property
    def warnings(self) -> int:
        """MH_WARN_* bits (include/mecano_hip.h): the two model classes in which the engine consciously departs from Mecano."""
        return int(_lib.load().mh_model_warnings(self._h))

    @property
    def warning_text(self) -> str:
        return _lib.load().mh_model_warning_text(self._h).decode()

    def reserve(self, max_batch: int):
        if self._ctx:
            _lib.check(_lib.load().mh_context_reserve(self._ctx, int(max_batch)))
        else:
            _lib.check(_lib.load().mh_reserve(self._h, int(max_batch)))

    # ------------------------------------------------------------------ helpers
    def _options(self, layout, consider_coriolis=True, consider_accelerations=True, stream=None, root_acceleration=None):
        o = _lib.MhOptions()
        o.context = self._ctx
        o.consider_coriolis = int(bool(consider_coriolis))
        o.consider_accelerations = int(bool(consider_accelerations))
        o.layout = int(layout)
        o.use_root_acceleration = 0 if root_acceleration is None else 1
        o.stream = stream
        for k in range(6):
            o.root_acceleration[k] = 0.0 if root_acceleration is None else float(root_acceleration[k])
        return o

    @staticmethod
    def _root(gravity):
        """The `gravity` argument of the compute calls: a 3-vector g (the root body accelerates with (0, -g): setGravity,
        InverseDynamicsCalculator.java:318-348) or a 6-vector, the root's spatial acceleration itself, angular then linear
        (setRootAcceleration, InverseDynamicsCalculator.java:413-427; mh_options.root_acceleration).  Returns (g[3] for the C call, the 6 or None)."""
        a = [float(v) for v in np.asarray(gravity, dtype=np.float64).reshape(-1)]
        if len(a) == 6:
            return (ctypes.c_double * 3)(0.0, 0.0, 0.0), a
        if len(a) != 3:
            raise _lib.MecanoHipError(2, f"gravity must have 3 entries (or 6: a root acceleration), got {len(a)}")
        return (ctypes.c_double * 3)(*a), None

    @staticmethod
    def _is_torch(x):
        return type(x).__module__.startswith("torch")

    def _batch(self, x, n, layout):
        if x.ndim != 2:
            raise _lib.MecanoHipError(2, f"expected a 2-D state matrix, got shape {tuple(x.shape)}")
        rows, B = (x.shape[1], x.shape[0]) if layout == _lib.LAYOUT_AOS else (x.shape[0], x.shape[1])
        if rows != n:
            # the MatrixDimensionException of ForwardDynamicsCalculator.java:522-533
            raise _lib.MecanoHipError(2, f"state matrix has {rows} rows per configuration, the system needs {n}")
        return B

    def _check_f_ext(self, f_ext, B, layout):
        """External wrenches are [B, n_joints, 6] (SoA: [n_joints * 6, B]): a wrong shape would be an out-of-bounds device read."""
        if f_ext is None:
            return
        want = (B, self.n_joints, 6) if layout == _lib.LAYOUT_AOS else (self.n_joints * 6, B)
        if tuple(f_ext.shape) != want and not (layout == _lib.LAYOUT_AOS and tuple(f_ext.shape) == (B, self.n_joints * 6)):
            raise _lib.MecanoHipError(2, f"external wrenches have shape {tuple(f_ext.shape)}, expected {want}")

    def _run(self, kind, q, qd, x3, gravity, f_ext, layout, consider_coriolis, consider_accelerations):
        lib = _lib.load()
        g, ra = self._root(gravity)
        if self._is_torch(q):
            import torch
            dt = q.dtype
            if dt not in (torch.float64, torch.float32):
                raise TypeError("state tensors must be float64 or float32")
            tensors = [q] if kind == "crba" else [q, qd, x3]
            if f_ext is not None:
                tensors.append(f_ext)
            for t in tensors:
                if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                    raise ValueError("device tensors must be contiguous, on the HIP device and of one dtype")
            B = self._batch(q, self.nq, layout)
            if kind != "crba":
                if self._batch(qd, self.nv, layout) != B or self._batch(x3, self.nv, layout) != B:
                    raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
            self._check_f_ext(f_ext, B, layout)
            stream = torch.cuda.current_stream(q.device).cuda_stream
            opts = self._options(layout, consider_coriolis, consider_accelerations, stream, root_acceleration=ra)
            sfx = "f64" if dt == torch.float64 else "f32"
            if kind == "crba":
                shape = (B, self.nv, self.nv) if layout == _lib.LAYOUT_AOS else (self.nv * self.nv, B)
                out = torch.empty(shape, dtype=dt, device=q.device)
                _lib.check(getattr(lib, f"mh_crba_{sfx}")(self._h, B, q.data_ptr(), ctypes.byref(opts), out.data_ptr()))
            else:
                out = torch.empty_like(qd)
                fp = f_ext.data_ptr() if f_ext is not None else None
                _lib.check(getattr(lib, f"mh_{kind}_{sfx}")(self._h, B, q.data_ptr(), qd.data_ptr(), x3.data_ptr(), g, fp, ctypes.byref(opts),
                                                            out.data_ptr()))
            return out
        # numpy: host-pointer entry points (fp64; float32 arrays go to the fp32 ones)
        ndt = np.float32 if getattr(q, "dtype", None) == np.float32 else np.float64
        sfx = "f32" if ndt == np.float32 else "f64"
        q = _np(q, ndt)
        B = self._batch(q, self.nq, layout)
        opts = self._options(layout, consider_coriolis, consider_accelerations, None, root_acceleration=ra)
        if kind == "crba":
            out = np.empty((B, self.nv, self.nv) if layout == _lib.LAYOUT_AOS else (self.nv * self.nv, B), dtype=ndt)
            _lib.check(getattr(lib, f"mh_crba_{sfx}_host")(self._h, B, q.ctypes.data, ctypes.byref(opts), out.ctypes.data))
            return out
        qd, x3 = _np(qd, ndt), _np(x3, ndt)
        if self._batch(qd, self.nv, layout) != B or self._batch(x3, self.nv, layout) != B:
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        f = None if f_ext is None else _np(f_ext, ndt)
        self._check_f_ext(f, B, layout)
        out = np.empty_like(qd)
        _lib.check(getattr(lib, f"mh_{kind}_{sfx}_host")(self._h, B, q.ctypes.data, qd.ctypes.data, x3.ctypes.data, g,
                                                        None if f is None else f.ctypes.data, ctypes.byref(opts), out.ctypes.data))
        return out

    # ------------------------------------------------------------------ the three hot-path calls
    def rnea(self, q, qd, qdd, gravity=(0.0, 0.0, -9.81), f_ext=None, layout=_lib.LAYOUT_AOS, consider_coriolis=True,
             consider_accelerations=True):
        return self._run("rnea", q, qd, qdd, gravity, f_ext, layout, consider_coriolis, consider_accelerations)

    def aba(self, q, qd, tau, gravity=(0.0, 0.0, -9.81), f_ext=None, layout=_lib.LAYOUT_AOS):
        return self._run("aba", q, qd, tau, gravity, f_ext, layout, True, True)

    def set_joint_source_modes(self, modes: Optional[Sequence[int]]):
        """One mode per joint of the description (0 = effort source, 1 = acceleration source); None resets all of them
        (ForwardDynamicsCalculator.java:400-444)."""
        lib = _lib.load()
        if modes is None:
            _lib.check(lib.mh_model_set_joint_source_modes(self._h, None))
            return
        m = _np(modes, np.int32)
        if m.shape != (self.n_joints,):
            raise _lib.MecanoHipError(2, f"expected {self.n_joints} joint source modes, got {m.shape}")
        _lib.check(lib.mh_model_set_joint_source_modes(self._h, m.ctypes.data))

    @property
    def n_acceleration_sources(self) -> int:
        return int(_lib.load().mh_model_n_acceleration_sources(self._h))

    def aba_locked(self, q, qd, tau, qdd_in, gravity=(0.0, 0.0, -9.81), f_ext=None, layout=_lib.LAYOUT_AOS):
        """Forward dynamics with acceleration-source joints: returns (qdd, tau) of all DoFs (fp64).  numpy inputs are moved to
        the current HIP device and the results back."""
        import torch
        lib = _lib.load()
        host = not self._is_torch(q)
        if host:
            dev = torch.device("cuda", torch.cuda.current_device())
            q, qd, tau, qdd_in = [torch.from_numpy(_np(x, np.float64)).to(dev) for x in (q, qd, tau, qdd_in)]
            f_ext = None if f_ext is None else torch.from_numpy(_np(f_ext, np.float64)).to(dev)
        dt = q.dtype
        if dt not in (torch.float64, torch.float32):
            raise TypeError("state tensors must be float64 or float32")
        for t in (q, qd, tau, qdd_in) + ((f_ext,) if f_ext is not None else ()):
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise ValueError("aba_locked needs contiguous tensors of one dtype on the HIP device")
        B = self._batch(q, self.nq, layout)
        if any(self._batch(x, self.nv, layout) != B for x in (qd, tau, qdd_in)):
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        self._check_f_ext(f_ext, B, layout)
        g, ra = self._root(gravity)
        opts = self._options(layout, True, True, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        qdd_out, tau_out = torch.empty_like(qd), torch.empty_like(qd)
        fn = lib.mh_aba_locked_f64 if dt == torch.float64 else lib.mh_aba_locked_f32
        _lib.check(fn(self._h, B, q.data_ptr(), qd.data_ptr(), tau.data_ptr(), qdd_in.data_ptr(), g,
                                         f_ext.data_ptr() if f_ext is not None else None, ctypes.byref(opts), qdd_out.data_ptr(),
                                         tau_out.data_ptr()))
        if host:
            return qdd_out.cpu().numpy(), tau_out.cpu().numpy()
        return qdd_out, tau_out

    def _bodies(self, kind, q, qd, x3, gravity, f_ext, layout, consider_coriolis=True, consider_accelerations=True):
        import torch
        lib = _lib.load()
        dt = q.dtype
        if dt not in (torch.float64, torch.float32) or (dt == torch.float32 and kind not in ("rnea", "aba")):
            raise TypeError("per-body outputs take float64 or float32 tensors, joint wrenches float64")
        for t in (q, qd, x3) + ((f_ext,) if f_ext is not None else ()):
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise ValueError("per-body outputs need contiguous tensors of one dtype on the HIP device")
        B = self._batch(q, self.nq, layout)
        if self._batch(qd, self.nv, layout) != B or self._batch(x3, self.nv, layout) != B:
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        self._check_f_ext(f_ext, B, layout)
        g, ra = self._root(gravity)
        opts = self._options(layout, consider_coriolis, consider_accelerations, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        out = torch.empty_like(qd)
        shape = (B, self.n_joints, 6) if layout == _lib.LAYOUT_AOS else (self.n_joints * 6, B)
        if kind in ("rnea_wrenches", "aba_wrenches"):
            w = torch.empty(shape, dtype=dt, device=q.device)
            fn = lib.mh_rnea_joint_wrenches_f64 if kind == "rnea_wrenches" else lib.mh_aba_joint_wrenches_f64
            _lib.check(fn(self._h, B, q.data_ptr(), qd.data_ptr(), x3.data_ptr(), g, f_ext.data_ptr() if f_ext is not None else None,
                          ctypes.byref(opts), out.data_ptr(), w.data_ptr()))
            return out, w
        acc, tw = torch.empty(shape, dtype=dt, device=q.device), torch.empty(shape, dtype=dt, device=q.device)
        sfx = "f64" if dt == torch.float64 else "f32"
        fn = getattr(lib, f"mh_rnea_bodies_{sfx}" if kind == "rnea" else f"mh_aba_bodies_{sfx}")
        _lib.check(fn(self._h, B, q.data_ptr(), qd.data_ptr(), x3.data_ptr(), g, f_ext.data_ptr() if f_ext is not None else None, ctypes.byref(opts),
                      out.data_ptr(), acc.data_ptr(), tw.data_ptr()))
        return out, acc, tw

    def rnea_bodies(self, q, qd, qdd, gravity=(0.0, 0.0, -9.81), f_ext=None, layout=_lib.LAYOUT_AOS, consider_coriolis=True,
                    consider_accelerations=True):
        """RNEA plus per-body outputs: (tau, body_acc, body_twist), the latter [B, n_joints, 6] in the body-fixed frames."""
        return self._bodies("rnea", q, qd, qdd, gravity, f_ext, layout, consider_coriolis, consider_accelerations)

    def aba_bodies(self, q, qd, tau, gravity=(0.0, 0.0, -9.81), f_ext=None, layout=_lib.LAYOUT_AOS):
        """ABA plus per-body outputs: (qdd, body_acc, body_twist)."""
        return self._bodies("aba", q, qd, tau, gravity, f_ext, layout)

    def rnea_joint_wrenches(self, q, qd, qdd, gravity=(0.0, 0.0, -9.81), f_ext=None, layout=_lib.LAYOUT_AOS, consider_coriolis=True,
                            consider_accelerations=True):
        """RNEA plus InverseDynamicsCalculator.getComputedJointWrench of every joint: (tau, joint_wrench [B, n_joints, 6]), the
        wrenches (moment, force) in the frames after the joints."""
        return self._bodies("rnea_wrenches", q, qd, qdd, gravity, f_ext, layout, consider_coriolis, consider_accelerations)

    def aba_joint_wrenches(self, q, qd, tau, gravity=(0.0, 0.0, -9.81), f_ext=None, layout=_lib.LAYOUT_AOS):
        """ABA plus ForwardDynamicsCalculator.getJointWrench of every joint: (qdd, joint_wrench)."""
        return self._bodies("aba_wrenches", q, qd, tau, gravity, f_ext, layout)

    def relative_acceleration(self, q, body_acc, body_twist, base_joints, body_joints, gravity=(0.0, 0.0, -9.81), layout=_lib.LAYOUT_AOS,
                              consider_velocities=True):
        """RigidBodyAccelerationProvider.getRelativeAcceleration for pairs of bodies (indices of listed joints, -1 = the root body) from
        the per-body outputs of rnea_bodies / aba_bodies on the same configurations: [B, n_pairs, 6] in the body's body-fixed frame."""
        import torch
        lib = _lib.load()
        tensors = (q, body_acc) + ((body_twist,) if body_twist is not None else ())
        for t in tensors:
            if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("relative_acceleration needs contiguous float64 tensors on the HIP device")
        B = self._batch(q, self.nq, layout)
        for t in tensors[1:]:
            self._check_f_ext(t, B, layout)
        base, body = _np(base_joints, np.int32).reshape(-1), _np(body_joints, np.int32).reshape(-1)
        if base.shape != body.shape:
            raise _lib.MecanoHipError(2, "base and body index lists differ in length")
        n_pairs = int(base.shape[0])
        g, ra = self._root(gravity)
        opts = self._options(layout, consider_velocities, True, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        out = torch.empty((B, n_pairs, 6) if layout == _lib.LAYOUT_AOS else (n_pairs * 6, B), dtype=torch.float64, device=q.device)
        _lib.check(lib.mh_relative_acceleration_f64(self._h, B, q.data_ptr(), body_acc.data_ptr(),
                                                    body_twist.data_ptr() if body_twist is not None else None, g, n_pairs, base.ctypes.data,
                                                    body.ctypes.data, ctypes.byref(opts), out.data_ptr()))
        return out

    def integrate(self, dt, q, qd, qdd, layout=_lib.LAYOUT_AOS, out=None, return_acceleration=False):
        """One step of MultiBodySystemStateIntegrator.doubleIntegrateFromAcceleration on device tensors (fp64 / fp32).  ``out`` =
        (q_out, qd_out[, qdd_out]) may name the inputs themselves for an in-place step; by default new tensors are returned."""
        import torch
        lib = _lib.load()
        dt_ = q.dtype
        if dt_ not in (torch.float64, torch.float32):
            raise TypeError("state tensors must be float64 or float32")
        for t in (q, qd, qdd):
            if not t.is_cuda or t.dtype != dt_ or not t.is_contiguous():
                raise ValueError("integrate needs contiguous tensors of one dtype on the HIP device")
        B = self._batch(q, self.nq, layout)
        if self._batch(qd, self.nv, layout) != B or self._batch(qdd, self.nv, layout) != B:
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        if out is None:
            # entries no joint owns are passed through unchanged
            out = (q.clone(), qd.clone()) + ((qdd.clone(),) if return_acceleration else ())
        q_out, qd_out = out[0], out[1]
        qdd_out = out[2] if len(out) > 2 else None
        opts = self._options(layout, True, True, torch.cuda.current_stream(q.device).cuda_stream)
        fn = lib.mh_integrate_f64 if dt_ == torch.float64 else lib.mh_integrate_f32
        _lib.check(fn(self._h, B, float(dt), q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), ctypes.byref(opts), q_out.data_ptr(), qd_out.data_ptr(),
                      qdd_out.data_ptr() if qdd_out is not None else None))
        return out

    def step(self, dt, q, qd, tau, gravity=(0.0, 0.0, -9.81), f_ext=None, inplace=False):
        """One simulation step (forward dynamics + state integration, mh_aba_integrate_f64): returns (q_next, qd_next, qdd).  AoS fp64
        device tensors; ``inplace=True`` overwrites q and qd."""
        import torch
        lib = _lib.load()
        for t in (q, qd, tau) + ((f_ext,) if f_ext is not None else ()):
            if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("step needs contiguous float64 tensors on the HIP device")
        B = self._batch(q, self.nq, _lib.LAYOUT_AOS)
        if self._batch(qd, self.nv, _lib.LAYOUT_AOS) != B or self._batch(tau, self.nv, _lib.LAYOUT_AOS) != B:
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        self._check_f_ext(f_ext, B, _lib.LAYOUT_AOS)
        g, ra = self._root(gravity)
        opts = self._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        qn, vn = (q, qd) if inplace else (q.clone(), qd.clone())
        qdd = torch.empty_like(qd)
        _lib.check(lib.mh_aba_integrate_f64(self._h, B, float(dt), q.data_ptr(), qd.data_ptr(), tau.data_ptr(), g,
                                            f_ext.data_ptr() if f_ext is not None else None, ctypes.byref(opts), qdd.data_ptr(), qn.data_ptr(),
                                            vn.data_ptr()))
        return qn, vn, qdd

    def rnea_aba(self, q, qd, qdd, tau, gravity=(0.0, 0.0, -9.81), f_ext=None, out=None, layout=_lib.LAYOUT_AOS):
        """tau_out = RNEA(q, qd, qdd) and qdd_out = ABA(q, qd, tau) in one call.  Device tensors: mh_rnea_aba_f64 / mh_rnea_aba_f32 (``layout``:
        AoS [B, n] or SoA [n, B]); numpy arrays (fp64, AoS): the pipelined host-pointer entry point mh_rnea_aba_f64_host (``out`` = (tau_out,
        qdd_out) arrays to write into, e.g. pinned ones)."""
        lib = _lib.load()
        if not self._is_torch(q):
            q, qd, qdd, tau = (_np(x, np.float64) for x in (q, qd, qdd, tau))
            B = self._batch(q, self.nq, _lib.LAYOUT_AOS)
            if any(self._batch(x, self.nv, _lib.LAYOUT_AOS) != B for x in (qd, qdd, tau)):
                raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
            f = None if f_ext is None else _np(f_ext, np.float64)
            self._check_f_ext(f, B, _lib.LAYOUT_AOS)
            tau_out, qdd_out = out if out is not None else (np.empty_like(qd), np.empty_like(qd))
            g, ra = self._root(gravity)
            opts = self._options(_lib.LAYOUT_AOS, root_acceleration=ra)
            _lib.check(lib.mh_rnea_aba_f64_host(self._h, B, q.ctypes.data, qd.ctypes.data, qdd.ctypes.data, tau.ctypes.data, g,
                                                None if f is None else f.ctypes.data, ctypes.byref(opts), tau_out.ctypes.data, qdd_out.ctypes.data))
            return tau_out, qdd_out
        import torch
        for t in (q, qd, qdd, tau) + ((f_ext,) if f_ext is not None else ()):
            if not t.is_cuda or t.dtype not in (torch.float64, torch.float32) or t.dtype != q.dtype or not t.is_contiguous():
                raise ValueError("rnea_aba needs contiguous float64 (or float32) tensors of one dtype on the HIP device")
        B = self._batch(q, self.nq, layout)
        if any(self._batch(x, self.nv, layout) != B for x in (qd, qdd, tau)):
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        g, ra = self._root(gravity)
        opts = self._options(layout, True, True, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        tau_out, qdd_out = torch.empty_like(qd), torch.empty_like(qd)
        fn = lib.mh_rnea_aba_f64 if q.dtype == torch.float64 else lib.mh_rnea_aba_f32
        _lib.check(fn(self._h, B, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), tau.data_ptr(), g,
                      f_ext.data_ptr() if f_ext is not None else None, ctypes.byref(opts), tau_out.data_ptr(), qdd_out.data_ptr()))
        return tau_out, qdd_out

    def rnea_crba(self, q, qd, qdd, gravity=(0.0, 0.0, -9.81), f_ext=None):
        """tau = RNEA(q, qd, qdd) and H = CRBA(q) in one call (mh_rnea_crba_f64; fp64, AoS, device tensors): (tau [B, nv], H [B, nv, nv])."""
        import torch
        lib = _lib.load()
        for t in (q, qd, qdd) + ((f_ext,) if f_ext is not None else ()):
            if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("rnea_crba needs contiguous float64 tensors on the HIP device")
        B = self._batch(q, self.nq, _lib.LAYOUT_AOS)
        if any(self._batch(x, self.nv, _lib.LAYOUT_AOS) != B for x in (qd, qdd)):
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        self._check_f_ext(f_ext, B, _lib.LAYOUT_AOS)
        g, ra = self._root(gravity)
        opts = self._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        tau_out = torch.empty_like(qd)
        H = torch.empty((B, self.nv, self.nv), dtype=torch.float64, device=q.device)
        _lib.check(lib.mh_rnea_crba_f64(self._h, B, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), g, f_ext.data_ptr() if f_ext is not None else None,
                                        ctypes.byref(opts), tau_out.data_ptr(), H.data_ptr()))
        return tau_out, H

    def bind_rnea_aba(self, q, qd, qdd, tau, tau_out, qdd_out, gravity=(0.0, 0.0, -9.81), f_ext=None):
        """Returns a zero-argument callable that issues mh_rnea_aba_f64 (float32 tensors: mh_rnea_aba_f32) on the given (caller-owned,
        device-resident) buffers.
        All argument marshalling is done once here: a steady-state caller (a simulation loop, bench.py) pays one C call per step."""
        import torch
        lib = _lib.load()
        tensors = (q, qd, qdd, tau, tau_out, qdd_out) + ((f_ext,) if f_ext is not None else ())
        for t in tensors:
            if not t.is_cuda or t.dtype not in (torch.float64, torch.float32) or t.dtype != q.dtype or not t.is_contiguous():
                raise ValueError("bind_rnea_aba needs contiguous float64 (or float32) tensors of one dtype on the HIP device")
        B = self._batch(q, self.nq, _lib.LAYOUT_AOS)
        if any(self._batch(x, self.nv, _lib.LAYOUT_AOS) != B for x in (qd, qdd, tau, tau_out, qdd_out)):
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        g, ra = self._root(gravity)
        opts = self._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        args = (self._h, ctypes.c_int64(B), ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(qd.data_ptr()), ctypes.c_void_p(qdd.data_ptr()),
                ctypes.c_void_p(tau.data_ptr()), g, ctypes.c_void_p(f_ext.data_ptr()) if f_ext is not None else None, ctypes.byref(opts),
                ctypes.c_void_p(tau_out.data_ptr()), ctypes.c_void_p(qdd_out.data_ptr()))
        fn = lib.mh_rnea_aba_f64 if q.dtype == torch.float64 else lib.mh_rnea_aba_f32
        keep = (tensors, g, opts)

        def call(_fn=fn, _args=args, _keep=keep):
            st = _fn(*_args)
            if st:
                _lib.check(st)

        return call

    def bind_rnea_crba(self, q, qd, qdd, tau_out, H_out, gravity=(0.0, 0.0, -9.81), f_ext=None):
        """bind_rnea_aba's counterpart for mh_rnea_crba_f64: a zero-argument callable on caller-owned device buffers (H_out is [B, nv, nv])."""
        import torch
        lib = _lib.load()
        tensors = (q, qd, qdd, tau_out, H_out) + ((f_ext,) if f_ext is not None else ())
        for t in tensors:
            if not t.is_cuda or t.dtype != torch.float64 or not t.is_contiguous():
                raise ValueError("bind_rnea_crba needs contiguous float64 tensors on the HIP device")
        B = self._batch(q, self.nq, _lib.LAYOUT_AOS)
        if any(self._batch(x, self.nv, _lib.LAYOUT_AOS) != B for x in (qd, qdd, tau_out)) or tuple(H_out.shape) != (B, self.nv, self.nv):
            raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        self._check_f_ext(f_ext, B, _lib.LAYOUT_AOS)
        g, ra = self._root(gravity)
        opts = self._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream(q.device).cuda_stream, root_acceleration=ra)
        args = (self._h, ctypes.c_int64(B), ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(qd.data_ptr()), ctypes.c_void_p(qdd.data_ptr()), g,
                ctypes.c_void_p(f_ext.data_ptr()) if f_ext is not None else None, ctypes.byref(opts), ctypes.c_void_p(tau_out.data_ptr()),
                ctypes.c_void_p(H_out.data_ptr()))
        fn = lib.mh_rnea_crba_f64
        keep = (tensors, g, opts)

        def call(_fn=fn, _args=args, _keep=keep):
            st = _fn(*_args)
            if st:
                _lib.check(st)

        return call

    def crba(self, q, layout=_lib.LAYOUT_AOS):
        return self._run("crba", q, None, None, (0.0, 0.0, 0.0), None, layout, True, True)


    # ------------------------------------------------------------------ Coriolis matrix, centroidal momentum
    def _device_inputs(self, tensors, layout):
        import torch
        dt = tensors[0].dtype
        if dt not in (torch.float64, torch.float32):
            raise TypeError("state tensors must be float64 or float32")
        for t in tensors:
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise ValueError("device tensors must be contiguous, on the HIP device and of one dtype")
        B = self._batch(tensors[0], self.nq, layout)
        for t in tensors[1:]:
            if self._batch(t, self.nv, layout) != B:
                raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
        return B, dt, ("f64" if dt == torch.float64 else "f32"), torch.cuda.current_stream(tensors[0].device).cuda_stream

    def crba_coriolis(self, q, qd, layout=_lib.LAYOUT_AOS):
        """Mass matrix and Coriolis matrix (CompositeRigidBodyMassMatrixCalculator with the Coriolis calculation enabled,
        CompositeRigidBodyMassMatrixCalculator.java:271-274, 344-365): (H, C), device tensors [B, nv, nv]."""
        if not self._is_torch(q):  # numpy: host-pointer entry point (fp64)
            q, qd = _np(q, np.float64), _np(qd, np.float64)
            B = self._batch(q, self.nq, layout)
            if self._batch(qd, self.nv, layout) != B:
                raise _lib.MecanoHipError(2, "batch sizes of the state matrices differ")
            shape = (B, self.nv, self.nv) if layout == _lib.LAYOUT_AOS else (self.nv * self.nv, B)
            H, C = np.empty(shape), np.empty(shape)
            opts = self._options(layout)
            _lib.check(_lib.load().mh_crba_coriolis_f64_host(self._h, B, q.ctypes.data, qd.ctypes.data, ctypes.byref(opts), H.ctypes.data,
                                                             C.ctypes.data))
            return H, C
        import torch
        B, dt, sfx, stream = self._device_inputs([q, qd], layout)
        shape = (B, self.nv, self.nv) if layout == _lib.LAYOUT_AOS else (self.nv * self.nv, B)
        H, C = torch.empty(shape, dtype=dt, device=q.device), torch.empty(shape, dtype=dt, device=q.device)
        opts = self._options(layout, stream=stream)
        _lib.check(getattr(_lib.load(), f"mh_crba_coriolis_{sfx}")(self._h, B, q.data_ptr(), qd.data_ptr(), ctypes.byref(opts), H.data_ptr(),
                                                                  C.data_ptr()))
        return H, C

    def regressor(self, q, qd, qdd, gravity=(0.0, 0.0, -9.81), layout=_lib.LAYOUT_AOS, consider_coriolis=True, consider_accelerations=True,
                  first_moment_columns=False):
        """Joint torque regressor (JointTorqueRegressorCalculator.compute, JointTorqueRegressorCalculator.java:173-190): Y [B, nv, 10 n_joints]
        with tau = Y pi; the ten columns of joint j's successor body start at column 10 j.  Device tensors (fp64 / fp32); ``layout`` is the
        layout of q, qd, qdd and Y (SoA: Y [nv, 10 n_joints, B]).  ``first_moment_columns``: d tau / d (m c) in columns 1..3 of every body instead of the reference's zeros."""
        import torch
        B, dt, sfx, stream = self._device_inputs([q, qd, qdd], layout)
        Y = torch.empty((B, self.nv, 10 * self.n_joints) if layout == _lib.LAYOUT_AOS else (self.nv, 10 * self.n_joints, B), dtype=dt, device=q.device)
        g, ra = self._root(gravity)
        opts = self._options(layout, consider_coriolis, consider_accelerations, stream, root_acceleration=ra)
        _lib.check(getattr(_lib.load(), f"mh_regressor_{sfx}")(self._h, B, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), g, ctypes.byref(opts),
                                                              1 if first_moment_columns else 0, Y.data_ptr()))
        return Y

    def centroidal(self, q, qd=None, frame=None, at_com=False, layout=_lib.LAYOUT_AOS):
        """Centroidal momentum matrix, convective term and frame origin (CompositeRigidBodyMassMatrixCalculator.java:375-420, 801-839):
        (A [B, 6, nv], b [B, 6] or None without qd, com [B, 3]).  ``frame``: 12 numbers (R row-major, p), pose of the centroidal momentum
        frame in the root body frame, None = the root body frame; ``at_com`` re-centres it on the centre of mass."""
        fr = None
        if frame is not None:
            fr = (ctypes.c_double * 12)(*[float(v) for v in np.asarray(frame, dtype=np.float64).reshape(12)])
        mode = _lib.CENTROIDAL_FRAME_AT_COM if at_com else _lib.CENTROIDAL_FRAME_FIXED
        if not self._is_torch(q):  # numpy: host-pointer entry point (fp64)
            q = _np(q, np.float64)
            qd = None if qd is None else _np(qd, np.float64)
            B = self._batch(q, self.nq, layout)
            aos = layout == _lib.LAYOUT_AOS
            A = np.empty((B, 6, self.nv) if aos else (6 * self.nv, B))
            b = None if qd is None else np.empty((B, 6) if aos else (6, B))
            com = np.empty((B, 3) if aos else (3, B))
            opts = self._options(layout)
            _lib.check(_lib.load().mh_centroidal_f64_host(self._h, B, q.ctypes.data, None if qd is None else qd.ctypes.data, fr, mode,
                                                          ctypes.byref(opts), A.ctypes.data, None if b is None else b.ctypes.data,
                                                          com.ctypes.data))
            return A, b, com
        import torch
        B, dt, sfx, stream = self._device_inputs([q] if qd is None else [q, qd], layout)
        aos = layout == _lib.LAYOUT_AOS
        A = torch.empty((B, 6, self.nv) if aos else (6 * self.nv, B), dtype=dt, device=q.device)
        b = None if qd is None else torch.empty((B, 6) if aos else (6, B), dtype=dt, device=q.device)
        com = torch.empty((B, 3) if aos else (3, B), dtype=dt, device=q.device)
        opts = self._options(layout, stream=stream)
        _lib.check(getattr(_lib.load(), f"mh_centroidal_{sfx}")(self._h, B, q.data_ptr(), None if qd is None else qd.data_ptr(), fr, mode,
                                                               ctypes.byref(opts), A.data_ptr(), None if b is None else b.data_ptr(),
                                                               com.data_ptr()))
        return A, b, com


def pinned_empty(shape, dtype=np.float64):
    """A numpy array in PINNED host memory (mh_host_alloc): the host-pointer entry points copy from / to it at PCIe rate, where pageable
    arrays go through the runtime's staging copies.  The memory is released when the array (and every view of it) is gone."""
    import weakref
    lib = _lib.load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    ptr = ctypes.c_void_p()
    _lib.check(lib.mh_host_alloc(max(1, n * dt.itemsize), ctypes.byref(ptr)))
    buf = (ctypes.c_char * max(1, n * dt.itemsize)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    weakref.finalize(buf, lib.mh_host_free, ctypes.c_void_p(ptr.value))
    return arr


class HipTimer:
    """HIP events recorded on the stream the kernels are launched on (bench.py, SURVEY.md section 8d timing protocol)."""

    def __init__(self):
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().mh_timer_create(ctypes.byref(self._h)))

    def start(self, stream=None):
        _lib.check(_lib.load().mh_timer_start(self._h, stream))

    def stop(self, stream=None):
        _lib.check(_lib.load().mh_timer_stop(self._h, stream))

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        _lib.check(_lib.load().mh_timer_elapsed_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self._h:
                _lib.load().mh_timer_destroy(self._h)
                self._h = None
        except Exception:
            pass


class HipCommunicator:
    """The C-ABI's RCCL communicator (mh_comm_*, include/mecano_hip.h): what a host without torch.distributed uses around the sharded
    compute calls -- shard_range, one broadcast of the robot description, one all-gather of the output rows.  mecano_amd/distributed.py
    does the same over torch.distributed; bench.py keeps that transport (the driver launches it under torch.distributed.run)."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(HipCommunicator.ID_BYTES)
        _lib.check(_lib.load().mh_comm_unique_id(buf))
        return buf.raw

    @staticmethod
    def shard_range(B: int, rank: int, world: int):
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.load().mh_shard_range(B, rank, world, ctypes.byref(lo), ctypes.byref(hi)))
        return int(lo.value), int(hi.value)

    def __init__(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != self.ID_BYTES:
            raise ValueError(f"a communicator id has {self.ID_BYTES} bytes")
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().mh_comm_create(ctypes.create_string_buffer(unique_id, self.ID_BYTES), rank, world, ctypes.byref(self._h)))
        r, w = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(_lib.load().mh_comm_size(self._h, ctypes.byref(r), ctypes.byref(w)))
        self.rank, self.world = int(r.value), int(w.value)

    def broadcast_bytes(self, payload: Optional[bytes], nbytes: int, root: int = 0) -> bytes:
        """`root` passes the payload, the others None; everybody returns the same nbytes bytes."""
        buf = ctypes.create_string_buffer(payload if self.rank == root else b"", nbytes)
        _lib.check(_lib.load().mh_comm_broadcast_host(self._h, buf, nbytes, root))
        return buf.raw

    def broadcast_model_desc(self, desc, root: int = 0):
        """The robot description from `root` (the others pass None): header of two lengths, then the packed integer and real arrays."""
        from .distributed import pack_desc, unpack_desc
        if self.rank == root:
            ints, f64 = pack_desc(desc)
            head = np.array([len(ints), len(f64)], dtype=np.int64).tobytes()
        else:
            ints = f64 = None
            head = None
        ni, nf = (int(x) for x in np.frombuffer(self.broadcast_bytes(head, 16, root), dtype=np.int64))
        raw = self.broadcast_bytes(ints.tobytes() + f64.tobytes() if self.rank == root else None, 8 * (ni + nf), root)
        return unpack_desc(np.frombuffer(raw[:8 * ni], dtype=np.int64).copy(), np.frombuffer(raw[8 * ni:], dtype=np.float64).copy())

    def all_gather_rows(self, local, B_total: int, stream=None):
        """local: this rank's rows of shard_range(B_total, rank, world), a contiguous CUDA tensor [rows, ...]; returns [B_total, ...]."""
        import torch
        lo, hi = self.shard_range(B_total, self.rank, self.world)
        if local.shape[0] != hi - lo or not local.is_contiguous():
            raise ValueError(f"rank {self.rank} owns rows [{lo}, {hi}) of {B_total}: pass exactly those, contiguous")
        out = torch.empty((B_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        row_bytes = local.element_size() * int(np.prod(local.shape[1:], dtype=np.int64))
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.load().mh_comm_all_gather_rows(self._h, local.data_ptr(), B_total, row_bytes, out.data_ptr(), s))
        return out

    def barrier(self, stream=None):
        _lib.check(_lib.load().mh_comm_barrier(self._h, stream))

    def close(self):
        if self._h:
            _lib.check(_lib.load().mh_comm_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
