"""Batched drop-ins for Mecano's three calculators, with the reference's method names.

* ``InverseDynamicsCalculator``              algorithms/InverseDynamicsCalculator.java:186-251, 291-306, 318-403, 444-501, 567
* ``ForwardDynamicsCalculator``              algorithms/ForwardDynamicsCalculator.java:112-196, 234-319, 348-381, 475-520, 556-567
* ``CompositeRigidBodyMassMatrixCalculator`` algorithms/CompositeRigidBodyMassMatrixCalculator.java:157-233, 286-303, 344-348
* ``JointTorqueRegressorCalculator``         algorithms/JointTorqueRegressorCalculator.java:101-133, 173-190, 360-502 (a caller of the first)
* ``MultiBodyResponseCalculator``            algorithms/MultiBodyResponseCalculator.java:120-140, 288-935 (a caller of the second)

Differences a user of the reference must know (all forced by batching, none changes results):

* Mecano's calculators read q and qd from the joints' reference frames, which the caller refreshes with
  ``rootBody.updateFramesRecursively()``.  Here the state of B configurations is passed explicitly:
  ``compute(q, qd, qdd)`` with matrices shaped [B, nq] / [B, nv] indexed by the system's
  JointMatrixIndexProvider.  The engine does its own forward kinematics on the GPU.
* results are returned as (and kept in) a batched matrix: ``getJointTauMatrix()`` is [B, nv].
* ``CompositeRigidBodyMassMatrixCalculator`` has no result cache to ``reset()``; ``reset()`` is a no-op kept for
  source compatibility.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib
from .engine import HipModel
from .multibody import MultiBodySystem, RigidBody


def _as_system(inp) -> MultiBodySystem:
    if isinstance(inp, MultiBodySystem):
        return inp
    if isinstance(inp, RigidBody):  # InverseDynamicsCalculator(RigidBodyReadOnly) :186-199
        return MultiBodySystem.toMultiBodySystemInput(inp)
    raise TypeError("expected a MultiBodySystem or a RigidBody")


class _Base:
    def __init__(self, input, considerIgnoredSubtreesInertia: bool = True):
        self.input = _as_system(input)
        # ignored subtrees: their inertia is lumped into the body they hang from (InverseDynamicsCalculator.java:226-236, 832-860)
        self.model = HipModel(self.input.toModelDesc(considerIgnoredSubtreesInertia=bool(considerIgnoredSubtreesInertia and self.input.getJointsToIgnore())))
        self._gravity = np.zeros(3)
        self._f_ext = None
        self.layout = _lib.LAYOUT_AOS

    def getInput(self) -> MultiBodySystem:
        return self.input

    # ---- gravity: InverseDynamicsCalculator.java:318-403 / ForwardDynamicsCalculator.java:234-319
    def setGravitationalAcceleration(self, *g):
        """``setGravitationalAcceleration(gz)`` (gravity along z, usually negative), ``(gx, gy, gz)`` or a 3-vector."""
        if len(g) == 1 and np.ndim(g[0]) == 0:
            self._gravity = np.array([0.0, 0.0, float(g[0])])
        elif len(g) == 1:
            self._gravity = np.asarray(g[0], dtype=np.float64).reshape(3).copy()
        elif len(g) == 3:
            self._gravity = np.array([float(g[0]), float(g[1]), float(g[2])])
        else:
            raise ValueError("gravity is a scalar (z) or three components")

    def setRootAcceleration(self, rootAcceleration):
        """``setRootAcceleration(SpatialAccelerationReadOnly)`` (InverseDynamicsCalculator.java:413-427, ForwardDynamicsCalculator.java:330-343):
        the spatial acceleration of the root body, (angular x y z, linear x y z) in root-body coordinates -- what a rotating / accelerating
        base contributes; it replaces whatever setGravitationalAcceleration stored (gravity g is the root acceleration (0, -g), :343-348).
        Three entries are taken as the linear part alone."""
        a = np.asarray(rootAcceleration, dtype=np.float64).reshape(-1)
        if a.size == 3:
            a = np.concatenate([np.zeros(3), a])
        if a.size != 6:
            raise ValueError("a root acceleration has six components (angular, linear)")
        self._gravity = a.copy()  # six entries: passed on as mh_options.root_acceleration

    # ---- the ONE-configuration face: the reference's own signatures (compute(), compute(DMatrix), setExternalWrench(body, wrench),
    # writeComputed...) on top of the batched path with B = 1.  State is read from the joints in the order of the index provider
    # (MultiBodySystemTools.extractJointsState, tools/MultiBodySystemTools.java:1433-1491), exactly what the reference does implicitly
    # through the joints' reference frames.
    def _joints_in_order(self):
        return self.input.getJointMatrixIndexProvider().getIndexedJointsInOrder()

    def _extract(self, which, size):
        from .multibody import MultiBodySystemTools
        m = np.zeros((size, 1))
        MultiBodySystemTools.extractJointsState(self._joints_in_order(), which, m)
        return m.reshape(1, size)

    def _column(self, matrix, size, what):
        m = np.asarray(matrix, dtype=np.float64)
        if m.size != size or (m.ndim == 2 and m.shape[1] != 1):  # MatrixDimensionException in the reference (ForwardDynamicsCalculator.java:522-533)
            raise _lib.MecanoHipError(2, f"{what}: expected {size} x 1, got {tuple(m.shape)}")
        return m.reshape(1, size)

    def getExternalWrench(self, rigidBody):
        """``getExternalWrench(rigidBody)`` (InverseDynamicsCalculator.java:444-461, ForwardDynamicsCalculator.java:353-369): the live
        external wrench of that body for the one-configuration calls -- six numbers (moment, force) expressed in the body-fixed frame,
        where the reference keeps it (setMatchingFrame); modify it in place."""
        w = self.__dict__.setdefault("_single_wrenches", {})
        joints = self._joints_in_order()
        if id(rigidBody) not in {id(j.getSuccessor()) for j in joints}:
            raise KeyError("the rigid-body is not a successor of a joint this calculator considers")
        return w.setdefault(id(rigidBody), np.zeros(6))

    def setExternalWrench(self, rigidBody, externalWrench):
        """``setExternalWrench(rigidBody, externalWrench)`` (:463-472 / :371-381); the wrench is given in the body-fixed frame."""
        self.getExternalWrench(rigidBody)[:] = np.asarray(externalWrench, dtype=np.float64).reshape(6)

    def _single_f_ext(self):
        w = self.__dict__.get("_single_wrenches") or {}
        if not any(np.any(v != 0.0) for v in w.values()):
            return None
        joints = self._joints_in_order()
        return np.stack([w.get(id(j.getSuccessor()), np.zeros(6)) for j in joints])[None, :, :]

    def _rows_of(self, joint, column):
        if id(joint) not in {id(j) for j in self._joints_in_order()}:
            return None
        rows = self.input.getJointMatrixIndexProvider().getJointDoFIndices(joint)
        return np.asarray(column).reshape(-1, 1)[rows]

    # ---- external wrenches: InverseDynamicsCalculator.java:444-472 / ForwardDynamicsCalculator.java:348-381
    def setExternalWrenches(self, f_ext):
        """[B, n_joints, 6] (moment, force) per successor body, in its body-fixed frame; joints in index-provider order."""
        self._f_ext = f_ext

    def setExternalWrenchesToZero(self):
        self._f_ext = None
        self.__dict__.pop("_single_wrenches", None)


class RigidBodyAccelerationProvider:
    """algorithms/interfaces/RigidBodyAccelerationProvider.java:137-260, batched: the spatial acceleration (and twist) of every
    successor body relative to the inertial frame, expressed in its body-fixed frame, as computed by the last ``compute``."""

    def __init__(self, system: MultiBodySystem, owner=None):
        self._pos = {id(j.getSuccessor()): k for k, j in enumerate(system.getJointsToConsider())}
        self._root = system.getRootBody()
        self._owner = owner  # the calculator whose last compute() filled this provider (model, q, gravity, switches)
        self.body_acc = None
        self.body_twist = None

    def getAccelerationOfBody(self, body):
        """[B, 6] (angular, linear); ``None`` for a body the calculator does not consider, like the reference (:242-246).  The root
        body's own acceleration is the root acceleration -g and is not stored per configuration."""
        k = self._pos.get(id(body))
        return None if k is None or self.body_acc is None else self.body_acc[:, k, :]

    def getTwistOfBody(self, body):
        k = self._pos.get(id(body))
        return None if k is None or self.body_twist is None else self.body_twist[:, k, :]

    def _index(self, body):
        if body is self._root:
            return -1
        return self._pos.get(id(body))

    def getRelativeAcceleration(self, base, body):
        """RigidBodyAccelerationProvider.java:66, 199-235: acceleration of ``body`` with respect to ``base``, expressed in the body-fixed
        frame of ``body``, [B, 6]; ``None`` when the calculator does not consider one of the two (:210-216).  ``base`` / ``body`` may
        also be sequences of bodies: [B, n_pairs, 6]."""
        many = isinstance(base, (list, tuple))
        bases, bodies = (list(base), list(body)) if many else ([base], [body])
        idx1, idx2 = [self._index(b) for b in bases], [self._index(b) for b in bodies]
        if any(i is None for i in idx1 + idx2) or self.body_acc is None or self._owner is None:
            return None
        o = self._owner
        out = o.model.relative_acceleration(o._last_q, self.body_acc, self.body_twist, idx1, idx2, o._gravity, o.layout,
                                            consider_velocities=getattr(o, "_coriolis", True))
        return out if many else out[:, 0, :]


class InverseDynamicsCalculator(_Base):
    def __init__(self, input, considerIgnoredSubtreesInertia: bool = True):
        super().__init__(input, considerIgnoredSubtreesInertia)
        self._coriolis = True
        self._accel = True
        self._tau = None
        self._wrenches = None
        self._last_q = None
        self._provider = RigidBodyAccelerationProvider(self.input, self)
        self._joint_pos = {id(j): k for k, j in enumerate(self.input.getJointsToConsider())}

    def setConsiderCoriolisAndCentrifugalForces(self, flag: bool):
        self._coriolis = bool(flag)

    def setConsiderJointAccelerations(self, flag: bool):
        self._accel = bool(flag)

    def compute(self, *args, bodies: bool = False, wrenches: bool = False):
        """Batched: ``compute(q, qd, qdd)``; ``bodies=True`` also fills the acceleration provider (InverseDynamicsCalculator.java:242-250,
        660-663), ``wrenches=True`` keeps the 6-D wrench of every joint for getComputedJointWrench (:578-585).
        The reference's own signatures: ``compute()`` (:481-484: configuration, velocity and desired acceleration read from the joints) and
        ``compute(jointAccelerationMatrix)`` (:496-501: nv x 1) evaluate ONE configuration through the same HIP path; afterwards
        getJointTauMatrix() is nv x 1 and writeComputedJointWrenches(joints) stores the efforts in the joints."""
        if len(args) <= 1:
            from .multibody import JointStateType
            nq, nv = self.model.nq, self.model.nv
            q, qd = self._extract(JointStateType.CONFIGURATION, nq), self._extract(JointStateType.VELOCITY, nv)
            qdd = self._extract(JointStateType.ACCELERATION, nv) if not args or args[0] is None else self._column(args[0], nv, "jointAccelerationMatrix")
            keep, self._f_ext = self._f_ext, self._single_f_ext()
            try:
                self.compute(q, qd, qdd, bodies=bodies, wrenches=wrenches)
            finally:
                self._f_ext = keep
            self._tau = np.asarray(self._tau).reshape(nv, 1)
            self._single = True
            return self._tau
        q, qd, qdd = args
        self._single = False
        self._last_q, self._wrenches = q, None
        if wrenches:
            self._tau, self._wrenches = self.model.rnea_joint_wrenches(q, qd, qdd, self._gravity, self._f_ext, self.layout, self._coriolis,
                                                                       self._accel)
            if bodies:
                _, self._provider.body_acc, self._provider.body_twist = self.model.rnea_bodies(
                    q, qd, qdd, self._gravity, self._f_ext, self.layout, self._coriolis, self._accel)
        elif bodies:
            self._tau, self._provider.body_acc, self._provider.body_twist = self.model.rnea_bodies(
                q, qd, qdd, self._gravity, self._f_ext, self.layout, self._coriolis, self._accel)
        else:
            self._tau = self.model.rnea(q, qd, qdd, self._gravity, self._f_ext, self.layout, self._coriolis, self._accel)
        return self._tau

    def getAccelerationProvider(self) -> RigidBodyAccelerationProvider:
        return self._provider

    def getJointTauMatrix(self):
        return self._tau

    def getComputedJointWrench(self, joint):
        """InverseDynamicsCalculator.java:578-585: [B, 6] (moment, force) in the frame after the joint, ``None`` for a joint this
        calculator does not consider; needs ``compute(..., wrenches=True)``."""
        k = joint if isinstance(joint, (int, np.integer)) else self._joint_pos.get(id(joint))
        if k is None:
            return None
        if self._wrenches is None:
            raise ValueError("call compute(q, qd, qdd, wrenches=True) first")
        return self._wrenches[:, int(k), :]

    def writeComputedJointWrench(self, joint) -> bool:
        """``writeComputedJointWrench(joint)`` (:639-653): joint.setJointTau(0, getComputedJointTau(joint)); False for a joint this
        calculator does not consider.  After a one-configuration compute."""
        if not getattr(self, "_single", False):
            raise ValueError("compute() or compute(jointAccelerationMatrix) first: a batch cannot be written into one joint")
        tau = self._rows_of(joint, self._tau)
        if tau is None:
            return False
        joint.setJointTau(0, tau)
        return True

    def writeComputedJointWrenches(self, joints):
        """``writeComputedJointWrenches(JointBasics[] | List)`` (:613-629)."""
        for joint in joints:
            self.writeComputedJointWrench(joint)

    def getComputedJointTau(self, joint):
        """:587-602: the rows of the joint in the tau matrix"""
        if getattr(self, "_single", False):
            return self._rows_of(joint, self._tau)  # N x 1, like the reference's
        if id(joint) not in self._joint_pos or self._tau is None:
            return None
        rows = self.input.getJointMatrixIndexProvider().getJointDoFIndices(joint)
        return self._tau[:, rows]


class JointTorqueRegressorCalculator(_Base):
    """algorithms/JointTorqueRegressorCalculator.java:101-133, 173-190, 360-502: tau = Y(q, qd, qdd) pi for the inverse dynamics without
    external wrenches, ten parameters per body (SpatialInertiaBasisOption, :514-516).  One kernel evaluates every column of every
    configuration (the reference: one second pass of the inverse dynamics per body and parameter).  Bodies are ordered like the system's
    joints to consider (the reference: iteration order of a HashMap, :85, :318-348 -- use the per-body accessors when porting).  The
    parameter vector is read once at construction, as the reference does (:130, :337-348).  numpy in -> numpy out; device tensors stay
    on the device."""

    PARAMETERS_PER_BODY = 10
    BASES = ("M", "MCOM_X", "MCOM_Y", "MCOM_Z", "I_XX", "I_XY", "I_XZ", "I_YY", "I_YZ", "I_ZZ")

    def __init__(self, input, firstMomentColumns: bool = False):
        super().__init__(input, considerIgnoredSubtreesInertia=False)
        self._coriolis, self._accel, self._first = True, True, bool(firstMomentColumns)
        self._Y = None
        self._joint_pos = {id(j): k for k, j in enumerate(self.input.getJointsToConsider())}
        d = self.model.desc
        n = d.n_joints
        J = np.asarray(d.inertia_J, dtype=np.float64).reshape(n, 3, 3)
        pi = np.zeros((n, 10))
        pi[:, 0] = np.asarray(d.inertia_mass, dtype=np.float64)
        pi[:, 1:4] = np.asarray(d.inertia_com, dtype=np.float64).reshape(n, 3) * (pi[:, 0:1] if self._first else 1.0)  # :877-889
        for c, (a, b) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
            pi[:, 4 + c] = J[:, a, b]
        self._pi = pi.reshape(-1)

    def setConsiderCoriolisAndCentrifugalForces(self, flag: bool):
        self._coriolis = bool(flag)

    def setConsiderJointAccelerations(self, flag: bool):
        self._accel = bool(flag)

    def compute(self, q, qd, qdd):
        """:173-190 for B configurations: Y [B, nv, 10 n]."""
        if HipModel._is_torch(q):
            self._Y = self.model.regressor(q, qd, qdd, self._gravity, self.layout, self._coriolis, self._accel, self._first)
        else:
            import torch
            tq, tqd, tqdd = (torch.tensor(np.ascontiguousarray(x, dtype=np.float64), device="cuda") for x in (q, qd, qdd))
            self._Y = self.model.regressor(tq, tqd, tqdd, self._gravity, self.layout, self._coriolis, self._accel, self._first).cpu().numpy()
        return self._Y

    def _body_index(self, body) -> int:
        k = self._joint_pos.get(id(body.getParentJoint())) if body is not None and body.getParentJoint() is not None else None
        if k is None:
            raise ValueError("the body is not the successor of a joint this calculator considers")
        return int(k)

    def getJointTorqueRegressorMatrix(self):
        return self._Y

    def getJointTorqueRegressorMatrixBlock(self, body):
        """:462-465: the nv x 10 block of ``body``"""
        k = self._body_index(body)
        return self._Y[:, :, 10 * k:10 * k + 10]

    def getJointTorqueRegressorMatrixSlice(self, body, basis):
        """:475-478: one column; ``basis`` an index 0..9 or a name of BASES"""
        b = self.BASES.index(basis) if isinstance(basis, str) else int(basis)
        return self._Y[:, :, 10 * self._body_index(body) + b]

    def getParameterVector(self):
        return self._pi

    def getParameterVectorSlice(self, body):
        k = self._body_index(body)
        return self._pi[10 * k:10 * k + 10]

    def getParameter(self, body, basis):
        b = self.BASES.index(basis) if isinstance(basis, str) else int(basis)
        return float(self._pi[10 * self._body_index(body) + b])


class JointSourceMode:
    """ForwardDynamicsCalculator.JointSourceMode (ForwardDynamicsCalculator.java:45-57)."""
    EFFORT_SOURCE = 0
    ACCELERATION_SOURCE = 1


class ForwardDynamicsCalculator(_Base):
    JointSourceMode = JointSourceMode

    def __init__(self, input, considerIgnoredSubtreesInertia: bool = True):
        super().__init__(input, considerIgnoredSubtreesInertia)
        self._qdd = None
        self._tau = None
        self._wrenches = None
        self._last_q = None
        self._provider = RigidBodyAccelerationProvider(self.input, self)
        self._modes = [JointSourceMode.EFFORT_SOURCE] * self.model.n_joints
        self._joint_pos = {id(j): k for k, j in enumerate(self.input.getJointsToConsider())}

    # ---- joint source modes: ForwardDynamicsCalculator.java:400-444
    def setJointSourceMode(self, joint, mode: int):
        """``joint`` is a joint of the system (or its position in getJointsToConsider()); ignored joints are rejected like in
        the reference (:407-409)."""
        k = joint if isinstance(joint, (int, np.integer)) else self._joint_pos.get(id(joint))
        if k is None:
            raise ValueError("the joint is not considered by this calculator")
        self._modes[int(k)] = int(mode)
        self.model.set_joint_source_modes(self._modes)

    def setJointSourceModes(self, modeFunction):
        """``modeFunction(joint) -> JointSourceMode`` evaluated on every considered joint (:423-433); a sequence works too."""
        joints = self.input.getJointsToConsider()
        self._modes = [int(modeFunction(j)) for j in joints] if callable(modeFunction) else [int(m) for m in modeFunction]
        self.model.set_joint_source_modes(self._modes)

    def resetJointSourceModes(self):
        self._modes = [JointSourceMode.EFFORT_SOURCE] * self.model.n_joints
        self.model.set_joint_source_modes(None)

    def getAccelerationProvider(self) -> RigidBodyAccelerationProvider:
        """ForwardDynamicsCalculator.java:170-180, 715-718; filled by ``compute(..., bodies=True)``."""
        return self._provider

    def compute(self, *args, bodies: bool = False, wrenches: bool = False):
        """Batched: ``compute(q, qd, tau)`` or, with acceleration-source joints, ``compute(q, qd, tau, qdd)``: tau is read for the
        effort sources, qdd for the acceleration sources.  ``wrenches=True`` keeps the joint wrenches (getJointWrench).
        The reference's own signatures evaluate ONE configuration read from the joints: ``compute()`` (ForwardDynamicsCalculator.java:475-478),
        ``compute(jointTauInput)`` (:489-492) and ``compute(jointTauInput, jointAccelerationInput)`` (:508-520), matrices nv x 1; afterwards
        getJointAccelerationMatrix() / getJointTauMatrix() are nv x 1 and writeComputedJointAccelerations(joints) stores the result."""
        if len(args) <= 2:
            from .multibody import JointStateType
            nq, nv = self.model.nq, self.model.nv
            q, qd = self._extract(JointStateType.CONFIGURATION, nq), self._extract(JointStateType.VELOCITY, nv)
            tau = self._extract(JointStateType.EFFORT, nv) if not args or args[0] is None else self._column(args[0], nv, "jointTauInput")
            locked = any(m == JointSourceMode.ACCELERATION_SOURCE for m in self._modes)
            qdd = None
            if locked:
                qdd = (self._extract(JointStateType.ACCELERATION, nv) if len(args) < 2 or args[1] is None
                       else self._column(args[1], nv, "jointAccelerationInput"))
            keep, self._f_ext = self._f_ext, self._single_f_ext()
            try:
                self.compute(q, qd, tau, qdd, bodies=bodies, wrenches=wrenches)
            finally:
                self._f_ext = keep
            self._qdd, self._tau = np.asarray(self._qdd).reshape(nv, 1), np.asarray(self._tau).reshape(nv, 1)
            self._single = True
            return self._qdd
        q, qd, tau = args[:3]
        qdd = args[3] if len(args) > 3 else None
        self._single = False
        self._last_q, self._wrenches = q, None
        if wrenches and not any(m == JointSourceMode.ACCELERATION_SOURCE for m in self._modes):
            self._qdd, self._wrenches = self.model.aba_joint_wrenches(q, qd, tau, self._gravity, self._f_ext, self.layout)
            self._tau = tau
            if bodies:
                _, self._provider.body_acc, self._provider.body_twist = self.model.aba_bodies(q, qd, tau, self._gravity, self._f_ext, self.layout)
            return self._qdd
        if any(m == JointSourceMode.ACCELERATION_SOURCE for m in self._modes):
            if qdd is None:
                raise ValueError("some joints are acceleration sources: their accelerations (qdd) are needed")
            self._qdd, self._tau = self.model.aba_locked(q, qd, tau, qdd, self._gravity, self._f_ext, self.layout)
        elif bodies:
            self._qdd, self._provider.body_acc, self._provider.body_twist = self.model.aba_bodies(q, qd, tau, self._gravity, self._f_ext, self.layout)
            self._tau = tau
        else:
            self._qdd = self.model.aba(q, qd, tau, self._gravity, self._f_ext, self.layout)
            self._tau = tau
        return self._qdd

    def getJointAccelerationMatrix(self):
        return self._qdd

    def getComputedJointAcceleration(self, joint):
        """``getComputedJointAcceleration(joint)`` (:600-610): N x 1 after a one-configuration compute, None for a joint that is not considered."""
        if not getattr(self, "_single", False):
            raise ValueError("compute(), compute(tau) or compute(tau, qdd) first")
        return self._rows_of(joint, self._qdd)

    def writeComputedJointAcceleration(self, joint) -> bool:
        """``writeComputedJointAcceleration(joint)`` (:699-708)."""
        a = self.getComputedJointAcceleration(joint)
        if a is None:
            return False
        joint.setJointAcceleration(0, a)
        return True

    def writeComputedJointAccelerations(self, joints):
        """``writeComputedJointAccelerations(JointBasics[] | List)`` (:670-688)."""
        for joint in joints:
            self.writeComputedJointAcceleration(joint)

    def getJointTauMatrix(self):
        """Efforts of all joints: the inputs for effort sources, the computed ones for acceleration sources (:556-567)."""
        return self._tau

    def getJointWrench(self, joint):
        """ForwardDynamicsCalculator.java:642-650: the wrench the joint exerts, before projection onto its motion subspace, [B, 6] in
        the frame after the joint; ``None`` for a joint this calculator does not consider; needs ``compute(..., wrenches=True)``."""
        k = joint if isinstance(joint, (int, np.integer)) else self._joint_pos.get(id(joint))
        if k is None:
            return None
        if self._wrenches is None:
            raise ValueError("call compute(q, qd, tau, wrenches=True) first")
        return self._wrenches[:, int(k), :]


class MultiBodyResponseCalculator:
    """algorithms/MultiBodyResponseCalculator.java:120-140, 224-250, 288-600, 608-840, 859-935, batched: the change in joint and body
    accelerations (or, for impulses, velocities) a test wrench on a body or a test effort at a joint produces.  The reference walks the
    disturbance up the articulated-body quantities its forward dynamics left behind (pA+ = -w, u+ = tau+ - S^T pA+, pa+ = pA+ + U D^-1 u+,
    :1206-1252) and the change in acceleration back down (qdd+ = D^-1 (u+ - U^T a+_parent), :1301-1338): that is the forward dynamics'
    own recursion with the velocities, the gravity and the efforts at zero and the test wrench as the only external wrench -- the bias
    acceleration c and the term Ia c vanish with the velocities -- so this mirror asks the ABA kernels for exactly that, for B
    configurations at once (acceleration-source joints keep a zero change, :1230-1238, 1275-1281, through ``mh_aba_locked_f64``).
    The response is linear in the disturbance and independent of q-dot, gravity, efforts and standing external wrenches: the identity the
    reference's tests pin is  qdd(with the test wrench) = qdd + propagateWrench()  (MultiBodyResponseCalculatorTest.java:301-344).

    ``reset(q)`` sets the configurations (the reference reads them from the joints' frames) and forgets the disturbances; test wrenches
    are [B, 6] (moment, force) on the target body, expressed in its body-fixed frame; efforts are [B] or [B, dofs of the joint]."""

    def __init__(self, input):
        self.forwardDynamicsCalculator = input if isinstance(input, ForwardDynamicsCalculator) else ForwardDynamicsCalculator(input)
        fd = self.forwardDynamicsCalculator
        self.input, self.model = fd.input, fd.model
        self._joints = self.input.getJointsToConsider()
        self._body_pos = {id(j.getSuccessor()): k for k, j in enumerate(self._joints)}
        self._joint_pos = {id(j): k for k, j in enumerate(self._joints)}
        self._provider = RigidBodyAccelerationProvider(self.input, None)
        self._q = None
        self._clear()

    def getForwardDynamicsCalculator(self) -> ForwardDynamicsCalculator:
        return self.forwardDynamicsCalculator

    def _clear(self):
        self._wrenches, self._efforts, self._change = None, None, None
        self._provider.body_acc = self._provider.body_twist = None

    def reset(self, q=None):
        """:232-250; ``q`` [B, nq]: the configurations the responses are evaluated at (kept when omitted)."""
        if q is not None:
            self._q = q
        self._clear()

    # ---- array helpers: numpy in -> numpy out, device tensors stay on the device
    def _zeros(self, *shape):
        if HipModel._is_torch(self._q):
            import torch
            return torch.zeros(shape, dtype=self._q.dtype, device=self._q.device)
        return np.zeros(shape)

    def _like(self, x):
        if HipModel._is_torch(self._q):
            import torch
            return x.to(device=self._q.device, dtype=self._q.dtype) if HipModel._is_torch(x) else torch.as_tensor(np.asarray(x), dtype=self._q.dtype, device=self._q.device)
        return np.asarray(x.cpu().numpy() if HipModel._is_torch(x) else x, dtype=np.float64)

    def _batch(self):
        if self._q is None:
            raise ValueError("call reset(q) with the configurations first")
        return int(self._q.shape[0])

    # ---- disturbances (:608-815); several calls accumulate (MultiBodyResponseCalculatorTest.java:749-877)
    def applyRigidBodyWrench(self, target, wrench) -> bool:
        """:608-627; False for a body this calculator does not consider."""
        k = self._body_pos.get(id(target))
        if k is None:
            return False
        B = self._batch()
        if self._wrenches is None:
            self._wrenches = self._zeros(B, self.model.n_joints, 6)
        self._wrenches[:, k, :] += self._like(wrench).reshape(B, 6)
        self._change = None
        return True

    applyRigidBodyImpulse = applyRigidBodyWrench  # :640-659: the same linear map, read as impulse -> change of twist

    def applyJointWrench(self, target, effort) -> bool:
        """:685-735; ``effort`` [B] (1-DoF joints) or [B, dofs]."""
        k = self._joint_pos.get(id(target))
        if k is None:
            return False
        B = self._batch()
        rows = list(self.input.getJointMatrixIndexProvider().getJointDoFIndices(target))
        if self._efforts is None:
            self._efforts = self._zeros(B, self.model.nv)
        self._efforts[:, rows] += self._like(effort).reshape(B, len(rows))
        self._change = None
        return True

    applyJointImpulse = applyJointWrench  # :750-815

    # ---- propagation (:823-840)
    def _propagate(self):
        if self._change is None:
            B = self._batch()
            zero_v = self._zeros(B, self.model.nv)
            tau = self._efforts if self._efforts is not None else zero_v
            g0 = (0.0, 0.0, 0.0)
            fd = self.forwardDynamicsCalculator
            if any(m == JointSourceMode.ACCELERATION_SOURCE for m in fd._modes):
                self._change, _ = self.model.aba_locked(self._q, zero_v, tau, zero_v, g0, self._wrenches, fd.layout)
            else:
                self._change, self._provider.body_acc, _ = self.model.aba_bodies(self._q, zero_v, tau, g0, self._wrenches, fd.layout)
                self._provider.body_twist = self._provider.body_acc
        return self._change

    def propagateWrench(self):
        """:823-829: the change in joint accelerations, [B, nv]"""
        return self._propagate()

    def propagateImpulse(self):
        """:836-842: the change in joint velocities, [B, nv]"""
        return self._propagate()

    def getAccelerationChangeProvider(self) -> RigidBodyAccelerationProvider:
        """:859-862: getAccelerationOfBody(body) = the change in the body's spatial acceleration, in its body-fixed frame"""
        self._propagate()
        return self._provider

    def getTwistChangeProvider(self) -> RigidBodyAccelerationProvider:
        """:873-876: getTwistOfBody(body) = the change in the body's twist after an impulse"""
        self._propagate()
        return self._provider

    def getJointAccelerationChange(self, joint):
        """:887-904: the rows of the joint, [B, dofs]"""
        rows = list(self.input.getJointMatrixIndexProvider().getJointDoFIndices(joint))
        return self._propagate()[:, rows]

    getJointTwistChange = getJointAccelerationChange  # :915-932

    # ---- apparent inertias (:288-600): columns = responses to unit disturbances
    def computeRigidBodyApparentSpatialInertiaInverse(self, target):
        """:288-330 with inertiaFrame = the target's body-fixed frame: [B, 6, 6], change of the body's spatial acceleration per unit
        wrench on it (symmetric); ``None`` for a body this calculator does not consider."""
        k = self._body_pos.get(id(target))
        if k is None:
            return None
        B = self._batch()
        saved = (self._wrenches, self._efforts)
        out = self._zeros(B, 6, 6)
        for c in range(6):
            self._clear()
            w = self._zeros(B, 6)
            w[:, c] = 1.0
            self.applyRigidBodyWrench(target, w)
            self._propagate()
            if self._provider.body_acc is None:
                raise ValueError("apparent inertias need every joint to be an effort source")
            out[:, :, c] = self._provider.body_acc[:, k, :]
        self._clear()
        self._wrenches, self._efforts = saved
        return out

    def computeRigidBodyApparentLinearInertiaInverse(self, target):
        """:449-500 with inertiaFrame = the target's body-fixed frame: [B, 3, 3], linear acceleration of the frame origin per unit force"""
        M = self.computeRigidBodyApparentSpatialInertiaInverse(target)
        return None if M is None else M[:, 3:, 3:]

    def computeJointApparentInertiaInverse(self, target):
        """:512-590: [B, dofs, dofs] (1-DoF joints: [B, 1, 1]), change of the joint's accelerations per unit effort"""
        if self._joint_pos.get(id(target)) is None:
            return None
        B = self._batch()
        rows = list(self.input.getJointMatrixIndexProvider().getJointDoFIndices(target))
        saved = (self._wrenches, self._efforts)
        out = self._zeros(B, len(rows), len(rows))
        for c in range(len(rows)):
            self._clear()
            e = self._zeros(B, len(rows))
            e[:, c] = 1.0
            self.applyJointWrench(target, e)
            out[:, :, c] = self._propagate()[:, rows]
        self._clear()
        self._wrenches, self._efforts = saved
        return out


class CompositeRigidBodyMassMatrixCalculator(_Base):
    """algorithms/CompositeRigidBodyMassMatrixCalculator.java: mass matrix, Coriolis matrix (:271-274, 352-365), centroidal momentum
    matrix and convective term (:367-420), batched.  The reference reads q, qd from the joints; here ``compute(q, qd)`` takes them and
    the getters return what that call produced (``reset()`` forgets it, :282-291)."""

    def __init__(self, input, centroidalMomentumFrame=None, considerIgnoredSubtreesInertia: bool = True):
        super().__init__(input, considerIgnoredSubtreesInertia)
        self._coriolis_enabled = False
        self._frame, self._at_com = None, False
        self.setCentroidalMomentumFrame(centroidalMomentumFrame)
        self.reset()

    def reset(self):
        self._H = self._C = self._A = self._b = self._com = None
        self._q = self._qd = None

    def setEnableCoriolisMatrixCalculation(self, enable: bool):
        """:271-274"""
        self._coriolis_enabled = bool(enable)

    def setCentroidalMomentumFrame(self, frame, atCenterOfMass: bool = False):
        """:375-384.  ``frame``: None (the root body frame, the reference's default :190-193), a RigidBodyTransform-like pose of a
        frame fixed in the root body (12 numbers: R row-major, p; or an object with ``.R`` / ``.p``), and ``atCenterOfMass=True`` to
        get a frames/CenterOfMassReferenceFrame under it."""
        if frame is not None and hasattr(frame, "R"):
            frame = np.concatenate([np.asarray(frame.R, dtype=np.float64).reshape(9), np.asarray(frame.p, dtype=np.float64).reshape(3)])
        self._frame, self._at_com = frame, bool(atCenterOfMass)
        self._A = self._b = self._com = None

    def compute(self, q, qd=None):
        self.reset()
        self._q, self._qd = q, qd
        if self._coriolis_enabled:
            if qd is None:
                raise ValueError("the Coriolis matrix needs the joint velocities")
            self._H, self._C = self.model.crba_coriolis(q, qd, self.layout)
        else:
            self._H = self.model.crba(q, self.layout)
        return self._H

    def getMassMatrix(self, q=None):
        if q is not None:
            return self.compute(q)
        return self._H

    def getCoriolisMatrix(self):
        """:352-365: UnsupportedOperationException when the calculation is disabled"""
        if not self._coriolis_enabled:
            raise NotImplementedError("Coriolis matrix calculation is disabled.")
        return self._C

    def _centroidal(self):
        if self._A is None:
            if self._q is None:
                raise ValueError("call compute(q, qd) first")
            self._A, self._b, self._com = self.model.centroidal(self._q, self._qd, self._frame, self._at_com, self.layout)

    def getCentroidalMomentumMatrix(self):
        """:386-398"""
        self._centroidal()
        return self._A

    def getCentroidalConvectiveTermMatrix(self):
        """:413-420"""
        self._centroidal()
        return self._b

    def getCentroidalConvectiveTerm(self):
        return self.getCentroidalConvectiveTermMatrix()

    def getCenterOfMass(self):
        """origin of a centre-of-mass centroidal frame in its parent frame (algorithms/CenterOfMassCalculator.java:70-91)"""
        self._centroidal()
        return self._com


class MultiBodySystemStateIntegrator:
    """tools/MultiBodySystemStateIntegrator.java:31-75, 365-441: explicit constant-acceleration integration of the joint states, batched.

    The reference walks the joints of one system and updates their state objects in place; here the state of B configurations is
    explicit: ``doubleIntegrateFromAcceleration(input, q, qd, qdd)`` returns the integrated ``(q, qd)`` (device tensors in, device
    tensors out; pass ``inplace=True`` to overwrite the inputs as the reference does)."""

    def __init__(self, dt: float = float("nan")):
        self.dt = float(dt)
        self._models = {}

    def setIntegrationDT(self, dt: float):
        self.dt = float(dt)

    def getIntegrationDT(self) -> float:
        return self.dt

    def _model(self, input) -> HipModel:
        if isinstance(input, HipModel):
            return input
        if isinstance(input, _Base):
            return input.model
        sys_ = _as_system(input)
        if id(sys_) not in self._models:
            self._models[id(sys_)] = (sys_, HipModel(sys_.toModelDesc()))
        return self._models[id(sys_)][1]

    def doubleIntegrateFromAcceleration(self, input, q, qd, qdd, inplace: bool = False, layout=_lib.LAYOUT_AOS):
        """``input``: a MultiBodySystem / RigidBody, one of the calculators of this module, or a HipModel."""
        out = self._model(input).integrate(self.dt, q, qd, qdd, layout, out=(q, qd) if inplace else None)
        return out[0], out[1]
