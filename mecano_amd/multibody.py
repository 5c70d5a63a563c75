"""Host-side mirror of the part of Mecano's multi-body API the three calculators are built from.

The reference is Java; this container (and the GPU box) has no JVM, so the host side above the
C-ABI is mirrored here in Python with the reference's names, argument meaning and error
behaviour, so that the parity tests read like the reference's own tests.  The Java shim a
maintainer would add on the reference side is in ``java/`` and ``INTEGRATION.md``.

Mirrored types (paths relative to /root/reference/src/main/java/us/ihmc/mecano/):

* ``RigidBody``            multiBodySystem/RigidBody.java:25-264
* ``RevoluteJoint``        multiBodySystem/RevoluteJoint.java:23-97
* ``PrismaticJoint``       multiBodySystem/PrismaticJoint.java:20-59
* ``SixDoFJoint``          multiBodySystem/SixDoFJoint.java:26-122
* ``FixedJoint``           multiBodySystem/FixedJoint.java
* ``JointMatrixIndexProvider``  multiBodySystem/interfaces/JointMatrixIndexProvider.java:71-123
* ``MultiBodySystem``      multiBodySystem/interfaces/MultiBodySystemReadOnly.java:26-335

Only *structure* lives here (topology, frames, inertias, index maps).  Joint state is not
stored per joint object as in Mecano: it is the batched matrices handed to ``compute``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import numpy as np

REVOLUTE, PRISMATIC, SIXDOF, FIXED, PLANAR, SPHERICAL = 0, 1, 2, 3, 4, 5


def _as_transform(transform) -> Optional[np.ndarray]:
    """Accepts None, a 4x4, a (R 3x3, p 3) pair or a flat 12-vector; returns flat 12 (R row-major, p)."""
    if transform is None:
        return None
    if isinstance(transform, tuple) and len(transform) == 2:
        R, p = np.asarray(transform[0], dtype=np.float64), np.asarray(transform[1], dtype=np.float64)
        return np.concatenate([R.reshape(9), p.reshape(3)])
    t = np.asarray(transform, dtype=np.float64)
    if t.shape == (4, 4):
        return np.concatenate([t[:3, :3].reshape(9), t[:3, 3]])
    if t.size == 12:
        return t.reshape(12).copy()
    raise ValueError("transform must be None, 4x4, (R, p) or 12 values")


IDENTITY12 = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], dtype=np.float64)


def _skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def _spatial_inertia6(J_origin, mass, com):
    """6x6 [[J, m c~], [m c~^T, m 1]] with J about the frame origin (spatial/interfaces/SpatialInertiaReadOnly.java:394-415)."""
    I = np.zeros((6, 6))
    I[:3, :3] = J_origin
    I[:3, 3:] = mass * _skew(com)
    I[3:, :3] = mass * _skew(com).T
    I[3:, 3:] = mass * np.eye(3)
    return I


def _xf(t12):
    return t12[:9].reshape(3, 3), t12[9:]


def _compose(a, b):
    """a o b for (R, p) pairs: x -> Ra (Rb x + pb) + pa."""
    return a[0] @ b[0], a[0] @ b[1] + a[1]


def _inertia_to_parent(I6, R, p):
    """Re-expresses a 6x6 inertia given in a child frame posed (R, p) in the parent frame: X^-T I X^-1 with the motion transform X."""
    X = np.zeros((6, 6))
    X[:3, :3] = R
    X[3:, 3:] = R
    X[3:, :3] = _skew(p) @ R
    Xi = np.linalg.inv(X)
    return Xi.T @ I6 @ Xi


def _subtree_inertia_in_parent_body_frame(joint):
    """MultiBodySystemTools.computeSubtreeInertia(joint) re-expressed in the body-fixed frame of joint.getPredecessor(), with every
    joint of the subtree at its zero configuration."""
    body = joint.successor
    I6 = _spatial_inertia6(body.momentOfInertia, body.mass, body.centerOfMassOffset)
    for child in body.childrenJoints:
        I6 = I6 + _subtree_inertia_in_parent_body_frame(child)
    # body-fixed(child body) -> after-joint (inertiaPose) -> [joint at zero: identity] -> before-joint -> predecessor's after-joint frame
    T = _xf(body.inertiaPose)
    if joint.transformToParent is not None:
        T = _compose(_xf(joint.transformToParent), T)
    pred = joint.predecessor
    if not pred.isRootBody():  # ... -> predecessor's body-fixed frame
        Rc, pc = _xf(pred.inertiaPose)
        T = _compose((Rc.T, -Rc.T @ pc), T)
    return _inertia_to_parent(I6, T[0], T[1])


class RigidBody:
    """multiBodySystem/RigidBody.java.  ``RigidBody(name)`` creates a root body ("elevator", :79-108);
    ``RigidBody(name, parentJoint, momentOfInertia, mass, centerOfMassOffset | inertiaPose)`` a moving body whose
    body-fixed frame is ``inertiaPose`` under the frame after the parent joint (:123-185)."""

    def __init__(self, name: str, parentJoint: "Joint" = None, momentOfInertia=None, mass: float = 0.0,
                 centerOfMassOffset=None, inertiaPose=None):
        if name is None:
            raise ValueError("Name can not be null")  # RigidBody.java:172-173
        self.name = name
        self.parentJoint = parentJoint
        self.childrenJoints: List[Joint] = []
        if parentJoint is None:
            self.momentOfInertia = None
            self.mass = 0.0
            self.inertiaPose = IDENTITY12.copy()
            return
        if inertiaPose is not None:
            self.inertiaPose = _as_transform(inertiaPose)
        else:
            c = np.zeros(3) if centerOfMassOffset is None else np.asarray(centerOfMassOffset, dtype=np.float64)
            self.inertiaPose = np.concatenate([np.eye(3).reshape(9), c.reshape(3)])
        J = np.asarray(momentOfInertia, dtype=np.float64)
        if J.shape == (3,):
            J = np.diag(J)
        self.momentOfInertia = J.reshape(3, 3).copy()
        self.mass = float(mass)
        # SpatialInertia's own CoM offset inside the body-fixed frame: zero for bodies built this way (RigidBody.java:141-146)
        self.centerOfMassOffset = np.zeros(3)
        parentJoint.setSuccessor(self)

    def isRootBody(self) -> bool:
        return self.parentJoint is None

    def getParentJoint(self):
        return self.parentJoint

    def getChildrenJoints(self):
        return self.childrenJoints

    def addChildJoint(self, joint: "Joint"):
        self.childrenJoints.append(joint)

    def getName(self):
        return self.name

    def subtreeJointList(self) -> List["Joint"]:
        """Depth-first pre-order, children in insertion order (iterators/JointIterator.java:130-177)."""
        out: List[Joint] = []
        stack = list(reversed(self.childrenJoints))
        while stack:
            j = stack.pop()
            out.append(j)
            if j.successor is not None:
                stack.extend(reversed(j.successor.childrenJoints))
        return out

    def __repr__(self):
        return f"RigidBody({self.name})"


class Joint:
    """multiBodySystem/Joint.java:12-105.  ``transformToParent`` = pose of the frame before the joint in the
    frame after the parent joint (``None`` = that frame itself, tools/MecanoFactories.java:81-91)."""

    joint_type = -1
    degreesOfFreedom = 0
    configurationMatrixSize = 0

    def __init__(self, name: str, predecessor: RigidBody, transformToParent=None):
        if name is None:
            raise ValueError("Name can not be null")
        self.name = name
        self.predecessor = predecessor
        self.successor: Optional[RigidBody] = None
        self.transformToParent = _as_transform(transformToParent)
        predecessor.addChildJoint(self)

    def setSuccessor(self, successor: RigidBody):
        self.successor = successor

    def getPredecessor(self):
        return self.predecessor

    def getSuccessor(self):
        return self.successor

    def getName(self):
        return self.name

    def getDegreesOfFreedom(self) -> int:
        return self.degreesOfFreedom

    def getConfigurationMatrixSize(self) -> int:
        return self.configurationMatrixSize

    # ---- joint state, as JointBasics / JointReadOnly carry it (multiBodySystem/interfaces/JointBasics.java:150-224,
    # JointReadOnly.java:438-514): the matrix forms the calculators' one-configuration calls and MultiBodySystemTools.extractJointsState /
    # insertJointsState go through.  Each returns rowStart + the rows it consumed, like the reference.  A floating joint's configuration
    # is (qx, qy, qz, qs, x, y, z), its velocity / acceleration / effort (angular, linear) in the frame after the joint
    # (SixDoFJointReadOnly.java:21-68); a fresh joint sits at the identity, at rest.
    def _state(self, which):
        st = self.__dict__.setdefault("_joint_state", {})
        if which not in st:
            n = self.configurationMatrixSize if which == "q" else self.degreesOfFreedom
            a = np.zeros(n)
            if which == "q" and self.joint_type in (SIXDOF, SPHERICAL):
                a[3] = 1.0  # identity quaternion
            st[which] = a
        return st[which]

    def _set(self, which, rowStart, matrix):
        a = self._state(which)
        m = np.asarray(matrix, dtype=np.float64).reshape(-1)
        a[:] = m[rowStart:rowStart + a.size]
        return rowStart + a.size

    def _get(self, which, rowStart, matrixToPack):
        a = self._state(which)
        np.asarray(matrixToPack).reshape(-1)[rowStart:rowStart + a.size] = a
        return rowStart + a.size

    def setJointConfiguration(self, rowStart, matrix):
        return self._set("q", rowStart, matrix)

    def setJointVelocity(self, rowStart, matrix):
        return self._set("qd", rowStart, matrix)

    def setJointAcceleration(self, rowStart, matrix):
        return self._set("qdd", rowStart, matrix)

    def setJointTau(self, rowStart, matrix):
        return self._set("tau", rowStart, matrix)

    def getJointConfiguration(self, rowStart, matrixToPack):
        return self._get("q", rowStart, matrixToPack)

    def getJointVelocity(self, rowStart, matrixToPack):
        return self._get("qd", rowStart, matrixToPack)

    def getJointAcceleration(self, rowStart, matrixToPack):
        return self._get("qdd", rowStart, matrixToPack)

    def getJointTau(self, rowStart, matrixToPack):
        return self._get("tau", rowStart, matrixToPack)

    def subtreeList(self) -> List["Joint"]:
        out = [self]
        if self.successor is not None:
            out.extend(self.successor.subtreeJointList())
        return out

    def __repr__(self):
        return f"{type(self).__name__}({self.name})"


class OneDoFJoint(Joint):
    degreesOfFreedom = 1
    configurationMatrixSize = 1

    def __init__(self, name, predecessor, transformToParent, jointAxis):
        super().__init__(name, predecessor, transformToParent)
        self.jointAxis = np.asarray(jointAxis, dtype=np.float64).reshape(3).copy()

    def getJointAxis(self):
        return self.jointAxis

    # OneDoFJointBasics.setQ / setQd / setQdd / setTau and their getters (multiBodySystem/interfaces/OneDoFJointBasics.java)
    def setQ(self, q):
        self._state("q")[0] = float(q)

    def setQd(self, qd):
        self._state("qd")[0] = float(qd)

    def setQdd(self, qdd):
        self._state("qdd")[0] = float(qdd)

    def setTau(self, tau):
        self._state("tau")[0] = float(tau)

    def getQ(self):
        return float(self._state("q")[0])

    def getQd(self):
        return float(self._state("qd")[0])

    def getQdd(self):
        return float(self._state("qdd")[0])

    def getTau(self):
        return float(self._state("tau")[0])


class RevoluteJoint(OneDoFJoint):
    joint_type = REVOLUTE

    def __init__(self, name, predecessor, transformToParent=None, jointAxis=(0.0, 0.0, 1.0)):
        super().__init__(name, predecessor, transformToParent, jointAxis)


class PrismaticJoint(OneDoFJoint):
    joint_type = PRISMATIC

    def __init__(self, name, predecessor, transformToParent=None, jointAxis=(0.0, 0.0, 1.0)):
        super().__init__(name, predecessor, transformToParent, jointAxis)


class SixDoFJoint(Joint):
    joint_type = SIXDOF
    degreesOfFreedom = 6
    configurationMatrixSize = 7


class FixedJoint(Joint):
    joint_type = FIXED
    degreesOfFreedom = 0
    configurationMatrixSize = 0


class PlanarJoint(Joint):
    """multiBodySystem/PlanarJoint.java, interfaces/PlanarJointReadOnly.java:17-72: motion in the XZ plane of the frame before the
    joint; q = (pitch, x, z), qd / qdd / tau = (about y, along x, along z) in the frame after the joint."""
    joint_type = PLANAR
    degreesOfFreedom = 3
    configurationMatrixSize = 3


class SphericalJoint(Joint):
    """multiBodySystem/SphericalJoint.java, interfaces/SphericalJointReadOnly.java:18-104: q = quaternion (x, y, z, s), qd / qdd /
    tau = angular velocity / acceleration / moment in the frame after the joint."""
    joint_type = SPHERICAL
    degreesOfFreedom = 3
    configurationMatrixSize = 4


class JointMatrixIndexProvider:
    """multiBodySystem/interfaces/JointMatrixIndexProvider.java:71-123: running sums in list order."""

    def __init__(self, joints: Sequence[Joint]):
        self._joints = list(joints)
        self._dof = {}
        self._cfg = {}
        d = c = 0
        for j in self._joints:
            self._dof[id(j)] = list(range(d, d + j.getDegreesOfFreedom()))
            self._cfg[id(j)] = list(range(c, c + j.getConfigurationMatrixSize()))
            d += j.getDegreesOfFreedom()
            c += j.getConfigurationMatrixSize()
        self.numberOfDoFs = d
        self.configurationSize = c

    @staticmethod
    def toIndexProvider(joints: Sequence[Joint]) -> "JointMatrixIndexProvider":
        return JointMatrixIndexProvider(joints)

    def getIndexedJointsInOrder(self) -> List[Joint]:
        return self._joints

    def getJointDoFIndices(self, joint: Joint) -> List[int]:
        return self._dof[id(joint)]

    def getJointConfigurationIndices(self, joint: Joint) -> List[int]:
        return self._cfg[id(joint)]


class JointStateType:
    """tools/JointStateType.java"""
    CONFIGURATION, VELOCITY, ACCELERATION, EFFORT = "q", "qd", "qdd", "tau"


class MultiBodySystemTools:
    """The two state-packing helpers of tools/MultiBodySystemTools.java the calculators' callers use (extractJointsState :1433-1491,
    insertJointsState :1578-1637): joints in the given order, each taking getConfigurationMatrixSize() / getDegreesOfFreedom() rows."""

    @staticmethod
    def extractJointsState(joints, stateSelection, matrixToPack):
        row = 0
        for j in joints:
            row = j._get(stateSelection, row, matrixToPack)
        return row

    @staticmethod
    def insertJointsState(joints, stateSelection, matrix):
        row = 0
        for j in joints:
            row = j._set(stateSelection, row, matrix)
        return row


@dataclass
class ModelDesc:
    """The flat arrays of ``mh_model_desc`` (include/mecano_hip.h)."""

    n_joints: int
    nq: int
    nv: int
    parent: np.ndarray
    joint_type: np.ndarray
    axis: np.ndarray
    X_before: np.ndarray
    X_com: np.ndarray
    inertia_J: np.ndarray
    inertia_mass: np.ndarray
    inertia_com: np.ndarray
    dof_indices: np.ndarray
    cfg_indices: np.ndarray

    def topology_key(self) -> str:
        """Stable key of (parents, joint kinds): what a topology-specialised kernel is compiled for."""
        import hashlib
        h = hashlib.sha1()
        h.update(self.parent.astype(np.int32).tobytes())
        h.update(self.joint_type.astype(np.int32).tobytes())
        return h.hexdigest()[:12]


class MultiBodySystem:
    """multiBodySystem/interfaces/MultiBodySystemReadOnly.java: root body + joints to consider / ignore + index provider."""

    def __init__(self, rootBody: RigidBody, jointsToIgnore: Iterable[Joint] = (), jointMatrixIndexProvider=None):
        self.rootBody = rootBody
        self.allJoints = rootBody.subtreeJointList()
        ignored = []
        for j in jointsToIgnore:  # a joint to ignore takes its whole subtree with it (MultiBodySystemReadOnly.java:167-171)
            for jj in j.subtreeList():
                if jj not in ignored:
                    ignored.append(jj)
        self.jointsToIgnore = ignored
        ignored_ids = {id(j) for j in ignored}
        self.jointsToConsider = [j for j in self.allJoints if id(j) not in ignored_ids]
        self.jointMatrixIndexProvider = jointMatrixIndexProvider or JointMatrixIndexProvider.toIndexProvider(self.jointsToConsider)

    @staticmethod
    def toMultiBodySystemInput(rootBody: RigidBody, jointsToIgnore: Iterable[Joint] = ()) -> "MultiBodySystem":
        while not rootBody.isRootBody():  # MultiBodySystemTools.getRootBody
            rootBody = rootBody.getParentJoint().getPredecessor()
        return MultiBodySystem(rootBody, jointsToIgnore)

    def getRootBody(self):
        return self.rootBody

    def getAllJoints(self):
        return self.allJoints

    def getJointsToConsider(self):
        return self.jointsToConsider

    def getJointsToIgnore(self):
        return self.jointsToIgnore

    def getJointMatrixIndexProvider(self):
        return self.jointMatrixIndexProvider

    def getNumberOfDoFs(self) -> int:
        return sum(j.getDegreesOfFreedom() for j in self.jointsToConsider)

    def getConfigurationSize(self) -> int:
        return sum(j.getConfigurationMatrixSize() for j in self.jointsToConsider)

    # ------------------------------------------------------------------ flattening (INTEGRATION.md, Appendix B of SURVEY.md)
    def toModelDesc(self, considerIgnoredSubtreesInertia: bool = False) -> ModelDesc:
        """Model-extraction recipe of tools/MultiBodySystemFactories.java:401-470,782-868 applied to this mirror.

        ``considerIgnoredSubtreesInertia``: the inertia of every ignored subtree is added to the body it hangs from, as
        InverseDynamicsCalculator.java:832-860 does with MultiBodySystemTools.computeSubtreeInertia (:47-66).  Mecano freezes that
        lump at the joint configuration current at construction time; the joints of this mirror carry no state, so the lump is
        taken at the zero configuration (every ignored joint at its identity transform)."""
        provider = self.jointMatrixIndexProvider
        joints = provider.getIndexedJointsInOrder()
        index_of = {id(j): i for i, j in enumerate(joints)}
        n = len(joints)
        parent = np.full(n, -1, dtype=np.int32)
        jtype = np.zeros(n, dtype=np.int32)
        axis = np.zeros((n, 3))
        Xb = np.tile(IDENTITY12, (n, 1))
        Xc = np.tile(IDENTITY12, (n, 1))
        J = np.zeros((n, 9))
        mass = np.zeros(n)
        com = np.zeros((n, 3))
        dof, cfg = [], []
        for i, j in enumerate(joints):
            if j.joint_type not in (REVOLUTE, PRISMATIC, SIXDOF, FIXED, PLANAR, SPHERICAL):
                raise NotImplementedError(f"unsupported joint kind: {j}")
            jtype[i] = j.joint_type
            pj = j.getPredecessor().getParentJoint()
            if pj is not None:
                if id(pj) not in index_of:
                    raise ValueError(f"{j} hangs below an ignored joint")
                parent[i] = index_of[id(pj)]
            if j.transformToParent is not None:
                Xb[i] = j.transformToParent
            if isinstance(j, OneDoFJoint):
                axis[i] = j.jointAxis
            body = j.getSuccessor()
            if body is None:
                raise ValueError(f"{j} has no successor")
            Xc[i] = body.inertiaPose
            J[i] = body.momentOfInertia.reshape(9)
            mass[i] = body.mass
            com[i] = body.centerOfMassOffset
            if considerIgnoredSubtreesInertia:
                ignored_ids = {id(x) for x in self.jointsToIgnore}
                I6 = _spatial_inertia6(body.momentOfInertia, body.mass, body.centerOfMassOffset)
                lumped = False
                for child in body.childrenJoints:
                    if id(child) in ignored_ids:
                        I6 = I6 + _subtree_inertia_in_parent_body_frame(child)
                        lumped = True
                if lumped:
                    m_tot = I6[5, 5]
                    mc = np.array([I6[2, 4], I6[0, 5], I6[1, 3]])  # m [c]x block: (2,1)->cx, (0,2)->cy, (1,0)->cz
                    J[i] = I6[:3, :3].reshape(9)
                    mass[i] = m_tot
                    com[i] = mc / m_tot if abs(m_tot) >= 1.0e-7 else np.zeros(3)
            dof.extend(provider.getJointDoFIndices(j))
            cfg.extend(provider.getJointConfigurationIndices(j))
        nv = (max(dof) + 1) if dof else 0
        nq = (max(cfg) + 1) if cfg else 0
        return ModelDesc(n, nq, nv, parent, jtype, axis.reshape(-1), Xb.reshape(-1), Xc.reshape(-1), J.reshape(-1), mass,
                         com.reshape(-1), np.asarray(dof, dtype=np.int32), np.asarray(cfg, dtype=np.int32))
