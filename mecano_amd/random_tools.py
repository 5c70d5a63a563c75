"""Seeded random multi-body systems and states, after the reference's test generators.

Distributions follow tools/MultiBodySystemRandomTools.java and tools/MecanoRandomTools.java of the
reference (paths relative to /root/reference/src/main/java/us/ihmc/mecano/):

* moment of inertia  J = L L^T, diag(L) ~ U(1e-4, 2), off-diagonal ~ U(-0.5, 0.5)   (MecanoRandomTools.java:623-647 as called at MultiBodySystemRandomTools.java:1367)
* mass = 0.1 + U(0, 1), CoM offset ~ U(-1, 1)^3                                      (MultiBodySystemRandomTools.java:1365-1370)
* joint axis: uniform unit vector; joint offset: random rigid transform, none when the predecessor is the root body (:1182,1198,1213,1229)
* trees: parent of joint i = successor of a uniformly chosen earlier joint          (:1108-1135)
* states: 1-DoF q, qd, qdd, tau ~ U(-1, 1) scaled; SixDoF: uniform unit quaternion, position / twist / ... ~ U(-1, 1)^3 (:45-125)

The random stream is numpy's PCG64, not java.util.Random: bit-compatibility with the Java
generators is NOT claimed.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .multibody import (FixedJoint, Joint, MultiBodySystem, PlanarJoint, PrismaticJoint, RevoluteJoint, RigidBody, SixDoFJoint, SphericalJoint)


def nextVector3D(rng, lo=-1.0, hi=1.0):
    return rng.uniform(lo, hi, 3)


def nextUnitVector3D(rng):
    v = rng.normal(size=3)
    return v / np.linalg.norm(v)


def nextQuaternion(rng):
    """Uniform unit quaternion (x, y, z, s)."""
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def quaternionToMatrix(q):
    x, y, z, s = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * s), 2 * (x * z + y * s)],
                     [2 * (x * y + z * s), 1 - 2 * (x * x + z * z), 2 * (y * z - x * s)],
                     [2 * (x * z - y * s), 2 * (y * z + x * s), 1 - 2 * (x * x + y * y)]])


def nextRigidBodyTransform(rng):
    return (quaternionToMatrix(nextQuaternion(rng)), nextVector3D(rng))


def nextSymmetricPositiveDefiniteMatrix3D(rng, minDiagonal=1.0e-4, maxDiagonal=2.0, minMaxOffDiagonal=0.5):
    L = np.zeros((3, 3))
    L[0, 0] = rng.uniform(minDiagonal, maxDiagonal)
    L[1, 0] = rng.uniform(-minMaxOffDiagonal, minMaxOffDiagonal)
    L[1, 1] = rng.uniform(minDiagonal, maxDiagonal)
    L[2, 0] = rng.uniform(-minMaxOffDiagonal, minMaxOffDiagonal)
    L[2, 1] = rng.uniform(-minMaxOffDiagonal, minMaxOffDiagonal)
    L[2, 2] = rng.uniform(minDiagonal, maxDiagonal)
    return L @ L.T


def nextRigidBody(rng, name: str, parentJoint: Joint) -> RigidBody:
    J = nextSymmetricPositiveDefiniteMatrix3D(rng)
    mass = 0.1 + rng.uniform()
    com = nextVector3D(rng)
    return RigidBody(name, parentJoint, J, mass, centerOfMassOffset=com)


def _offset(rng, predecessor: RigidBody):
    return None if predecessor.isRootBody() else nextRigidBodyTransform(rng)


def nextRevoluteJoint(rng, name, predecessor, jointAxis=None):
    axis = nextUnitVector3D(rng) if jointAxis is None else jointAxis
    return RevoluteJoint(name, predecessor, _offset(rng, predecessor), axis)


def nextPrismaticJoint(rng, name, predecessor, jointAxis=None):
    axis = nextUnitVector3D(rng) if jointAxis is None else jointAxis
    return PrismaticJoint(name, predecessor, _offset(rng, predecessor), axis)


def nextSixDoFJoint(rng, name, predecessor):
    return SixDoFJoint(name, predecessor, _offset(rng, predecessor))


def nextFixedJoint(rng, name, predecessor):
    return FixedJoint(name, predecessor, _offset(rng, predecessor))


def nextPlanarJoint(rng, name, predecessor):
    return PlanarJoint(name, predecessor, _offset(rng, predecessor))


def nextSphericalJoint(rng, name, predecessor):
    return SphericalJoint(name, predecessor, _offset(rng, predecessor))


_KINDS = {"revolute": nextRevoluteJoint, "prismatic": nextPrismaticJoint, "sixdof": nextSixDoFJoint, "fixed": nextFixedJoint,
          "planar": nextPlanarJoint, "spherical": nextSphericalJoint}


def _next_joint(rng, kinds: Sequence[str], name, predecessor):
    return _KINDS[kinds[rng.integers(len(kinds))]](rng, name, predecessor)


def nextJointChain(rng, numberOfJoints: int, kinds=("revolute",), rootBody: RigidBody = None, prefix="") -> List[Joint]:
    """nextRevoluteJointChain / nextPrismaticJointChain / nextOneDoFJointChain / nextJointChain."""
    predecessor = rootBody or RigidBody(prefix + "rootBody")
    joints = []
    for i in range(numberOfJoints):
        j = _next_joint(rng, kinds, f"{prefix}joint{i}", predecessor)
        predecessor = nextRigidBody(rng, f"{prefix}body{i}", j)
        joints.append(j)
    return joints


def nextJointTree(rng, numberOfJoints: int, kinds=("revolute",), rootBody: RigidBody = None, prefix="") -> List[Joint]:
    """MultiBodySystemRandomTools.nextJointTree (:1108-1135); returns the joints in DFS pre-order like the reference."""
    root = rootBody or RigidBody(prefix + "rootBody")
    predecessor = root
    created = []
    for i in range(numberOfJoints):
        j = _next_joint(rng, kinds, f"{prefix}joint{i}", predecessor)
        nextRigidBody(rng, f"{prefix}body{i}", j)
        created.append(j)
        predecessor = created[rng.integers(len(created))].getSuccessor()
    return root.subtreeJointList()


def nextFloatingChain(rng, numberOfJoints: int, kinds=("revolute",), tree=False) -> List[Joint]:
    """SixDoF root joint (no offset, as InverseDynamicsCalculatorTest.java:129-133) + a random chain or tree below it."""
    root = RigidBody("rootBody")
    floating = SixDoFJoint("floatingJoint", root)
    pelvis = nextRigidBody(rng, "floatingBody", floating)
    if tree:
        nextJointTree(rng, numberOfJoints, kinds, rootBody=pelvis)
    else:
        nextJointChain(rng, numberOfJoints, kinds, rootBody=pelvis)
    return root.subtreeJointList()


def humanoid30Desc():
    """The committed benchmark model (mecano_amd/models/humanoid30.json): the 30-DoF humanoid of BASELINE.json's metric as a ModelDesc,
    byte for byte what nextHumanoid(numpy.random.default_rng(43)).toModelDesc() produced when it was generated."""
    import json
    import os
    from .multibody import ModelDesc
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "humanoid30.json")))
    return ModelDesc(d["n_joints"], d["nq"], d["nv"], np.array(d["parent"], dtype=np.int32), np.array(d["joint_type"], dtype=np.int32),
                     np.array(d["axis"]), np.array(d["X_before"]), np.array(d["X_com"]), np.array(d["inertia_J"]), np.array(d["inertia_mass"]),
                     np.array(d["inertia_com"]), np.array(d["dof_indices"], dtype=np.int32), np.array(d["cfg_indices"], dtype=np.int32))


def modelDescFromJson(name: str):
    """A committed model of mecano_amd/models/ (flat mh_model_desc form) as a ModelDesc: "humanoid30", "arm7", "tree128"."""
    import json
    import os
    from .multibody import ModelDesc
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", name + ".json")))
    return ModelDesc(d["n_joints"], d["nq"], d["nv"], np.array(d["parent"], dtype=np.int32), np.array(d["joint_type"], dtype=np.int32),
                     np.array(d["axis"]), np.array(d["X_before"]), np.array(d["X_com"]), np.array(d["inertia_J"]), np.array(d["inertia_mass"]),
                     np.array(d["inertia_com"]), np.array(d["dof_indices"], dtype=np.int32), np.array(d["cfg_indices"], dtype=np.int32))


def committedBenchmarkSystems():
    """The three mechanisms BASELINE.json's configs name, as this repository generates them (model seeds fixed here; the committed JSON files
    under mecano_amd/models/ are their flattened form, tests/golden/make_model_fixtures.py):
      humanoid30  configs 3 / 4 and the metric: nextHumanoid(default_rng(43))
      arm7        configs 1 / 2: a 7-joint revolute chain, nextJointChain(default_rng(43), 7)
      tree128     config 5: nextJointTree(default_rng(128), 128, revolute / prismatic / 6-DoF) -- the tree bench.py --config 5 times"""
    return {"humanoid30": nextHumanoid(np.random.default_rng(43)),
            "arm7": MultiBodySystem.toMultiBodySystemInput(nextJointChain(np.random.default_rng(43), 7)[0].getPredecessor()),
            "tree128": MultiBodySystem.toMultiBodySystemInput(
                nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())}


def referenceBenchmarkSystems(seed: int = 43, numberOfJoints: int = 30):
    """The four systems of the reference's own (disabled) RNEA benchmarks, InverseDynamicsCalculatorTest.java:24-158: a 30-joint random
    1-DoF chain and tree on a fixed base, and the same below a SixDoF root joint; seed 43 each (the reference's seed; the RNG is numpy's,
    bit compatibility with java.util.Random is not claimed)."""
    one_dof = ("revolute", "prismatic")
    out = {}
    rng = np.random.default_rng(seed)
    out["chain30"] = MultiBodySystem.toMultiBodySystemInput(nextJointChain(rng, numberOfJoints, one_dof)[0].getPredecessor())
    rng = np.random.default_rng(seed)
    out["floating_chain30"] = MultiBodySystem.toMultiBodySystemInput(nextFloatingChain(rng, numberOfJoints, one_dof)[0].getPredecessor())
    rng = np.random.default_rng(seed)
    out["tree30"] = MultiBodySystem.toMultiBodySystemInput(nextJointTree(rng, numberOfJoints, one_dof)[0].getPredecessor())
    rng = np.random.default_rng(seed)
    out["floating_tree30"] = MultiBodySystem.toMultiBodySystemInput(nextFloatingChain(rng, numberOfJoints, one_dof, tree=True)[0].getPredecessor())
    return out


def nextHumanoid(rng) -> MultiBodySystem:
    """The 30-DoF humanoid of BASELINE.json configs[2..3] / SURVEY.md section 8d: SixDoF pelvis + 24 revolute joints:
    legs 6 x 2, waist/torso 3, arms 4 x 2, neck 1  =>  nv = 30, nq = 31, 25 moving bodies, depth 8 pelvis->hand.
    DFS pre-order with children in creation order: left leg, right leg, torso chain, left arm, right arm, neck."""
    root = RigidBody("elevator")
    pelvis_joint = SixDoFJoint("pelvis", root)
    pelvis = nextRigidBody(rng, "pelvisBody", pelvis_joint)

    def chain(prefix, base, count):
        body = base
        for k in range(count):
            j = nextRevoluteJoint(rng, f"{prefix}{k}", body)
            body = nextRigidBody(rng, f"{prefix}{k}Body", j)
        return body

    chain("leftLeg", pelvis, 6)
    chain("rightLeg", pelvis, 6)
    chest = chain("spine", pelvis, 3)
    chain("leftArm", chest, 4)
    chain("rightArm", chest, 4)
    chain("neck", chest, 1)
    return MultiBodySystem.toMultiBodySystemInput(root)


def nextQuadruped(rng) -> MultiBodySystem:
    """SixDoF trunk + four 3-joint legs (13 bodies, nv = 18): the only branching body is the root, so a tree-split code object has limbs
    but no sub-trunk (plain, un-staged plan)."""
    root = RigidBody("elevator")
    trunk = nextRigidBody(rng, "trunkBody", SixDoFJoint("trunk", root))
    for leg in ("frontLeft", "frontRight", "hindLeft", "hindRight"):
        body = trunk
        for k in range(3):
            body = nextRigidBody(rng, f"{leg}{k}Body", nextRevoluteJoint(rng, f"{leg}{k}", body))
    return MultiBodySystem.toMultiBodySystemInput(root)


def nextFixedBaseTorso(rng) -> MultiBodySystem:
    """A fixed-base machine that exercises the other corners of the tree-split planner: a REVOLUTE root body carrying (i) a two-body
    sub-trunk that ends in two mixed revolute / prismatic arms of different length plus a one-body head, (ii) a four-body late limb and
    (iii) a ONE-body late limb (no room for a cut: its owner takes the explicit barrier).  13 bodies, all 1-DoF."""
    root = RigidBody("base")
    waist = nextRigidBody(rng, "waistBody", nextRevoluteJoint(rng, "waist", root))

    def chain(prefix, base, kinds):
        body = base
        for k, kind in enumerate(kinds):
            j = (nextRevoluteJoint if kind == "r" else nextPrismaticJoint)(rng, f"{prefix}{k}", body)
            body = nextRigidBody(rng, f"{prefix}{k}Body", j)
        return body

    chest = chain("spine", waist, "rp")
    chain("leftArm", chest, "rrp")
    chain("rightArm", chest, "rr")
    chain("head", chest, "r")
    chain("boom", waist, "rprr")
    chain("stub", waist, "p")
    return MultiBodySystem.toMultiBodySystemInput(root)


def nextCentaur(rng) -> MultiBodySystem:
    """SixDoF barrel with TWO sub-trunks -- a two-body torso carrying two 2-joint arms, a one-body rump carrying two 3-joint hind legs --
    and a one-body prismatic tail on the barrel itself (the only late limb): the staged tree-split plan with two sub-trunks folded by two
    different waves between the barriers.  15 bodies, nv = 20."""
    root = RigidBody("elevator")
    barrel = nextRigidBody(rng, "barrelBody", SixDoFJoint("barrel", root))

    def chain(prefix, base, count, prismatic=False):
        body = base
        for k in range(count):
            j = (nextPrismaticJoint if prismatic else nextRevoluteJoint)(rng, f"{prefix}{k}", body)
            body = nextRigidBody(rng, f"{prefix}{k}Body", j)
        return body

    chest = chain("torso", barrel, 2)
    chain("leftArm", chest, 2)
    chain("rightArm", chest, 2)
    rump = chain("rump", barrel, 1)
    chain("leftHind", rump, 3)
    chain("rightHind", rump, 3)
    chain("tail", barrel, 1, prismatic=True)
    return MultiBodySystem.toMultiBodySystemInput(root)


def nextState(rng, system: MultiBodySystem, batch: int, q_range=np.pi):
    """Random batched (q, qd, qdd, tau) matrices [B, nq] / [B, nv] honouring the system's index provider."""
    provider = system.getJointMatrixIndexProvider()
    nq = max((max(provider.getJointConfigurationIndices(j), default=-1) for j in provider.getIndexedJointsInOrder()), default=-1) + 1
    nv = max((max(provider.getJointDoFIndices(j), default=-1) for j in provider.getIndexedJointsInOrder()), default=-1) + 1
    q = np.zeros((batch, nq))
    qd = rng.uniform(-1, 1, (batch, nv))
    qdd = rng.uniform(-1, 1, (batch, nv))
    tau = rng.uniform(-1, 1, (batch, nv))
    for j in provider.getIndexedJointsInOrder():
        ci = provider.getJointConfigurationIndices(j)
        if isinstance(j, SixDoFJoint):
            quat = rng.normal(size=(batch, 4))
            quat /= np.linalg.norm(quat, axis=1, keepdims=True)
            q[:, ci[:4]] = quat
            q[:, ci[4:]] = rng.uniform(-1, 1, (batch, 3))
        elif isinstance(j, SphericalJoint):
            quat = rng.normal(size=(batch, 4))
            quat /= np.linalg.norm(quat, axis=1, keepdims=True)
            q[:, ci] = quat
        elif isinstance(j, PlanarJoint):
            q[:, ci[0]] = rng.uniform(-q_range, q_range, batch)
            q[:, ci[1:]] = rng.uniform(-1, 1, (batch, 2))
        elif isinstance(j, RevoluteJoint):
            q[:, ci[0]] = rng.uniform(-q_range, q_range, batch)
        elif isinstance(j, PrismaticJoint):
            q[:, ci[0]] = rng.uniform(-1, 1, batch)
    return q, qd, qdd, tau
