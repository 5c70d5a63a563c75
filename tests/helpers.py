"""Shared builders for the tests."""
import numpy as np

from mecano_amd import random_tools as rt
from mecano_amd.multibody import FixedJoint, PrismaticJoint, RevoluteJoint, RigidBody


def build_lump_pair(weld):
    """A 3-joint tree with a 2-joint side subtree below the first body.  weld=True: the side subtree hangs on fixed joints;
    weld=False: on a revolute and a prismatic joint (to be ignored and lumped).  Identical physical parameters in both builds."""
    r = np.random.default_rng(5)
    root = RigidBody("root")
    j0 = rt.nextRevoluteJoint(r, "j0", root)
    b0 = rt.nextRigidBody(r, "b0", j0)
    j1 = rt.nextPrismaticJoint(r, "j1", b0)
    b1 = rt.nextRigidBody(r, "b1", j1)
    ax2, off2 = rt.nextUnitVector3D(r), rt.nextRigidBodyTransform(r)
    k0 = FixedJoint("k0", b0, off2) if weld else RevoluteJoint("k0", b0, off2, ax2)
    c0 = rt.nextRigidBody(r, "c0", k0)
    ax3, off3 = rt.nextUnitVector3D(r), rt.nextRigidBodyTransform(r)
    k1 = FixedJoint("k1", c0, off3) if weld else PrismaticJoint("k1", c0, off3, ax3)
    rt.nextRigidBody(r, "c1", k1)
    j2 = rt.nextRevoluteJoint(r, "j2", b1)
    rt.nextRigidBody(r, "b2", j2)
    return root, k0
