"""Shared builders for the tests, and the log of achieved parity errors (written to gpurun_out/parity_errors.json at session end)."""
import os

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


# ---- achieved errors per test: {test id: {"max_err": worst |actual - ref| seen, "bound": the bound it was held to, "checks": n}}
PARITY_LOG = {}


def record_parity(err, bound, label=None):
    name = os.environ.get("PYTEST_CURRENT_TEST", "unknown").split(" (")[0]
    if label:
        name += " :: " + label
    e = PARITY_LOG.setdefault(name, {"max_err": 0.0, "bound": float(bound), "checks": 0})
    e["max_err"] = max(e["max_err"], float(err))
    e["bound"] = max(e["bound"], float(bound))
    e["checks"] += 1


def close(actual, ref, tol=1.0e-10, absolute=False, label=None):
    """|actual - ref|_inf <= tol * max(1, |ref|_inf), or <= tol outright with absolute=True (the north star's "within 1e-10" on
    BASELINE.json's fp64 configurations).  Every check lands in PARITY_LOG."""
    actual, ref = np.asarray(actual), np.asarray(ref)
    assert actual.shape == ref.shape, (actual.shape, ref.shape)
    err = float(np.abs(actual - ref).max()) if ref.size else 0.0
    bound = tol if absolute else tol * max(1.0, float(np.abs(ref).max()) if ref.size else 0.0)
    record_parity(err, bound, label)
    assert err <= bound, f"max err {err:.3e} > {bound:.3e}"
    return err
