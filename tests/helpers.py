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


# ---- fp32 tolerances.  u = 2^-24.  Every output of RNEA / CRBA is a sum along tree paths of at most n bodies with about eight rounded
# operations per body and path (two 3 x 3 rotations, a cross product, the accumulation); a worst-case bound grows like 8 n u, but rounding
# errors of that many independent operations add up like a random walk (Higham & Mary, "A new approach to probabilistic rounding error
# analysis", SIAM J. Sci. Comput. 41, 2019): sqrt(8 n) u with a modest constant.  The constants below were set from the achieved errors
# of round 2 (profiles/r02_parity_errors.json: worst achieved / bound = 0.22 for RNEA, 0.11 for CRBA, 0.16 for the backward error of
# forward dynamics, 0.10 for its forward error) -- round 2's 64 n u bounds were 300-1000 x looser than what the kernels achieve, so that a
# hundredfold loss of accuracy would have passed.
U32 = 2.0 ** -24


def f32_forward_tol(n_bodies, c=4.0):
    """relative tolerance (times max(1, |ref|_inf), as `close` applies it) of a forward recursion over <= n_bodies bodies in fp32"""
    return c * (8.0 * n_bodies) ** 0.5 * U32


def f32_aba_backward_tol(n_bodies):
    """forward dynamics in fp32, backward error in effort space relative to (|tau| + |bias|)_inf: three sweeps and a division per body"""
    return f32_forward_tol(n_bodies, 16.0)


def f32_aba_forward_factor(n_bodies):
    """forward error of fp32 forward dynamics relative to cond_inf(H) u of the row"""
    return 8.0 * (8.0 * n_bodies) ** 0.5


def close_aba(actual, ref, H_ref, n_bodies, u=2.0 ** -53, label=None):
    """Forward dynamics against the oracle with a bound per ROW instead of one loosened tolerance for a whole family: the solve
    H qdd = tau - h amplifies rounding by cond(H), and random mixed trees reach cond_inf(H) of 1e6 .. 1e9 on some rows while most stay
    near 1e2.  |qdd - ref|_inf <= 8 sqrt(8 n) cond_inf(H) u max(1, |ref|_inf) on every row, H from the oracle's own mass matrix (u: the unit
    roundoff of the precision under test).  Logs the worst err / (cond u scale) against the factor."""
    actual, ref, H_ref = np.asarray(actual, dtype=np.float64), np.asarray(ref, dtype=np.float64), np.asarray(H_ref, dtype=np.float64)
    assert actual.shape == ref.shape and H_ref.shape[0] == ref.shape[0]
    if ref.size == 0:
        return 0.0
    conds = np.array([np.linalg.cond(H_ref[k], np.inf) for k in range(len(H_ref))])
    rel = np.abs(actual - ref).max(axis=1) / np.maximum(1.0, np.abs(ref).max(axis=1))
    ratio = float((rel / (conds * u)).max())
    factor = f32_aba_forward_factor(n_bodies)
    record_parity(ratio, factor, (label or "aba") + " forward error / (cond_inf(H) u)")
    assert ratio <= factor, f"worst row: err / (cond u) = {ratio:.3e} > {factor:.3e} (cond_inf(H) {conds.min():.1e} .. {conds.max():.1e})"
    return ratio
