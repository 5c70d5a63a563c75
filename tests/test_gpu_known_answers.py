"""The reference's known answers and unit-level invariants, through the C-ABI on the GPU.

tests/test_oracle_units.py and tests/test_oracle.py restate them on the CPU oracle; here the HIP path has to meet the same closed forms
by itself (no oracle in the loop where a closed form exists), so that the device primitives -- (A, L, C) congruences, rigid-inertia
shifts, twist propagation, Newton-Euler wrench, planar / spherical joint maps, integrator -- are pinned the way the reference pins its
own (paths relative to /root/reference/src/test/java/us/ihmc/mecano/).  fp32 entry points are compared with the fp64 oracle.
"""
import zlib

import numpy as np
import pytest

from helpers import close, f32_aba_backward_tol, f32_aba_forward_factor, f32_forward_tol, record_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def dev(torch, x, dtype=None):
    return torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dtype or torch.float64)


def system_of(joints):
    from mecano_amd.multibody import MultiBodySystem
    return MultiBodySystem.toMultiBodySystemInput(joints[0].getPredecessor())


def test_planar_joint_ballistic_on_the_device(torch_cuda):
    """tools/MultiBodySystemStateIntegratorTest.java:433-498 (testPlanarJointBallistic, EPSILON 1e-12): 1000 steps of mh_aba_f64 +
    mh_integrate_f64 on a unit ball on a PlanarJoint land on the closed-form parabola."""
    torch = torch_cuda
    from mecano_amd.engine import HipModel
    from test_oracle import _free_object, planar_ballistic_check
    hm = HipModel(_free_object("planar").toModelDesc())
    rng = np.random.default_rng(4366346)

    def aba(q, qd, g):
        return hm.aba(dev(torch, q), dev(torch, qd), dev(torch, np.zeros_like(qd)), (0.0, 0.0, g)).cpu().numpy()

    def integrate(dt, q, qd, qdd):
        return [t.cpu().numpy() for t in hm.integrate(dt, dev(torch, q), dev(torch, qd), dev(torch, qdd), return_acceleration=True)]

    for it in range(2):
        worst = planar_ballistic_check(aba, integrate, rng, B=64, steps=1000, checks=(0, 1, 499, 999))
        record_parity(worst, 1e-12, "planar ballistic closed form")
        assert worst <= 1e-12, worst


@pytest.mark.parametrize("kind", ["planar", "spherical"])
def test_planar_and_spherical_steps_against_finite_differences(torch_cuda, kind):
    """tools/MultiBodySystemStateIntegratorTest.java:273-431, 505-575 on the device: zero velocity and acceleration leave the state
    alone; the pose difference over dt reproduces the velocity (first order in dt); a free spinning unit ball has zero angular
    acceleration and keeps its velocity."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    from test_oracle import _free_object, _quat_R
    sys_ = _free_object(kind)
    hm = HipModel(sys_.toModelDesc())
    rng = np.random.default_rng(5464576)
    B = 32
    step = lambda dt, q, v, a: [t.cpu().numpy() for t in hm.integrate(dt, dev(torch, q), dev(torch, v), dev(torch, a), return_acceleration=True)]
    for it in range(4):
        dt = float(rng.uniform(1.0e-5, 1.0e-3))
        q, qd, _, _ = rt.nextState(rng, sys_, B)
        z = np.zeros((B, 3))
        qn, vn, an = step(dt, q, z, z)
        assert np.abs(qn - q).max() <= 1e-12 and not vn.any() and not an.any()
        if kind == "planar":
            v = np.column_stack([np.zeros(B), rng.uniform(-10, 10, B), rng.uniform(-10, 10, B)])
            qn, vn, an = step(dt, q, v, z)
            c, s = np.cos(q[:, 0]), np.sin(q[:, 0])
            dx, dz = (qn[:, 1] - q[:, 1]) / dt, (qn[:, 2] - q[:, 2]) / dt
            fd = np.column_stack([c * dx - s * dz, s * dx + c * dz])
            assert np.abs(fd - v[:, 1:]).max() <= 1e-8 and np.abs(vn - v).max() <= 1e-12 and np.abs(an).max() <= 1e-12
            assert np.abs(qn[:, 0] - q[:, 0]).max() <= 1e-12
        else:
            qdd = hm.aba(dev(torch, q), dev(torch, qd), dev(torch, z), (0.0, 0.0, 0.0)).cpu().numpy()
            assert np.abs(qdd).max() <= 1e-12  # unit ball: w x J w = 0
            qn, vn, an = step(dt, q, qd, qdd)
            for b in range(4):
                dR = _quat_R(q[b]).T @ _quat_R(qn[b])
                w_fd = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (2.0 * dt)
                assert np.abs(w_fd - qd[b]).max() <= 20 * dt
            assert np.abs(vn - qd).max() <= 1e-12 and np.abs(np.linalg.norm(qn, axis=1) - 1.0).max() <= 1e-12


def _floating_with_welded_bodies(rng, n_fixed):
    """A SixDoF root body carrying a chain of n_fixed bodies on FixedJoints with random poses and random inertias."""
    from mecano_amd import random_tools as rt
    from mecano_amd.multibody import RigidBody
    root = RigidBody("root")
    j = rt.nextSixDoFJoint(rng, "floating", root)
    b = rt.nextRigidBody(rng, "base", j)
    for k in range(n_fixed):
        b = rt.nextRigidBody(rng, f"welded{k}", rt.nextFixedJoint(rng, f"weld{k}", b))
    return system_of([j])


def test_articulated_inertia_transform_equals_rigid_inertia_transform_on_the_device(torch_cuda):
    """algorithms/ArticulatedBodyInertiaTest.java:25-60 restated on device code.  On a floating body that carries two bodies through
    two FixedJoints with random poses ("apply two random transforms"), forward dynamics hands the (A, L, C) blocks up through the
    articulated-inertia congruence (rotate, then translate) and inverts the 6x6 it arrives at; the mass-matrix kernel hands (m, m c, I)
    up through the rigid-inertia shift.  Both 6x6 matrices describe the same rigid assembly: H(q) ABA(tau = e_k; qd = 0, g = 0) = e_k."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    rng = np.random.default_rng(2552)
    worst = 0.0
    for it in range(12):
        sys_ = _floating_with_welded_bodies(rng, 2)
        hm = HipModel(sys_.toModelDesc())
        B = 6
        q, _, _, _ = rt.nextState(rng, sys_, 1)
        q = np.repeat(q, B, axis=0)
        tau = np.eye(6)
        z = np.zeros((B, 6))
        X = hm.aba(dev(torch, q), dev(torch, z), dev(torch, tau), (0.0, 0.0, 0.0)).cpu().numpy()  # rows: IA^-1 e_k
        H = hm.crba(dev(torch, q)).cpu().numpy()[0]
        assert np.array_equal(H, H.T)
        worst = max(worst, np.abs(X @ H - np.eye(6)).max() / np.linalg.cond(H))
    record_parity(worst, 1e-12, "|IA^-1 H - 1| / cond(H)")
    assert worst <= 1e-12, worst


def test_kinetic_coenergy_is_frame_invariant_on_the_device(torch_cuda):
    """spatial/SpatialInertiaBasicsTest.java:76-98, 216-248 and tools/MecanoToolsTest.java:618-655 on device code: the kinetic co-energy
    1/2 qd^T H qd (composite inertias shifted and rotated down to every ancestor's frame by the mass-matrix kernel) equals the sum over
    bodies of 1/2 tw^T I tw with the body twists the RNEA kernel propagates, each evaluated in the body's own frame from the description's
    (J, m, c) with the reference's formula (tools/MecanoTools.java:844-890)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    rng = np.random.default_rng(334523)
    kinds = [("revolute", "prismatic"), ("revolute", "prismatic", "sixdof", "fixed"), ("planar", "spherical", "revolute")]
    for it in range(9):
        joints = rt.nextJointTree(rng, int(rng.integers(2, 30)), kinds[it % 3])
        sys_ = system_of(joints)
        d = sys_.toModelDesc()
        hm = HipModel(d)
        B = 17
        q, qd, qdd, _ = rt.nextState(rng, sys_, B)
        H = hm.crba(dev(torch, q)).cpu().numpy()
        _, _, tw = hm.rnea_bodies(dev(torch, q), dev(torch, qd), dev(torch, qdd), (0.0, 0.0, 0.0))
        tw = tw.cpu().numpy()
        T_H = 0.5 * np.einsum("bi,bij,bj->b", qd, H, qd)
        J, m, c = np.asarray(d.inertia_J).reshape(-1, 3, 3), np.asarray(d.inertia_mass), np.asarray(d.inertia_com).reshape(-1, 3)
        w, v = tw[:, :, :3], tw[:, :, 3:]
        T_b = 0.5 * (m[None] * np.einsum("bni,bni->bn", v, v) + 2.0 * m[None] * np.einsum("bni,bni->bn", w, np.cross(c[None], v))
                     + np.einsum("bni,nij,bnj->bn", w, J, w)).sum(axis=1)
        close(T_H, T_b, 1e-11, label="kinetic co-energy")


def test_offset_centre_of_mass_equals_the_same_body_described_about_its_centre_of_mass(torch_cuda):
    """tools/MecanoToolsTest.java:292-460 / spatial/SpatialInertiaBasicsTest.java:129-157 (general Newton-Euler expressions == the fast
    ones) at system level: every body described (i) with its centre of mass offset c in the body-fixed frame, inertia about that frame's
    origin (the reference's general branch; the oracle takes it) and (ii) with the body-fixed frame moved onto the centre of mass, c = 0
    (fast branch).  Same physical system: RNEA, ABA and CRBA agree between the two descriptions on the device, and (i) matches the oracle."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    from mecano_amd.multibody import ModelDesc
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(3453)
    tilde = lambda v: np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])
    for it in range(6):
        joints = rt.nextJointTree(rng, int(rng.integers(2, 25)), ("revolute", "prismatic", "sixdof") if it % 2 else ("revolute", "prismatic"))
        sys_ = system_of(joints)
        d = sys_.toModelDesc()
        n = d.n_joints
        c = rng.uniform(-0.5, 0.5, (n, 3))
        Jc = np.asarray(d.inertia_J).reshape(n, 3, 3)  # taken as the inertia about the centre of mass
        m = np.asarray(d.inertia_mass)
        Xc = np.asarray(d.X_com).reshape(n, 12)
        # (i) same body-fixed frames, CoM at c there: J about the frame origin = Jc - m c~ c~
        J_i = np.stack([Jc[k] - m[k] * tilde(c[k]) @ tilde(c[k]) for k in range(n)])
        d_i = ModelDesc(n, d.nq, d.nv, d.parent, d.joint_type, d.axis, d.X_before, d.X_com, J_i.reshape(-1), m, c.reshape(-1), d.dof_indices,
                        d.cfg_indices)
        # (ii) body-fixed frames moved onto the CoM: p' = p + R c, c = 0, J = Jc
        X_ii = Xc.copy()
        for k in range(n):
            X_ii[k, 9:] += Xc[k, :9].reshape(3, 3) @ c[k]
        d_ii = ModelDesc(n, d.nq, d.nv, d.parent, d.joint_type, d.axis, d.X_before, X_ii.reshape(-1), Jc.reshape(-1), m, np.zeros(3 * n),
                         d.dof_indices, d.cfg_indices)
        h_i, h_ii, om = HipModel(d_i), HipModel(d_ii), OracleModel(d_i)
        B = 33
        q, qd, qdd, tau = rt.nextState(rng, sys_, B)
        g = (0.1, -0.2, -9.81)
        tq, tqd, tqdd, ttau = (dev(torch, x) for x in (q, qd, qdd, tau))
        t_i, t_ii = hm_np(h_i.rnea(tq, tqd, tqdd, g)), hm_np(h_ii.rnea(tq, tqd, tqdd, g))
        close(t_i, t_ii, 1e-11, label="rnea general == fast")
        close(t_i, om.rnea(q, qd, qdd, g), 1e-10, label="rnea general == oracle general")
        a_i = hm_np(h_i.aba(tq, tqd, ttau, g))
        close(a_i, hm_np(h_ii.aba(tq, tqd, ttau, g)), 1e-8 if it % 2 else 1e-9, label="aba general == fast")
        close(a_i, om.aba(q, qd, tau, g), 1e-8 if it % 2 else 1e-9, label="aba general == oracle general")
        close(hm_np(h_i.crba(tq)), hm_np(h_ii.crba(tq)), 1e-11, label="crba general == fast")


def hm_np(t):
    return t.cpu().numpy()


FAMILIES = {
    "prismatic_tree": ("prismatic",), "revolute_chain": ("revolute",), "revolute_tree": ("revolute",), "onedof_tree": ("revolute", "prismatic"),
    "floating_onedof_tree": ("revolute", "prismatic"), "mixed_tree": ("revolute", "prismatic", "sixdof", "fixed"),
    "all_kinds_tree": ("revolute", "prismatic", "sixdof", "fixed", "planar", "spherical"),
}


@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_fp32_entry_points_against_the_fp64_oracle(torch_cuda, family):
    """mh_rnea_f32 / mh_aba_f32 / mh_crba_f32 on the reference's random families (ForwardDynamicsCalculatorTest.java:42-280), against
    the fp64 oracle.  u = 2^-24.  RNEA and CRBA are forward computations along paths of at most n bodies: |err| <= 64 n u max|ref|
    (CRBA 16 n u).  ABA: backward error in tau-space <= 64 n u (|tau| + |bias|), forward error <= 16 n u cond_inf(H) per configuration
    (cond from the oracle's H)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(zlib.crc32(("f32" + family).encode()))
    u32, f32 = 2.0 ** -24, torch.float32
    for it in range(4):
        n = int(rng.integers(1, 41))
        if family == "revolute_chain":
            joints = rt.nextJointChain(rng, n, FAMILIES[family])
        elif family.startswith("floating"):
            joints = rt.nextFloatingChain(rng, n, FAMILIES[family], tree=True)
        else:
            joints = rt.nextJointTree(rng, n, FAMILIES[family])
        sys_ = system_of(joints)
        d = sys_.toModelDesc()
        nb = d.n_joints
        hm, om = HipModel(d), OracleModel(d)
        B = int(rng.integers(1, 150))
        q, qd, qdd, tau = rt.nextState(rng, sys_, B)
        g = (0.0, 0.0, -9.81)
        fext = rng.uniform(-1, 1, (B, nb, 6)) if it % 2 else None
        tf = None if fext is None else dev(torch, fext, f32)
        t32 = hm.rnea(dev(torch, q, f32), dev(torch, qd, f32), dev(torch, qdd, f32), g, tf).cpu().numpy()
        assert t32.dtype == np.float32
        close(t32.astype(np.float64), om.rnea(q, qd, qdd, g, fext), f32_forward_tol(nb), label="rnea_f32")
        H_ref = om.crba(q)
        H32 = hm.crba(dev(torch, q, f32)).cpu().numpy()
        assert H32.dtype == np.float32
        close(H32.astype(np.float64), H_ref, f32_forward_tol(nb), label="crba_f32")
        assert np.array_equal(H32 == 0, H_ref == 0)
        if d.nv == 0:
            continue
        a32 = hm.aba(dev(torch, q, f32), dev(torch, qd, f32), dev(torch, tau, f32), g, tf).cpu().numpy().astype(np.float64)
        a_ref = om.aba(q, qd, tau, g, fext)
        bias = om.rnea(q, qd, np.zeros_like(qdd), g, fext)
        scale = np.abs(tau).max() + np.abs(bias).max()
        berr = np.abs(om.rnea(q, qd, a32, g, fext) - tau).max()
        record_parity(berr, f32_aba_backward_tol(nb) * scale, "aba_f32 backward error")
        assert berr <= f32_aba_backward_tol(nb) * scale, (berr, scale)
        conds = np.array([np.linalg.cond(H_ref[k], np.inf) for k in range(B)])
        ferr = np.abs(a32 - a_ref).max(axis=1) / np.maximum(1.0, np.abs(a_ref).max(axis=1))
        record_parity(float((ferr / (conds * u32)).max()), f32_aba_forward_factor(nb), "aba_f32 forward error / (cond_inf(H) u)")
        assert (ferr <= f32_aba_forward_factor(nb) * u32 * conds).all(), (ferr.max(), conds.max())


WRENCH_FAMILIES = {"revolute_chain": ("revolute",), "onedof_tree": ("revolute", "prismatic"), "floating_onedof_tree": ("revolute", "prismatic"),
                   "mixed_tree": ("revolute", "prismatic", "sixdof", "fixed"), "all_kinds_tree": ("revolute", "prismatic", "sixdof", "fixed", "planar", "spherical")}


def _family_system(rng, family, n, table):
    from mecano_amd import random_tools as rt
    if family == "revolute_chain":
        return system_of(rt.nextJointChain(rng, n, table[family]))
    if family.startswith("floating"):
        return system_of(rt.nextFloatingChain(rng, n, table[family], tree=True))
    return system_of(rt.nextJointTree(rng, n, table[family]))


@pytest.mark.parametrize("family", sorted(WRENCH_FAMILIES))
def test_joint_wrenches_and_relative_accelerations(torch_cuda, family):
    """The rest of RigidBodyAccelerationProvider / the wrench getters, after ForwardDynamicsCalculatorTest.java:
    847-901: per-joint wrenches of inverse dynamics (getComputedJointWrench) against the oracle; forward dynamics' joint wrenches
    (getJointWrench) against inverse dynamics' on the accelerations it produced; relative accelerations between random pairs of bodies
    (root body included) against the oracle, with and without velocities, AoS and SoA."""
    torch = torch_cuda
    from mecano_amd import _lib
    from mecano_amd import random_tools as rt
    from mecano_amd.calculators import ForwardDynamicsCalculator, InverseDynamicsCalculator
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(zlib.crc32(("n2" + family).encode()))
    for it in range(4):
        sys_ = _family_system(rng, family, int(rng.integers(2, 30)), WRENCH_FAMILIES)
        d = sys_.toModelDesc()
        nj = d.n_joints
        hm, om = HipModel(d), OracleModel(d)
        B = int(rng.integers(1, 150))
        q, qd, qdd, tau = rt.nextState(rng, sys_, B)
        g = (0.2, -0.1, -9.81)
        fext = rng.uniform(-1, 1, (B, nj, 6)) if it % 2 else None
        tq, tqd, tqdd, ttau = (dev(torch, x) for x in (q, qd, qdd, tau))
        tf = None if fext is None else dev(torch, fext)
        t_ref, w_ref = om.rnea_wrenches(q, qd, qdd, g, fext)
        t_gpu, w_gpu = hm.rnea_joint_wrenches(tq, tqd, tqdd, g, tf)
        close(t_gpu.cpu().numpy(), t_ref, 1e-10, label="tau")
        close(w_gpu.cpu().numpy(), w_ref, 1e-10, label="rnea joint wrenches")
        if d.nv:
            eps = 1e-7 if "mixed" in family or "all_kinds" in family else 1e-9
            a_gpu, w2_gpu = hm.aba_joint_wrenches(tq, tqd, ttau, g, tf)
            a_ref = om.aba(q, qd, tau, g, fext)
            close(a_gpu.cpu().numpy(), a_ref, eps, label="qdd")
            close(w2_gpu.cpu().numpy(), om.rnea_wrenches(q, qd, a_ref, g, fext)[1], eps, label="aba joint wrenches")
        # relative accelerations
        npairs = 12
        base = rng.integers(-1, nj, npairs).astype(np.int32)
        body = rng.integers(-1, nj, npairs).astype(np.int32)
        _, acc, tw = hm.rnea_bodies(tq, tqd, tqdd, g, tf)
        rel = hm.relative_acceleration(tq, acc, tw, base, body, g)
        close(rel.cpu().numpy(), om.relative_acceleration(q, qd, qdd, base, body, g), 1e-10, label="relative acceleration")
        rel0 = hm.relative_acceleration(tq, acc, None, base, body, g, consider_velocities=False)
        # velocities ignored: the plain change of frame of the accelerations the sweep produced (which do contain velocity terms)
        T = lambda x: x.reshape(B, -1).t().contiguous()
        rel_soa = hm.relative_acceleration(T(tq), T(acc), T(tw), base, body, g, layout=_lib.LAYOUT_SOA)
        assert torch.equal(rel_soa.t().reshape(B, npairs, 6), rel)
        _, acc_nv, tw_nv = hm.rnea_bodies(tq, tqd, tqdd, g, tf, consider_coriolis=False)
        rel_nv = hm.relative_acceleration(tq, acc_nv, None, base, body, g, consider_velocities=False)
        close(rel_nv.cpu().numpy(), om.relative_acceleration(q, qd, qdd, base, body, g, consider_coriolis=False), 1e-10,
              label="relative acceleration, velocities ignored")
        assert rel0.shape == rel.shape
    # the calculators' getters, as ForwardDynamicsCalculatorTest.java:847-901 uses them
    sys_ = _family_system(rng, family, 12, WRENCH_FAMILIES)
    joints = sys_.getJointsToConsider()
    q, qd, qdd, _ = (dev(torch, x) for x in rt.nextState(rng, sys_, 40))
    idc, fdc = InverseDynamicsCalculator(sys_), ForwardDynamicsCalculator(sys_)
    for c in (idc, fdc):
        c.setGravitationalAcceleration(-9.81)
    tau = idc.compute(q, qd, qdd, bodies=True, wrenches=True)
    fdc.compute(q, qd, tau, bodies=True, wrenches=True)
    eps = 1e-7 if "mixed" in family or "all_kinds" in family else 1e-9
    for j in joints:
        expected = idc.getComputedJointWrench(j).cpu().numpy()
        close(fdc.getJointWrench(j).cpu().numpy(), expected, eps, label="getJointWrench == getComputedJointWrench")
    bodies = [j.getSuccessor() for j in joints]
    for k in range(5):
        b1, b2 = bodies[int(rng.integers(len(bodies)))], bodies[int(rng.integers(len(bodies)))]
        e = idc.getAccelerationProvider().getRelativeAcceleration(b1, b2).cpu().numpy()
        close(fdc.getAccelerationProvider().getRelativeAcceleration(b1, b2).cpu().numpy(), e, eps, label="getRelativeAcceleration, FD == ID")
    root_rel = idc.getAccelerationProvider().getRelativeAcceleration(sys_.getRootBody(), bodies[-1])
    assert root_rel is not None and root_rel.shape == (40, 6)
    from mecano_amd.multibody import RigidBody
    assert idc.getAccelerationProvider().getRelativeAcceleration(RigidBody("stranger"), bodies[0]) is None


def test_fp32_forms_of_the_wider_entry_points(torch_cuda):
    """mh_rnea_bodies_f32 / mh_aba_bodies_f32 / mh_aba_locked_f32 and the fp32 host-pointer entry points against their fp64 twins
    (forward bounds ~ n u max|ref| with u = 2^-24; forward dynamics by its backward error, see test_fp32_entry_points_against_the_fp64_oracle)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    rng = np.random.default_rng(3232)
    sys_ = system_of(rt.nextFloatingChain(rng, 14, ("revolute", "prismatic"), tree=True))
    d = sys_.toModelDesc()
    hm = HipModel(d)
    B, u32, nb = 300, 2.0 ** -24, d.n_joints
    q, qd, qdd, tau = rt.nextState(rng, sys_, B)
    g = (0.0, 0.0, -9.81)
    f32 = torch.float32
    d64 = [dev(torch, x) for x in (q, qd, qdd, tau)]
    d32 = [dev(torch, x, f32) for x in (q, qd, qdd, tau)]
    t64, acc64, tw64 = hm.rnea_bodies(d64[0], d64[1], d64[2], g)
    t32, acc32, tw32 = hm.rnea_bodies(d32[0], d32[1], d32[2], g)
    assert t32.dtype == f32 and acc32.dtype == f32
    for a, b, name in ((t32, t64, "tau"), (acc32, acc64, "body acc"), (tw32, tw64, "body twist")):
        close(a.double().cpu().numpy(), b.cpu().numpy(), f32_forward_tol(nb), label="rnea_bodies_f32 " + name)
    a64, bacc64, _ = hm.aba_bodies(d64[0], d64[1], t64, g)
    a32, bacc32, _ = hm.aba_bodies(d32[0], d32[1], t32, g)
    back = hm.rnea(d64[0], d64[1], a32.double().contiguous(), g)  # fp64 inverse dynamics of the fp32 answer
    close(back.cpu().numpy(), t64.cpu().numpy(), f32_aba_backward_tol(nb), label="aba_bodies_f32 backward error")
    modes = [1 if k % 3 == 1 else 0 for k in range(nb)]
    hm.set_joint_source_modes(modes)
    ql64, tl64 = hm.aba_locked(d64[0], d64[1], t64, d64[2], g)
    ql32, tl32 = hm.aba_locked(d32[0], d32[1], t32, d32[2], g)
    hm.set_joint_source_modes(None)
    assert ql32.dtype == f32
    back = hm.rnea(d64[0], d64[1], ql32.double().contiguous(), g)
    close(back.cpu().numpy(), tl64.cpu().numpy(), f32_aba_backward_tol(nb), label="aba_locked_f32 backward error")
    # fp32 host-pointer entry points = the fp32 device-pointer ones, bit for bit
    h32 = [x.astype(np.float32) for x in (q, qd, qdd, tau)]
    assert np.array_equal(hm.rnea(h32[0], h32[1], h32[2], g), hm.rnea(d32[0], d32[1], d32[2], g).cpu().numpy())
    assert np.array_equal(hm.aba(h32[0], h32[1], h32[3], g), hm.aba(d32[0], d32[1], d32[3], g).cpu().numpy())
    assert np.array_equal(hm.crba(h32[0]), hm.crba(d32[0]).cpu().numpy())
