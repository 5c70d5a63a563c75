"""Joint torque regressor (JointTorqueRegressorCalculator, a caller of the inverse dynamics): oracle pins on the CPU, parity through the
C-ABI on the GPU.

The reference holds no golden vectors for it; its tests pin the identity  Y(q, qd, qdd) pi = tau_inverse_dynamics  to 1e-12 on random
chains and trees (JointTorqueRegressorCalculatorTest.java:37-66, 98-135, 172-214, 255-292, 329-371), with joint accelerations or
Coriolis / centrifugal terms switched off on both sides (:479-518, :521-560), and the unit spatial-inertia bases (:1081-1140).  Restated
here on the oracle (which evaluates the regressor the way the reference does: one inverse dynamics per body and basis) and on the HIP
kernel (one sweep, all bases together)."""
import dataclasses
import zlib

import numpy as np
import pytest

from helpers import close

FAMILIES = {
    "onedof_chain": lambda rt, rng, n: rt.nextJointChain(rng, n, ("revolute", "prismatic")),
    "floating_onedof_chain": lambda rt, rng, n: rt.nextFloatingChain(rng, n, ("revolute", "prismatic")),
    "onedof_tree": lambda rt, rng, n: rt.nextJointTree(rng, n, ("revolute", "prismatic")),
    "floating_onedof_tree": lambda rt, rng, n: rt.nextFloatingChain(rng, n, ("revolute", "prismatic"), tree=True),
    "mixed_tree": lambda rt, rng, n: rt.nextJointTree(rng, n, ("revolute", "prismatic", "sixdof", "fixed", "planar", "spherical")),
}
G = (0.0, 0.0, -9.81)


def system_of(joints):
    from mecano_amd.multibody import MultiBodySystem
    return MultiBodySystem.toMultiBodySystemInput(joints[0].getPredecessor())


def with_com_offsets(rng, d):
    """The same system with centre-of-mass offsets in the body-fixed frames (Mecano's generators leave them at zero)."""
    return dataclasses.replace(d, inertia_com=rng.uniform(-0.3, 0.3, 3 * d.n_joints))


def first_moment_parameters(om):
    """pi for first_moment_columns = 1: slots 1..3 hold m c instead of c."""
    pi = om.parameter_vector().reshape(-1, 10).copy()
    pi[:, 1:4] *= pi[:, 0:1]
    return pi.reshape(-1)


# ------------------------------------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_oracle_regressor_times_parameters_is_inverse_dynamics(family):
    """JointTorqueRegressorCalculatorTest.java:98-135, 172-214, 255-292, 329-371 (+ :479-560 for the two switches)."""
    from mecano_amd import random_tools as rt
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(zlib.crc32(family.encode()))
    for it in range(3):
        sys_ = system_of(FAMILIES[family](rt, rng, int(rng.integers(1, 12))))
        om = OracleModel(sys_.toModelDesc())
        q, qd, qdd, _ = rt.nextState(rng, sys_, 3)
        pi = om.parameter_vector()
        for cor, acc in ((True, True), (True, False), (False, True)):
            Y = om.regressor(q, qd, qdd, G, cor, acc)
            close(Y @ pi, om.rnea(q, qd, qdd, G, None, cor, acc), 1e-11, label=f"{family} cor={cor} acc={acc}")
            # the reference's centre-of-mass bases sit on a body of zero mass: with a twist those columns are exactly zero (without one
            # computeDynamicMoment leaves c x a unscaled, tools/MecanoTools.java:650-692 -- the oracle follows it, and so does the kernel)
            if cor:
                assert not Y.reshape(3, om.nv, om.n, 10)[..., 1:4].any()


def test_oracle_spatial_inertia_bases():
    """JointTorqueRegressorCalculatorTest.java:1081-1140: the ten bases are unit (mass | centre-of-mass offset | symmetric moment) inertias."""
    from oracle.cpu_oracle import OracleModel
    for k in range(10):
        J, m, c = OracleModel._basis_inertia(k)
        assert np.array_equal(J, J.T)
        assert (m, np.abs(c).sum(), np.abs(np.triu(J)).sum()) == ((1.0, 0.0, 0.0) if k == 0 else ((0.0, 1.0, 0.0) if k <= 3 else (0.0, 0.0, 1.0)))


# ------------------------------------------------------------------------------------------------ HIP path (GPU)
@pytest.fixture(scope="module")
def torch_cuda(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def dev(torch, x, dtype=None):
    return torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dtype or torch.float64)


@pytest.mark.gpu
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_regressor_matches_oracle_and_inverse_dynamics(torch_cuda, family):
    """mh_regressor_f64 against the oracle's column-by-column evaluation (1e-10), and Y pi against mh_rnea_f64 on a bigger batch,
    with both switches, AoS and SoA state matrices."""
    torch = torch_cuda
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(zlib.crc32(family.encode()) + 1)
    for it in range(4):
        sys_ = system_of(FAMILIES[family](rt, rng, int(rng.integers(1, 31))))  # JointTorqueRegressorCalculatorTest.java:107: 1..30 joints
        d = sys_.toModelDesc()
        om, hm = OracleModel(d), HipModel(d)
        pi = om.parameter_vector()
        B = int(rng.integers(65, 400))
        q, qd, qdd, _ = rt.nextState(rng, sys_, B)
        tq, tqd, tqdd = (dev(torch, x) for x in (q, qd, qdd))
        for cor, acc in ((True, True), (True, False), (False, True)):
            Y = hm.regressor(tq, tqd, tqdd, G, consider_coriolis=cor, consider_accelerations=acc)
            assert tuple(Y.shape) == (B, d.nv, 10 * d.n_joints)
            Yh = Y.cpu().numpy()
            close(Yh[:4], om.regressor(q[:4], qd[:4], qdd[:4], G, cor, acc), 1e-10, label=f"{family} Y cor={cor} acc={acc}")
            tau = hm.rnea(tq, tqd, tqdd, G, consider_coriolis=cor, consider_accelerations=acc).cpu().numpy()
            close(Yh @ pi, tau, 1e-10, label=f"{family} Y pi cor={cor} acc={acc}")
        Ys = hm.regressor(tq.t().contiguous(), tqd.t().contiguous(), tqdd.t().contiguous(), G, layout=_lib.LAYOUT_SOA)
        assert tuple(Ys.shape) == (d.nv, 10 * d.n_joints, B) and torch.equal(Ys.permute(2, 0, 1), hm.regressor(tq, tqd, tqdd, G))


@pytest.mark.gpu
def test_regressor_humanoid_first_moments_and_fp32(torch_cuda):
    """The humanoid of the benchmark (every entry against the oracle on a sample; zero rows for joints that do not support a body); a
    copy of it with centre-of-mass offsets: first_moment_columns = 1 makes tau linear in (m, m c, J) -- Y pi' = inverse dynamics -- and
    each first-moment column is the difference of two inverse dynamics (unit mass with and without a unit offset); fp32 within
    64 n u max|Y|."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(4242)
    sys_ = rt.nextHumanoid(rng)
    d = sys_.toModelDesc()
    om, hm = OracleModel(d), HipModel(d)
    B = 4096
    q, qd, qdd, _ = rt.nextState(rng, sys_, B)
    tq, tqd, tqdd = (dev(torch, x) for x in (q, qd, qdd))
    Y = hm.regressor(tq, tqd, tqdd, G)
    idx = np.arange(0, B, B // 4)[:4]
    close(Y[torch.as_tensor(idx, device="cuda")].cpu().numpy(), om.regressor(q[idx], qd[idx], qdd[idx], G), 1e-10, label="humanoid Y")
    close((Y @ dev(torch, om.parameter_vector())).cpu().numpy(), hm.rnea(tq, tqd, tqdd, G).cpu().numpy(), 1e-10, label="humanoid Y pi")
    # a hand's parameters load the arm, the torso and the pelvis only
    parent = np.asarray(d.parent)
    leaf = int(max(range(d.n_joints), key=lambda j: (np.sum(parent == j) == 0, j)))
    support, j = set(), leaf
    while j >= 0:
        support.add(j)
        j = int(parent[j])
    rows = np.concatenate([np.asarray(d.dof_indices[sum(_ndof(d, i) for i in range(k)):][:_ndof(d, k)]) for k in range(d.n_joints) if k not in support])
    assert not Y[:, torch.as_tensor(rows, device="cuda"), 10 * leaf:10 * leaf + 10].any()

    # centre-of-mass offsets: the reference's columns stay zero; the first-moment form is linear in (m, m c, J)
    dc = with_com_offsets(rng, d)
    omc, hmc = OracleModel(dc), HipModel(dc)
    Yc = hmc.regressor(tq[:512], tqd[:512], tqdd[:512], G, first_moment_columns=True)
    tau_c = hmc.rnea(tq[:512], tqd[:512], tqdd[:512], G).cpu().numpy()
    close((Yc @ dev(torch, first_moment_parameters(omc))).cpu().numpy(), tau_c, 1e-10, label="com offsets: Y pi'")
    close(tau_c[:8], omc.rnea(q[:8], qd[:8], qdd[:8], G), 1e-10, label="com offsets: rnea")
    Y0 = hmc.regressor(tq[:512], tqd[:512], tqdd[:512], G)
    assert not Y0.reshape(512, d.nv, d.n_joints, 10)[..., 1:4].any()
    keep = [k for k in range(10 * d.n_joints) if k % 10 not in (1, 2, 3)]
    close(Y0[:, :, keep].cpu().numpy(), Yc[:, :, keep].cpu().numpy(), 1e-13, label="other columns unchanged")  # (two instantiations: last-bit contraction differences)
    n = d.n_joints
    for body, axis in ((0, 0), (leaf, 1), (n // 2, 2)):
        def unit(c):
            J, mass, com = np.zeros((n, 9)), np.zeros(n), np.zeros((n, 3))
            mass[body], com[body] = 1.0, c
            return OracleModel(dataclasses.replace(dc, inertia_J=J.reshape(-1), inertia_mass=mass, inertia_com=com.reshape(-1))).rnea(q[:8], qd[:8], qdd[:8], G)
        e = np.zeros(3)
        e[axis] = 1.0
        close(Yc[:8, :, 10 * body + 1 + axis].cpu().numpy(), unit(e) - unit(np.zeros(3)), 1e-10, label="first-moment column")

    # fp32
    Y32 = hm.regressor(*(t.float() for t in (tq[:2048], tqd[:2048], tqdd[:2048])), G)
    assert Y32.dtype == torch.float32
    scale = float(Y[:2048].abs().max())
    err = float((Y32.double() - Y[:2048]).abs().max())
    from helpers import f32_forward_tol, record_parity
    bound = f32_forward_tol(d.n_joints) * scale
    record_parity(err, bound, "fp32 Y")
    assert err <= bound, (err, bound)


def _ndof(d, j):
    return {0: 1, 1: 1, 2: 6, 3: 0, 4: 3, 5: 3}[int(d.joint_type[j])]


@pytest.mark.gpu
def test_regressor_calculator_mirror(torch_cuda):
    """The host-side mirror of JointTorqueRegressorCalculator: compute(), getJointTorqueRegressorMatrix(), getParameterVector(),
    getJointTorqueRegressorMatrixBlock(body), the two switches -- the reference's own test shape (JointTorqueRegressorCalculatorTest.java:37-66)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.calculators import InverseDynamicsCalculator, JointTorqueRegressorCalculator
    rng = np.random.default_rng(25)
    sys_ = system_of(rt.nextJointChain(rng, 2, ("revolute", "prismatic")))
    B = 16
    q, qd, qdd, _ = rt.nextState(rng, sys_, B)
    idc = InverseDynamicsCalculator(sys_)
    idc.setGravitationalAcceleration(-9.81)
    idc.compute(q, qd, qdd)
    expected = idc.getJointTauMatrix()
    rc = JointTorqueRegressorCalculator(sys_)
    pi = rc.getParameterVector()
    rc.setGravitationalAcceleration(-9.81)
    rc.compute(q, qd, qdd)
    Y = rc.getJointTorqueRegressorMatrix()
    assert Y.shape == (B, 2, 20) and pi.shape == (20,)
    close(Y @ pi, expected, 1e-10, label="calculator Y pi")
    body = sys_.getJointsToConsider()[1].getSuccessor()
    assert np.array_equal(rc.getJointTorqueRegressorMatrixBlock(body), Y[:, :, 10:20])
    assert np.array_equal(rc.getParameterVectorSlice(body), pi[10:20])
    rc.setConsiderJointAccelerations(False)
    rc.compute(q, qd, qdd)
    idc.setConsiderJointAccelerations(False)
    idc.compute(q, qd, qdd)
    close(rc.getJointTorqueRegressorMatrix() @ pi, idc.getJointTauMatrix(), 1e-10, label="calculator no accelerations")


@pytest.mark.gpu
def test_identification_example_predicts_unseen_torques(torch_cuda):
    """examples/identify_parameters.py: parameters regressed on Y (first-moment columns) reproduce the inverse dynamics of new states."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("identify_parameters", os.path.join(os.path.dirname(__file__), "..", "examples", "identify_parameters.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    fit, err, scale, rank = mod.main(samples=512, verbose=False)
    assert rank < 70 and fit <= 1e-9 * max(1.0, scale) and err <= 1e-8 * max(1.0, scale), (fit, err, scale, rank)


@pytest.mark.gpu
def test_regressor_and_pair_call_argument_errors(torch_cuda):
    """Status codes of mh_regressor_f64 and mh_rnea_crba_f64: an empty batch is MH_OK with NULL pointers, NULL state / output pointers are
    MH_ERR_INVALID_ARGUMENT, a negative batch MH_ERR_BAD_DIMENSION; after mh_reserve the calls succeed at the reserved size."""
    torch = torch_cuda
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.engine import HipModel
    lib = _lib.load()
    sys_ = rt.nextHumanoid(np.random.default_rng(3))
    hm = HipModel(sys_.toModelDesc())
    hm.reserve(300)
    import ctypes
    g = (ctypes.c_double * 3)(0.0, 0.0, -9.81)
    assert lib.mh_regressor_f64(hm._h, 0, None, None, None, g, None, 0, None) == 0
    assert lib.mh_rnea_crba_f64(hm._h, 0, None, None, None, g, None, None, None, None) == 0
    q, qd, qdd, _ = (dev(torch, x) for x in rt.nextState(np.random.default_rng(4), sys_, 300))
    Y = torch.empty((300, hm.nv, 10 * hm.n_joints), dtype=torch.float64, device="cuda")
    H = torch.empty((300, hm.nv, hm.nv), dtype=torch.float64, device="cuda")
    t = torch.empty_like(qd)
    assert lib.mh_regressor_f64(hm._h, 300, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), g, None, 0, None) == 1
    assert b"NULL" in lib.mh_last_error()
    assert lib.mh_regressor_f64(hm._h, 300, q.data_ptr(), None, qdd.data_ptr(), g, None, 0, Y.data_ptr()) == 1
    assert lib.mh_regressor_f64(hm._h, -5, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), g, None, 0, Y.data_ptr()) == 2
    assert lib.mh_rnea_crba_f64(hm._h, 300, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), g, None, None, t.data_ptr(), None) == 1
    assert lib.mh_rnea_crba_f64(hm._h, 300, q.data_ptr(), qd.data_ptr(), None, g, None, None, t.data_ptr(), H.data_ptr()) == 1
    assert lib.mh_regressor_f64(hm._h, 300, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), g, None, 0, Y.data_ptr()) == 0
    assert lib.mh_rnea_crba_f64(hm._h, 300, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), g, None, None, t.data_ptr(), H.data_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(t, hm.rnea(q, qd, qdd)) and torch.equal(H, hm.crba(q))
