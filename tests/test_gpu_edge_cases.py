"""Edge cases of the hot path the reference's code distinguishes and the random-state tests never reach, HIP path
against the oracle (paths relative to /root/reference/src/main/java/us/ihmc/mecano/):

* revolute angles far outside (-pi, pi]: |q| >= 2^19 takes the library sincos on the device (mh_device.h: sincos_slow), below it the
  Cody-Waite reduction -- both sides of the switch and |q| up to 1e7;
* quaternions that are not unit (Euclid's Quaternion.set normalises: the engine normalises on input, include/mecano_hip.h);
* revolute axes within 1e-7 of X / Y / Z (tools/MecanoFactories.java:51, 237-248) -- exactly on a coordinate axis, just inside the
  threshold and just outside it;
* bodies with |m| < 1e-7 (spatial/interfaces/FixedFrameSpatialInertiaBasics.java:174-175);
* zero-length batches on every entry point.
"""
import ctypes

import numpy as np
import pytest

from helpers import close

pytestmark = pytest.mark.gpu
G = (0.3, -0.2, -9.81)


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


def _against_oracle(torch, desc, q, qd, qdd, tau, label, tol=1e-10, crba=True):
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    hm, om = HipModel(desc), OracleModel(desc)
    close(hm.rnea(dev(torch, q), dev(torch, qd), dev(torch, qdd), G).cpu().numpy(), om.rnea(q, qd, qdd, G), tol, label=label + " rnea")
    close(hm.aba(dev(torch, q), dev(torch, qd), dev(torch, tau), G).cpu().numpy(), om.aba(q, qd, tau, G), tol, label=label + " aba")
    if crba:
        close(hm.crba(dev(torch, q)).cpu().numpy(), om.crba(q), tol, label=label + " crba")
    t2, a2 = hm.rnea_aba(dev(torch, q), dev(torch, qd), dev(torch, qdd), dev(torch, tau), G)
    close(t2.cpu().numpy(), om.rnea(q, qd, qdd, G), tol, label=label + " pair tau")
    close(a2.cpu().numpy(), om.aba(q, qd, tau, G), tol, label=label + " pair qdd")
    return hm, om


@pytest.mark.parametrize("which", ["humanoid", "arm7", "tree12"])
def test_revolute_angles_far_outside_one_turn(torch_cuda, which):
    """q on both sides of the device's 2^19 switch between its own range reduction and the library's, and up to 1e7 (a joint that has been
    integrated for a long time without wrapping: MultiBodySystemStateIntegrator never wraps q, tools/MultiBodySystemStateIntegrator.java:433-441)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    rng = np.random.default_rng(5)
    sys_ = {"humanoid": lambda: rt.nextHumanoid(rng), "arm7": lambda: system_of(rt.nextJointChain(rng, 7)),
            "tree12": lambda: system_of(rt.nextJointTree(rng, 12, ("revolute", "prismatic")))}[which]()
    desc = sys_.toModelDesc()
    B = 256
    q, qd, qdd, tau = rt.nextState(rng, sys_, B)
    rev = [int(desc.cfg_indices[k]) for k in _revolute_cfg_slots(desc)]
    special = np.array([2.0 ** 19, -(2.0 ** 19), np.nextafter(2.0 ** 19, 0), -np.nextafter(2.0 ** 19, 0), 2.0 ** 19 + 0.5, 1.0e7, -1.0e7, 12345678.9,
                        -9876543.21, 1.0e6 * np.pi, 524287.99999, 3.0e5])
    for b in range(B):
        for k, c in enumerate(rev):
            if b < 64:
                q[b, c] = special[(b + k) % len(special)] + (0.01 * b if b % 2 else 0.0)
            elif b < 160:
                q[b, c] = rng.uniform(-1.0e7, 1.0e7)
            # (the rest keeps its draw from (-pi, pi]: both paths inside one wave)
    _against_oracle(torch, desc, q, qd, qdd, tau, f"large angles {which}")


def _revolute_cfg_slots(desc):
    """positions in desc.cfg_indices (concatenated joint by joint) that belong to revolute joints"""
    out, pos = [], 0
    for t in desc.joint_type:
        n = {0: 1, 1: 1, 2: 7, 3: 0, 4: 3, 5: 4}[int(t)]
        if int(t) == 0:
            out.append(pos)
        pos += n
    return out


def test_quaternions_that_are_not_unit(torch_cuda):
    """SixDoF and spherical joints: quaternions scaled by 1e-3 ... 1e3 (and the same unit quaternion beside them): identical outputs, and the
    oracle's (which normalises where Euclid's Quaternion.set does)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    rng = np.random.default_rng(6)
    for sys_ in (rt.nextHumanoid(rng), system_of(rt.nextJointTree(rng, 9, ("revolute", "spherical", "sixdof")))):
        desc = sys_.toModelDesc()
        B = 192
        q, qd, qdd, tau = rt.nextState(rng, sys_, B)
        q_unit = q.copy()
        pos = 0
        for t in desc.joint_type:
            n = {0: 1, 1: 1, 2: 7, 3: 0, 4: 3, 5: 4}[int(t)]
            if int(t) in (2, 5):
                cols = [int(c) for c in desc.cfg_indices[pos:pos + 4]]
                scale = 10.0 ** rng.uniform(-3, 3, size=B)
                scale[::7] = 1.0
                q[:, cols] *= scale[:, None]
            pos += n
        hm, om = _against_oracle(torch, desc, q, qd, qdd, tau, "non-unit quaternions", crba=True)
        # the scaled and the unit quaternion are the same rotation: same efforts to rounding
        a = hm.rnea(dev(torch, q), dev(torch, qd), dev(torch, qdd), G)
        b = hm.rnea(dev(torch, q_unit), dev(torch, qd), dev(torch, qdd), G)
        assert (a - b).abs().max().item() <= 1e-10 * max(1.0, b.abs().max().item())


def _chain_with_axes(rng, axes, tiny_mass_bodies=(), tiny=5.0e-8):
    from mecano_amd import random_tools as rt
    from mecano_amd.multibody import RigidBody, RevoluteJoint
    root = RigidBody("root")
    pred, joints = root, []
    for i, axis in enumerate(axes):
        j = RevoluteJoint(f"j{i}", pred, None if pred.isRootBody() else rt.nextRigidBodyTransform(rng), axis)
        J = rt.nextSymmetricPositiveDefiniteMatrix3D(rng)
        mass = 0.1 + rng.uniform()
        if i in tiny_mass_bodies:
            mass, J = tiny, J * tiny
        pred = RigidBody(f"b{i}", j, J, mass, centerOfMassOffset=rt.nextVector3D(rng))
        joints.append(j)
    return system_of(joints)


def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v)


def test_revolute_axes_at_and_around_the_coordinate_axes(torch_cuda):
    """tools/MecanoFactories.java:237-248: an axis that geometricallyEquals X, Y or Z within 1e-7 gets the roll / pitch / yaw closed form.
    Exactly on the axis and outside the threshold the closed form and the axis-angle form are the same rotation about the same axis: 1e-10.
    INSIDE the threshold but not on the axis the reference rotates about the exact coordinate axis while its unit twist keeps the axis as
    given -- a model that is inconsistent with itself at the 1e-7 level; the oracle reproduces it, the device keeps ONE axis for both
    (conscious divergences): the two then differ by at most the axis error times the scale of the result, asserted as such."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    rng = np.random.default_rng(8)
    exact = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1), (1, 0, 0)]
    outside = [_unit((1, 3e-7, 0)), _unit((0, 1, -2.5e-7)), _unit((2e-7, 2e-7, 1)), _unit((1, 0, 1.5e-7)), _unit((1e-6, 1, 0)), _unit((0, 4e-7, 1)),
               _unit((1, 1e-3, 0))]
    inside = [_unit((1, 3e-8, -4e-8)), _unit((5e-8, 1, 0)), _unit((0, -6e-8, 1)), _unit((1, 0, 9e-8)), _unit((2e-8, 1, 2e-8)), _unit((7e-8, 0, 1)),
              _unit((1, -5e-8, 5e-8))]
    for label, axes, tol in (("axes exactly on X / Y / Z", exact, 1e-10), ("axes just outside 1e-7", outside, 1e-10)):
        sys_ = _chain_with_axes(rng, axes)
        q, qd, qdd, tau = rt.nextState(rng, sys_, 128)
        _against_oracle(torch, sys_.toModelDesc(), q, qd, qdd, tau, label, tol)
    sys_ = _chain_with_axes(rng, inside)
    desc = sys_.toModelDesc()
    q, qd, qdd, tau = rt.nextState(rng, sys_, 128)
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    hm, om = HipModel(desc), OracleModel(desc)
    # the boundary says so (include/mecano_hip.h, MH_WARN_NEAR_COORDINATE_AXIS): bit, text naming a joint, text left in mh_last_error
    from mecano_amd import _lib
    assert hm.warnings == 1 and "within 1e-7" in hm.warning_text and "joint" in hm.warning_text, (hm.warnings, hm.warning_text)
    for label, axes in (("exact", exact), ("outside", outside)):
        assert HipModel(_chain_with_axes(np.random.default_rng(8), axes).toModelDesc()).warnings == 0, label
    assert HipModel(rt.nextHumanoid(np.random.default_rng(43)).toModelDesc()).warnings == 0
    HipModel(desc)
    assert _lib.load().mh_last_error().decode().startswith("warning: joint")
    t_dev, t_ref = hm.rnea(dev(torch, q), dev(torch, qd), dev(torch, qdd), G).cpu().numpy(), om.rnea(q, qd, qdd, G)
    gap = np.abs(t_dev - t_ref).max() / max(1.0, np.abs(t_ref).max())
    assert gap <= 4.0e-7, gap          # the axis error (<= 1e-7) times a handful of transforms
    # ... and it IS that divergence, nothing else: with the axes snapped onto the coordinate axes both sides agree to 1e-10
    snapped = _chain_with_axes(np.random.default_rng(8), [np.round(a) for a in inside])
    # (same random stream as a chain built with `inside` would have used is not needed: a fresh consistent model is compared on both sides)
    q, qd, qdd, tau = rt.nextState(rng, snapped, 128)
    _against_oracle(torch, snapped.toModelDesc(), q, qd, qdd, tau, "axes snapped", 1e-10)


def test_bodies_with_tiny_mass(torch_cuda):
    """|m| < 1e-7.  RNEA and ABA never add inertias: 1e-10 against the oracle whatever the masses.  CRBA: SpatialInertia.add renormalises the
    centre of mass only if |m| >= 1e-7 (FixedFrameSpatialInertiaBasics.java:174-175) -- a guard against 0 / 0 that leaves m c in the place
    of c when a COMPOSITE mass stays under the threshold; a light body under an ordinary one never triggers it (1e-10), two light bodies at
    the end of a chain do: there the device keeps (m, m c, I) and stays physically consistent (H e_j = RNEA(q, 0, e_j) at zero gravity, to
    1e-10), the oracle follows the reference, and the two differ by less than the masses involved (conscious divergences)."""
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(9)
    axes = [rt.nextUnitVector3D(rng) for _ in range(6)]
    # a light body in the middle and a light leaf under an ordinary body: no composite under the threshold
    sys_ = _chain_with_axes(rng, axes, tiny_mass_bodies=(2, 5))
    q, qd, qdd, tau = rt.nextState(rng, sys_, 128)
    tau[:, 5] *= 1e-7  # (efforts on a 5e-8 kg leaf: keep its acceleration finite-sized)
    _against_oracle(torch, sys_.toModelDesc(), q, qd, qdd, tau, "light bodies, ordinary composites", 1e-10, crba=False)
    hm, om = HipModel(sys_.toModelDesc()), OracleModel(sys_.toModelDesc())
    assert hm.warnings == 0, hm.warning_text  # no composite under the threshold: nothing to warn about
    close(hm.crba(dev(torch, q)).cpu().numpy(), om.crba(q), 1e-10, label="light leaf crba")
    # two light bodies at the end: the composite of the last two (8e-8) is under the threshold (two of 5e-8 add up to 1e-7 exactly, which
    # the reference still renormalises: |m| >= 1e-7)
    assert HipModel(_chain_with_axes(np.random.default_rng(9), axes, tiny_mass_bodies=(4, 5)).toModelDesc()).warnings == 0
    sys2 = _chain_with_axes(rng, axes, tiny_mass_bodies=(4, 5), tiny=4.0e-8)
    desc2 = sys2.toModelDesc()
    hm2, om2 = HipModel(desc2), OracleModel(desc2)
    assert hm2.warnings == 2 and "1e-7" in hm2.warning_text, (hm2.warnings, hm2.warning_text)  # MH_WARN_TINY_COMPOSITE_MASS
    q, qd, qdd, tau = rt.nextState(rng, sys2, 64)
    close(hm2.rnea(dev(torch, q), dev(torch, qd), dev(torch, qdd), G).cpu().numpy(), om2.rnea(q, qd, qdd, G), 1e-10, label="two light bodies rnea")
    H = hm2.crba(dev(torch, q)).cpu().numpy()
    zero = np.zeros_like(qd)
    for j in range(desc2.nv):
        e = zero.copy()
        e[:, j] = 1.0
        col = hm2.rnea(dev(torch, q), dev(torch, zero), dev(torch, e), (0.0, 0.0, 0.0)).cpu().numpy()
        assert np.abs(H[:, :, j] - col).max() <= 1e-10 * max(1.0, np.abs(col).max()), j
    assert np.abs(H - om2.crba(q)).max() <= 1e-6  # the reference's un-renormalised composite: differs by less than the masses it concerns


def test_zero_length_batches_on_every_entry_point(torch_cuda):
    """B = 0 is MH_OK and touches nothing, with NULL data pointers, on every compute entry point of include/mecano_hip.h (device- and
    host-pointer forms, fp64 and fp32)."""
    torch = torch_cuda
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.engine import HipModel
    lib = _lib.load()
    hm = HipModel(rt.nextHumanoid(np.random.default_rng(10)).toModelDesc())
    h = hm._h
    g = (ctypes.c_double * 3)(0.0, 0.0, -9.81)
    N = None
    calls = {
        "mh_rnea_f64": (h, 0, N, N, N, g, N, N, N), "mh_aba_f64": (h, 0, N, N, N, g, N, N, N), "mh_crba_f64": (h, 0, N, N, N),
        "mh_rnea_f32": (h, 0, N, N, N, g, N, N, N), "mh_aba_f32": (h, 0, N, N, N, g, N, N, N), "mh_crba_f32": (h, 0, N, N, N),
        "mh_rnea_aba_f64": (h, 0, N, N, N, N, g, N, N, N, N), "mh_rnea_aba_f32": (h, 0, N, N, N, N, g, N, N, N, N),
        "mh_rnea_crba_f64": (h, 0, N, N, N, g, N, N, N, N),
        "mh_aba_locked_f64": (h, 0, N, N, N, N, g, N, N, N, N), "mh_aba_locked_f32": (h, 0, N, N, N, N, g, N, N, N, N),
        "mh_rnea_bodies_f64": (h, 0, N, N, N, g, N, N, N, N, N), "mh_aba_bodies_f64": (h, 0, N, N, N, g, N, N, N, N, N),
        "mh_rnea_bodies_f32": (h, 0, N, N, N, g, N, N, N, N, N), "mh_aba_bodies_f32": (h, 0, N, N, N, g, N, N, N, N, N),
        "mh_rnea_joint_wrenches_f64": (h, 0, N, N, N, g, N, N, N, N), "mh_aba_joint_wrenches_f64": (h, 0, N, N, N, g, N, N, N, N),
        "mh_crba_coriolis_f64": (h, 0, N, N, N, N, N), "mh_crba_coriolis_f32": (h, 0, N, N, N, N, N),
        "mh_centroidal_f64": (h, 0, N, N, N, 0, N, N, N, N), "mh_centroidal_f32": (h, 0, N, N, N, 0, N, N, N, N),
        "mh_integrate_f64": (h, 0, 1e-3, N, N, N, N, N, N, N), "mh_integrate_f32": (h, 0, 1e-3, N, N, N, N, N, N, N),
        "mh_aba_integrate_f64": (h, 0, 1e-3, N, N, N, g, N, N, N, N, N),
        "mh_regressor_f64": (h, 0, N, N, N, g, N, 0, N), "mh_regressor_f32": (h, 0, N, N, N, g, N, 0, N),
        "mh_rnea_f64_host": (h, 0, N, N, N, g, N, N, N), "mh_aba_f64_host": (h, 0, N, N, N, g, N, N, N), "mh_crba_f64_host": (h, 0, N, N, N),
        "mh_rnea_f32_host": (h, 0, N, N, N, g, N, N, N), "mh_aba_f32_host": (h, 0, N, N, N, g, N, N, N), "mh_crba_f32_host": (h, 0, N, N, N),
        "mh_rnea_aba_f64_host": (h, 0, N, N, N, N, g, N, N, N, N),
        "mh_crba_coriolis_f64_host": (h, 0, N, N, N, N, N), "mh_centroidal_f64_host": (h, 0, N, N, N, 0, N, N, N, N),
    }
    for name, args in calls.items():
        assert getattr(lib, name)(*args) == _lib.MH_OK, (name, lib.mh_last_error())
    base = np.zeros(4, dtype=np.int32)
    assert lib.mh_relative_acceleration_f64(h, 0, N, N, N, g, 0, base.ctypes.data, base.ctypes.data, N, N) == _lib.MH_OK
    # the Python mirror: empty tensors in, empty tensors out
    e = lambda n: torch.empty((0, n), dtype=torch.float64, device="cuda")
    assert hm.rnea(e(hm.nq), e(hm.nv), e(hm.nv), G).shape == (0, hm.nv)
    assert hm.aba(e(hm.nq), e(hm.nv), e(hm.nv), G).shape == (0, hm.nv)
    assert hm.crba(e(hm.nq)).shape == (0, hm.nv, hm.nv)
    assert lib.mh_reserve(h, 0) == _lib.MH_OK and lib.mh_model_check(h, None, None) == _lib.MH_OK


def test_inverse_dynamics_loop_that_requests_rows_ahead(torch_cuda):
    """The persistent inverse-dynamics loop of device-filling batches (spec_zvb_bias_kernel<.., BIAS = false>; default beyond two groups of
    64 configurations per CU, forced here by MH_RNEA_AHEAD=2 on a ragged batch of a few groups): external wrenches, the Coriolis /
    acceleration switches (InverseDynamicsCalculator.java:291-306) and index maps that are dense but not the identity -- against the
    oracle, and bit for bit against the tree-split kernel's own loop (MH_RNEA_AHEAD=0), whose walk it shares."""
    import os
    torch = torch_cuda
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    from mecano_amd.multibody import ModelDesc
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(90210)
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    d = sys_.toModelDesc()
    # the same robot with the joints' rows dealt out in reverse listing order (a custom JointMatrixIndexProvider: dense, not the identity)
    nd = [6 if t == 2 else (0 if t == 3 else 1) for t in d.joint_type]
    ncfg = [7 if t == 2 else (0 if t == 3 else 1) for t in d.joint_type]
    ofs_v, ofs_q = np.concatenate([[0], np.cumsum(nd)]), np.concatenate([[0], np.cumsum(ncfg)])  # where joint j's entries sit by default
    dof_idx, cfg_idx, nv, nq = np.zeros(d.nv, dtype=np.int32), np.zeros(d.nq, dtype=np.int32), 0, 0
    for j in reversed(range(d.n_joints)):
        dof_idx[ofs_v[j]:ofs_v[j + 1]] = nv + np.arange(nd[j])
        cfg_idx[ofs_q[j]:ofs_q[j + 1]] = nq + np.arange(ncfg[j])
        nv, nq = nv + nd[j], nq + ncfg[j]
    assert np.array_equal(d.dof_indices, np.arange(d.nv)) and np.array_equal(d.cfg_indices, np.arange(d.nq))
    d2 = ModelDesc(d.n_joints, d.nq, d.nv, d.parent, d.joint_type, d.axis, d.X_before, d.X_com, d.inertia_J, d.inertia_mass, d.inertia_com, dof_idx, cfg_idx)
    B = 3 * 64 + 17
    q, qd, qdd, _ = rt.nextState(rng, sys_, B)
    fext = rng.uniform(-2, 2, (B, d.n_joints, 6))
    # the permuted model's matrices: column cfg_idx[c] of q2 is column c of q
    q2, qd2, qdd2 = np.zeros_like(q), np.zeros_like(qd), np.zeros_like(qdd)
    q2[:, cfg_idx], qd2[:, dof_idx], qdd2[:, dof_idx] = q, qd, qdd
    results = {}
    try:
        for mode in ("2", "0"):
            os.environ["MH_RNEA_AHEAD"] = mode
            out = []
            for desc, (a, b, c) in ((d, (q, qd, qdd)), (d2, (q2, qd2, qdd2))):
                hm = HipModel(desc)
                assert hm.kernel_variant.startswith("topo:"), hm.kernel_variant
                ta, tb, tc, tf = dev(torch, a), dev(torch, b), dev(torch, c), dev(torch, fext)
                out.append(hm.rnea(ta, tb, tc, G))
                out.append(hm.rnea(ta, tb, tc, G, tf))
                for cc, ca in ((False, True), (True, False), (False, False)):
                    out.append(hm.rnea(ta, tb, tc, G, tf, consider_coriolis=cc, consider_accelerations=ca))
            results[mode] = out
    finally:
        os.environ.pop("MH_RNEA_AHEAD", None)
    for x, y in zip(results["2"], results["0"]):
        assert torch.equal(x, y)
    k = 0
    for desc, (a, b, c) in ((d, (q, qd, qdd)), (d2, (q2, qd2, qdd2))):
        om = OracleModel(desc)
        close(results["2"][k].cpu().numpy(), om.rnea(a, b, c, G), 1e-10, label="rows ahead"); k += 1
        close(results["2"][k].cpu().numpy(), om.rnea(a, b, c, G, fext), 1e-10, label="rows ahead, wrenches"); k += 1
        for cc, ca in ((False, True), (True, False), (False, False)):
            close(results["2"][k].cpu().numpy(), om.rnea(a, b, c, G, fext, cc, ca), 1e-10, label=f"rows ahead, switches {cc} {ca}"); k += 1


def test_a_consumer_that_gives_up_writes_nan_and_reports_at_the_next_synchronisation(torch_cuda, tmp_path):
    """The one failure of an asynchronous call (include/mecano_hip.h, "Asynchronous failures"): a bias-split forward dynamics
    launch whose inertia job waits longer than MH_ZV_WAIT_MS for its bias job.  It cannot be provoked on a healthy device, so ONE code
    object is built here with -DMH_ZV_TEST_NO_FLAG (a bias job that never stores its flags; minimal kernel set, into a scratch directory,
    loaded through MH_SPEC_DIR with the create-time self-check off) and the contract is checked on it: the call returns MH_OK, every row
    of the output is NaN -- never numbers formed from stale scratch --, mh_model_check reports MH_ERR_HIP once, the error word is then
    clear again, and a call that does not take the bias split (MH_ZV=0 view of the same model) is unaffected."""
    import os
    import subprocess
    torch = torch_cuda
    from mecano_amd import _lib, build as mbuild, random_tools as rt
    from mecano_amd.engine import HipModel
    from oracle.cpu_oracle import OracleModel
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    d = sys_.toModelDesc()
    key, parents, kinds = mbuild.topology_of(d)
    out = tmp_path / os.path.basename(mbuild.spec_path(key))
    subprocess.check_call([mbuild.hipcc()] + mbuild.SPEC_FLAGS + mbuild.spec_defines(parents, kinds, ("-DMH_SPEC_MINIMAL", "-DMH_ZV_TEST_NO_FLAG"))
                          + ["-o", str(out), mbuild.SPEC_SOURCE], stderr=subprocess.DEVNULL)
    B = 200  # four groups of 64 configurations, the last one ragged
    q, qd, _, tau = rt.nextState(np.random.default_rng(5), sys_, B)
    keys = ("MH_SPEC_DIR", "MH_SPEC_SELFCHECK", "MH_ZV_WAIT_MS", "MH_ZV")
    try:
        os.environ.update({"MH_SPEC_DIR": str(tmp_path), "MH_SPEC_SELFCHECK": "0", "MH_ZV_WAIT_MS": "3"})
        hm = HipModel(d)
        assert hm.kernel_variant.startswith("topo:"), hm.kernel_variant
        a = hm.aba(dev(torch, q), dev(torch, qd), dev(torch, tau), G)  # returns MH_OK: the failure happens on the device, later
        torch.cuda.synchronize()
        assert bool(torch.isnan(a).all()), "rows of a group whose bias efforts never came must be NaN"
        with pytest.raises(_lib.MecanoHipError) as e:
            hm.check()
        assert "gave up waiting" in str(e.value)
        hm.check()  # reported once: the word is clear again
        # Round 5 (the limb columns signal themselves, mh_zv_kernels.h "Stage one without a flag"): a give-up poisons the context -- a
        # producer that publishes late must not be taken for the next launch's -- until the library has seen it, waited for the device and
        # refilled the hand-off matrix with sentinels, which it does at the context's next bias-split launch.  This producer never
        # publishes, so that launch gives up as well: NaN rows again, reported once again.
        a = hm.aba(dev(torch, q), dev(torch, qd), dev(torch, tau), G)
        torch.cuda.synchronize()
        assert bool(torch.isnan(a).all())
        with pytest.raises(_lib.MecanoHipError):
            hm.check()
        hm.check()
        os.environ["MH_ZV"] = "0"  # the same code object without the bias split: the tree-split kernels need no hand-off
        hm2 = HipModel(d)
        close(hm2.aba(dev(torch, q), dev(torch, qd), dev(torch, tau), G).cpu().numpy(), OracleModel(d).aba(q, qd, tau, G), 1e-10, label="tree split")
        hm2.check()
    finally:
        for k in keys:
            os.environ.pop(k, None)


def test_bias_split_on_every_registered_shape_and_under_graph_replay():
    """tools/check_zv.py all: the bias-split forward dynamics (two-stage hand-off where the index maps are the identity) forced at every batch
    size (MH_ZV=2) on the five registered tree shapes -- staged trunks, a plain split, a chain without one -- at B = 1 ... 12 000 with
    ragged last groups, three launches each, against the oracle; and five replays of a captured mh_rnea_aba_f64 launch (the same epoch every
    time: the consumer must have put both flags back to zero), bit for bit.  A subprocess: the tool pins MH_ZV before the library loads."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("MH_")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_zv.py"), "all"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.startswith("worst scaled error")]
    assert last and float(last[-1].split()[-1]) < 1e-9, r.stdout[-500:]


def test_pair_call_refuses_outputs_that_overlap_its_inputs(torch_cuda):
    """mh_rnea_aba_f64 / _f32 run their two algorithms concurrently (side by side on different workgroups, or phase by phase in one): an
    output that overlaps an input of the other algorithm would be read half-written.  The call says so (MH_ERR_INVALID_ARGUMENT) instead
    of computing something; separate buffers work, and the single calls remain the way to compute in place."""
    torch = torch_cuda
    from mecano_amd import _lib
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(sys_.toModelDesc())
    for dt in (torch.float64, torch.float32):
        q, qd, qdd, tau = (dev(torch, x, dt) for x in rt.nextState(np.random.default_rng(3), sys_, 256))
        o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
        hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, G)()
        assert torch.isfinite(o1).all() and torch.isfinite(o2).all()
        for bad in ((tau, o2), (o1, qdd), (o1, o1), (qd, o2)):
            with pytest.raises(_lib.MecanoHipError) as e:
                hm.bind_rnea_aba(q, qd, qdd, tau, bad[0], bad[1], G)()
            assert "overlap" in str(e.value)
        # in place, one after the other
        t_ref = hm.rnea(q, qd, qdd, G)
        assert torch.equal(t_ref, o1) or (t_ref - o1).abs().max().item() <= 1e-4

