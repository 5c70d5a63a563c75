"""The reference's own UNIT tests restated on the oracle's building blocks (oracle/mecano_oracle.c, mo_unit_* entry points).

The reference holds no golden vectors for RNEA / ABA / CRBA; what it does hold are unit-level invariants and
closed forms for the pieces those algorithms are made of.  Each test below names the reference test it restates (paths relative to
/root/reference/src/test/java/us/ihmc/mecano/) and uses its epsilon (1e-12 unless stated).  Random inputs follow the reference
generators' distributions (mass in [0, 1], |c_i| <= 1, principal inertias satisfying the triangle inequality plus the parallel-axis
term of c, tools/MecanoRandomTools.java:572-600); the RNG is numpy's -- bit compatibility with java.util.Random is not claimed.
"""
import ctypes

import numpy as np
import pytest

from oracle import cpu_oracle

EPS = 1.0e-12
ITERATIONS = 300


@pytest.fixture(scope="module")
def lib():
    L = cpu_oracle._load()
    P, D, I = ctypes.c_void_p, ctypes.c_double, ctypes.c_int
    L.mo_unit_dynamic_wrench.argtypes = [P, D, P, P, P, I, P]
    L.mo_unit_dynamic_wrench.restype = None
    L.mo_unit_rigid_apply_transform.argtypes = [P, I, P, P, P]
    L.mo_unit_rigid_apply_transform.restype = None
    L.mo_unit_abi_apply_transform.argtypes = [P, I, P, P, P]
    L.mo_unit_abi_apply_transform.restype = None
    L.mo_unit_abi_from_rigid.argtypes = [P, D, P, P, P, P]
    L.mo_unit_abi_from_rigid.restype = None
    L.mo_unit_abi_to_dense.argtypes = [P, P, P, P]
    L.mo_unit_abi_to_dense.restype = None
    L.mo_unit_rigid_mulv.argtypes = [P, D, P, P, P]
    L.mo_unit_rigid_mulv.restype = None
    L.mo_unit_motion_transform.argtypes = [P, I, P, P]
    L.mo_unit_motion_transform.restype = None
    L.mo_unit_force_transform.argtypes = [P, P, P]
    L.mo_unit_force_transform.restype = None
    L.mo_unit_kinetic_coenergy.argtypes = [P, D, P, P]
    L.mo_unit_kinetic_coenergy.restype = D
    return L


def p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def tilde(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def next_rotation(rng):
    q = rng.normal(size=4)
    x, y, z, s = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * s), 2 * (x * z + y * s)],
                     [2 * (x * y + z * s), 1 - 2 * (x * x + z * z), 2 * (y * z - x * s)],
                     [2 * (x * z - y * s), 2 * (y * z + x * s), 1 - 2 * (x * x + y * y)]])


def next_transform(rng):
    """EuclidCoreRandomTools.nextRigidBodyTransform: random rotation, translation in [-1, 1]^3; packed as R row-major, then p."""
    return np.concatenate([next_rotation(rng).reshape(-1), rng.uniform(-1.0, 1.0, 3)])


def next_spatial_inertia(rng):
    """tools/MecanoRandomTools.java:572-600."""
    mass = rng.uniform(0.0, 1.0)
    c = rng.uniform(-1.0, 1.0, 3)
    a, b = rng.uniform(0.0, 1.0, 2)
    pr = np.array([a, b, rng.uniform(abs(a - b), a + b)])  # triangle inequality
    J = np.diag(pr) + tilde(c) @ tilde(c).T
    return np.ascontiguousarray(J.reshape(-1)), mass, c.copy()


def dense_rigid(J, m, c):
    """SpatialInertiaReadOnly.get (spatial/interfaces/SpatialInertiaReadOnly.java:394-415): [[J, m c~], [-m c~, m 1]]."""
    M = np.zeros((6, 6))
    M[:3, :3] = np.asarray(J).reshape(3, 3)
    M[:3, 3:] = m * tilde(c)
    M[3:, :3] = -m * tilde(c)
    M[3:, 3:] = m * np.eye(3)
    return M


def rigid_apply(lib, X, J, m, c, inverse=False):
    J, c, mm = J.copy(), c.copy(), ctypes.c_double(m)
    lib.mo_unit_rigid_apply_transform(p(X), int(inverse), p(J), ctypes.byref(mm), p(c))
    return J, mm.value, c


def abi_of(lib, J, m, c):
    A, L, C = np.zeros(9), np.zeros(9), np.zeros(9)
    lib.mo_unit_abi_from_rigid(p(J), m, p(c), p(A), p(L), p(C))
    return A, L, C


def abi_apply(lib, X, A, L, C, inverse=False):
    A, L, C = A.copy(), L.copy(), C.copy()
    lib.mo_unit_abi_apply_transform(p(X), int(inverse), p(A), p(L), p(C))
    return A, L, C


def abi_dense(lib, A, L, C):
    M = np.zeros(36)
    lib.mo_unit_abi_to_dense(p(A), p(L), p(C), p(M))
    return M.reshape(6, 6)


def motion(lib, X, v, inverse=False):
    o = np.zeros(6)
    lib.mo_unit_motion_transform(p(X), int(inverse), p(np.ascontiguousarray(v)), p(o))
    return o


def force(lib, X, w):
    o = np.zeros(6)
    lib.mo_unit_force_transform(p(X), p(np.ascontiguousarray(w)), p(o))
    return o


def wrench(lib, J, m, c, acc, tw, general=False):
    o = np.zeros(6)
    lib.mo_unit_dynamic_wrench(p(J), m, p(c), p(acc), p(tw), int(general), p(o))
    return o


@pytest.mark.parametrize("inverse", [False, True])
def test_articulated_inertia_transform_equals_spatial_inertia_transform(lib, inverse):
    """algorithms/ArticulatedBodyInertiaTest.java:25-60 (applyTransform) and :62-104 (applyInverseTransform): a rigid inertia carried
    through two random transforms as an ArticulatedBodyInertia ((A, L, C) blocks, rotate then translate) and as a SpatialInertia
    ((J, m, c), parallel axis) gives the same 6x6 matrix."""
    rng = np.random.default_rng(2552)
    for _ in range(ITERATIONS):
        J, m, c = next_spatial_inertia(rng)
        A, L, C = abi_of(lib, J, m, c)
        for _ in range(2):
            X = next_transform(rng)
            J, m, c = rigid_apply(lib, X, J, m, c, inverse)
            A, L, C = abi_apply(lib, X, A, L, C, inverse)
        assert np.abs(abi_dense(lib, A, L, C) - dense_rigid(J, m, c)).max() <= EPS


def test_articulated_inertia_inverse_transform_undoes_transform(lib):
    """algorithms/ArticulatedBodyInertiaTest.java:106-130 on general (non-rigid) articulated inertias."""
    rng = np.random.default_rng(2553)
    for _ in range(ITERATIONS):
        def spd():
            a = rng.uniform(-1, 1, (3, 3))
            return np.ascontiguousarray((a @ a.T + 0.1 * np.eye(3)).reshape(-1))
        A, L, C = spd(), spd(), np.ascontiguousarray(rng.uniform(-1, 1, 9))
        X = next_transform(rng)
        A2, L2, C2 = abi_apply(lib, X, *abi_apply(lib, X, A, L, C), inverse=True)
        assert np.abs(abi_dense(lib, A2, L2, C2) - abi_dense(lib, A, L, C)).max() <= 4 * EPS


def test_kinetic_coenergy_matrix_form_and_frame_invariance(lib):
    """tools/MecanoToolsTest.java:618-655 (T = 1/2 tw^T I tw with the dense form), spatial/SpatialInertiaBasicsTest.java:76-98 and
    :216-248 (the co-energy does not depend on the frame inertia and twist are expressed in; the mass does not change)."""
    rng = np.random.default_rng(334523)
    for _ in range(ITERATIONS):
        J, m, c = next_spatial_inertia(rng)
        tw = rng.uniform(-1, 1, 6)
        T0 = lib.mo_unit_kinetic_coenergy(p(J), m, p(c), p(tw))
        assert abs(T0 - 0.5 * tw @ dense_rigid(J, m, c) @ tw) <= EPS
        X = next_transform(rng)
        J2, m2, c2 = rigid_apply(lib, X, J, m, c)
        tw2 = motion(lib, X, tw)
        assert abs(lib.mo_unit_kinetic_coenergy(p(J2), m2, p(c2), p(tw2)) - T0) <= 10 * EPS
        assert m2 == m


def test_newtons_law_in_two_frames_and_adjoint_form(lib):
    """spatial/SpatialInertiaBasicsTest.java:250-271 (F = I a computed in another frame and brought back is the same wrench) and
    :273-293 (changeFrame equals Ad^T I Ad with the adjoint of the final -> initial transform)."""
    rng = np.random.default_rng(43)
    for _ in range(ITERATIONS):
        J, m, c = next_spatial_inertia(rng)
        a = rng.uniform(-1, 1, 6)
        X = next_transform(rng)  # initial -> final coordinates
        w0 = np.zeros(6)
        lib.mo_unit_rigid_mulv(p(J), m, p(c), p(a), p(w0))
        J2, m2, c2 = rigid_apply(lib, X, J, m, c)
        a2 = motion(lib, X, a)
        w2 = np.zeros(6)
        lib.mo_unit_rigid_mulv(p(J2), m2, p(c2), p(a2), p(w2))
        R, t = X[:9].reshape(3, 3), X[9:]
        Xinv = np.concatenate([R.T.reshape(-1), -R.T @ t])
        assert np.abs(force(lib, Xinv, w2) - w0).max() <= 10 * EPS
        Ad = np.column_stack([motion(lib, Xinv, e) for e in np.eye(6)])  # motion adjoint of final -> initial
        assert np.abs(Ad.T @ dense_rigid(J, m, c) @ Ad - dense_rigid(J2, m2, c2)).max() <= 10 * EPS
        # a pure rotation keeps |c| and the principal inertias (:295-340)
        Xr = np.concatenate([next_rotation(rng).reshape(-1), np.zeros(3)])
        J3, _, c3 = rigid_apply(lib, Xr, J, m, c)
        assert abs(np.linalg.norm(c3) - np.linalg.norm(c)) <= EPS
        assert np.abs(np.sort(np.linalg.eigvalsh(J3.reshape(3, 3))) - np.sort(np.linalg.eigvalsh(J.reshape(3, 3)))).max() <= 10 * EPS


def test_parallel_axis_translation(lib):
    """tools/MecanoToolsTest.java:218-290 (translateMomentOfInertia): there and back again; J' = J - m (c~ p~ + p~ c~ + p~ p~)."""
    rng = np.random.default_rng(2342)
    eye = np.eye(3).reshape(-1)
    for _ in range(ITERATIONS):
        m = rng.uniform(0, 1)
        c, t = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        J = np.ascontiguousarray(rng.uniform(-1, 1, 9))
        J1, _, c1 = rigid_apply(lib, np.concatenate([eye, t]), J, m, c)
        expected = J.reshape(3, 3) - m * (tilde(c) @ tilde(t) + tilde(t) @ tilde(c) + tilde(t) @ tilde(t))
        assert np.abs(J1.reshape(3, 3) - expected).max() <= EPS
        assert np.abs(c1 - (c + t)).max() == 0.0
        J2, _, c2 = rigid_apply(lib, np.concatenate([eye, -t]), J1, m, c1)
        assert np.abs(J2 - J).max() <= 10 * EPS


def test_symmetric_matrix_rotation(lib):
    """tools/MecanoToolsTest.java:658-694 (transformSymmetricMatrix3D / inverse): R M R^T and R^T M R."""
    rng = np.random.default_rng(4363)
    for _ in range(ITERATIONS):
        R = next_rotation(rng)
        M = rng.uniform(-1, 1, (3, 3))
        M = 0.5 * (M + M.T)
        X = np.concatenate([R.reshape(-1), np.zeros(3)])
        J1, _, _ = rigid_apply(lib, X, np.ascontiguousarray(M.reshape(-1)), 0.3, np.zeros(3))
        assert np.abs(J1.reshape(3, 3) - R @ M @ R.T).max() <= EPS
        J2, _, _ = rigid_apply(lib, X, np.ascontiguousarray(M.reshape(-1)), 0.3, np.zeros(3), inverse=True)
        assert np.abs(J2.reshape(3, 3) - R.T @ M @ R).max() <= EPS


def test_dynamic_wrench_fast_path_equals_general_path(lib):
    """tools/MecanoToolsTest.java:292-460 and spatial/SpatialInertiaBasicsTest.java:129-157: with the centre of mass at the origin the
    general Newton-Euler expressions (tools/MecanoTools.java:632-702, 785-822) give what the fast ones do (:571-598, 728-752); zero
    motion gives a zero wrench; spinning about the CoM offset axis with the linear velocity along it gives a zero force / moment."""
    rng = np.random.default_rng(3453)
    z3 = np.zeros(3)
    for _ in range(ITERATIONS):
        J = np.ascontiguousarray(rng.uniform(-1, 1, 9))
        m = rng.uniform(0, 1)
        acc, tw = rng.uniform(-1, 1, 6), rng.uniform(-1, 1, 6)
        fast, gen = wrench(lib, J, m, z3, acc, tw), wrench(lib, J, m, z3, acc, tw, general=True)
        assert np.abs(fast - gen).max() <= EPS
        # the fast path is n = J wd + w x J w, f = m (a + w x v)
        Jm = J.reshape(3, 3)
        assert np.abs(fast[:3] - (Jm @ acc[:3] + np.cross(tw[:3], Jm @ tw[:3]))).max() <= EPS
        assert np.abs(fast[3:] - m * (acc[3:] + np.cross(tw[:3], tw[3:]))).max() <= EPS
        c = rng.uniform(-1, 1, 3)
        assert np.abs(wrench(lib, J, m, c, np.zeros(6), np.zeros(6))).max() == 0.0
        assert np.abs(wrench(lib, J, m, z3, np.zeros(6), np.zeros(6))).max() == 0.0
        w = rng.uniform(-10, 10) * c
        tw0 = np.concatenate([w, rng.uniform(-10, 10) * w])
        Ji = np.ascontiguousarray((rng.uniform(0, 1) * np.eye(3)).reshape(-1))
        assert np.abs(wrench(lib, Ji, m, c, np.zeros(6), tw0, general=True)).max() <= 1e-10
        assert np.abs(wrench(lib, Ji, m, z3, np.zeros(6), tw0)).max() <= 1e-10


def test_general_wrench_is_the_fast_wrench_of_the_same_body_seen_from_another_origin(lib):
    """The general branch (offset centre of mass) against the fast branch on the SAME physical body: describe the body about its
    centre of mass (c = 0, J_c), evaluate the fast wrench there, bring it to a frame whose origin is offset by -c; the general branch
    evaluated in that frame with (J = J_c - m c~ c~, c) must agree (MecanoTools.java:632-702, 785-822 are that derivation)."""
    rng = np.random.default_rng(9)
    for _ in range(ITERATIONS):
        a = rng.uniform(-1, 1, (3, 3))
        Jc = a @ a.T + 0.1 * np.eye(3)
        m = rng.uniform(0.1, 1)
        c = rng.uniform(-1, 1, 3)
        # frame F: origin at the body point o, CoM at c in F; frame G: same axes, origin at the CoM.  X maps G -> F coordinates.
        X = np.concatenate([np.eye(3).reshape(-1), c])
        accF, twF = rng.uniform(-1, 1, 6), rng.uniform(-1, 1, 6)
        accG, twG = motion(lib, X, accF, inverse=True), motion(lib, X, twF, inverse=True)
        wG = wrench(lib, np.ascontiguousarray(Jc.reshape(-1)), m, np.zeros(3), accG, twG)
        JF = Jc - m * tilde(c) @ tilde(c)
        wF = wrench(lib, np.ascontiguousarray(JF.reshape(-1)), m, c, accF, twF, general=True)
        assert np.abs(force(lib, X, wG) - wF).max() <= 1e-11
