"""Second, independent CPU restatement: textbook Featherstone (RBDA 2008, tables 5.1, 6.2, 7.1) with dense
6x6 Pluecker matrices in link (after-joint) frames.  TEST INFRASTRUCTURE ONLY.

It shares no code and no frame conventions with oracle/mecano_oracle.c (which follows Mecano's own frames
and step order); agreement of the two to ~1e-12 is one of the pins of the oracle (tests/test_oracle.py).
Pure numpy, one configuration at a time: use it on small cases only.
"""
from __future__ import annotations

import numpy as np

REVOLUTE, PRISMATIC, SIXDOF, FIXED = 0, 1, 2, 3


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def plucker_motion(R, p):
    """6x6 X such that a motion vector in the child frame maps to the parent frame, for a child frame posed (R, p) in its parent."""
    X = np.zeros((6, 6))
    X[:3, :3] = R
    X[3:, 3:] = R
    X[3:, :3] = skew(p) @ R
    return X


def rot_axis_angle(axis, angle):
    k = np.asarray(axis, dtype=float)
    k = k / np.linalg.norm(k)
    K = skew(k)
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def quat_to_R(q):
    x, y, z, s = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * s), 2 * (x * z + y * s)],
                     [2 * (x * y + z * s), 1 - 2 * (x * x + z * z), 2 * (y * z - x * s)],
                     [2 * (x * z - y * s), 2 * (y * z + x * s), 1 - 2 * (x * x + y * y)]])


def spatial_inertia(J_com, m, c):
    """6x6 inertia about a frame origin, for rotational inertia J_com about the CoM located at c."""
    C = skew(c)
    I = np.zeros((6, 6))
    I[:3, :3] = J_com + m * (C @ C.T)
    I[:3, 3:] = m * C
    I[3:, :3] = m * C.T
    I[3:, 3:] = m * np.eye(3)
    return I


def crm(v):
    out = np.zeros((6, 6))
    out[:3, :3] = skew(v[:3])
    out[3:, 3:] = skew(v[:3])
    out[3:, :3] = skew(v[3:])
    return out


def crf(v):
    return -crm(v).T


class Model:
    def __init__(self, desc):
        self.d = desc
        n = desc.n_joints
        self.n = n
        self.parent = np.asarray(desc.parent)
        self.type = np.asarray(desc.joint_type)
        self.axis = np.asarray(desc.axis).reshape(n, 3)
        self.Xb = np.asarray(desc.X_before).reshape(n, 12)
        Xc = np.asarray(desc.X_com).reshape(n, 12)
        J = np.asarray(desc.inertia_J).reshape(n, 3, 3)
        com = np.asarray(desc.inertia_com).reshape(n, 3)
        self.ndof = np.array([{REVOLUTE: 1, PRISMATIC: 1, SIXDOF: 6, FIXED: 0}[int(t)] for t in self.type])
        self.ncfg = np.array([{REVOLUTE: 1, PRISMATIC: 1, SIXDOF: 7, FIXED: 0}[int(t)] for t in self.type])
        self.dof_ofs = np.concatenate([[0], np.cumsum(self.ndof)])
        self.cfg_ofs = np.concatenate([[0], np.cumsum(self.ncfg)])
        self.dof_idx = np.asarray(desc.dof_indices)
        self.cfg_idx = np.asarray(desc.cfg_indices)
        self.nv, self.nq = desc.nv, desc.nq
        # body inertia expressed in the after-joint frame: rotate the body-fixed description, shift to the frame origin
        self.I = []
        for i in range(n):
            Rc, pc = Xc[i, :9].reshape(3, 3), Xc[i, 9:]
            # inertia about the body-fixed origin expressed in body axes -> about CoM
            c_b = com[i]
            J_about_com = J[i] - desc.inertia_mass[i] * (skew(c_b) @ skew(c_b).T)
            self.I.append(spatial_inertia(Rc @ J_about_com @ Rc.T, desc.inertia_mass[i], Rc @ c_b + pc))

    def S(self, i):
        t = self.type[i]
        if t == REVOLUTE:
            return np.concatenate([self.axis[i], np.zeros(3)]).reshape(6, 1)
        if t == PRISMATIC:
            return np.concatenate([np.zeros(3), self.axis[i]]).reshape(6, 1)
        if t == SIXDOF:
            return np.eye(6)
        return np.zeros((6, 0))

    def X_child_to_parent(self, i, q):
        """Motion transform from after-joint_i coordinates to parent after-joint coordinates."""
        ci = self.cfg_idx[self.cfg_ofs[i]:self.cfg_ofs[i + 1]]
        Rb, pb = self.Xb[i, :9].reshape(3, 3), self.Xb[i, 9:]
        t = self.type[i]
        if t == REVOLUTE:
            RJ, pJ = rot_axis_angle(self.axis[i], q[ci[0]]), np.zeros(3)
        elif t == PRISMATIC:
            RJ, pJ = np.eye(3), q[ci[0]] * self.axis[i]
        elif t == SIXDOF:
            RJ, pJ = quat_to_R(q[ci[:4]]), q[ci[4:7]]
        else:
            RJ, pJ = np.eye(3), np.zeros(3)
        return plucker_motion(Rb, pb) @ plucker_motion(RJ, pJ)

    def dofs(self, i):
        return self.dof_idx[self.dof_ofs[i]:self.dof_ofs[i + 1]]


def _root_acceleration(g):
    """(0, -g) for a gravity 3-vector; a 6-vector is the root's spatial acceleration itself (angular, linear)."""
    g = np.asarray(g, dtype=float).reshape(-1)
    return g.copy() if g.size == 6 else np.concatenate([np.zeros(3), -g])


def rnea(m: Model, q, qd, qdd, g, fext=None, return_wrenches=False):
    n = m.n
    v = [None] * n
    a = [None] * n
    f = [None] * n
    Xup = [None] * n  # parent -> child motion transform
    a0 = _root_acceleration(g)
    for i in range(n):
        Xcp = m.X_child_to_parent(i, q)
        Xup[i] = np.linalg.inv(Xcp)
        S = m.S(i)
        vJ = S @ qd[m.dofs(i)]
        p = m.parent[i]
        vp = np.zeros(6) if p < 0 else v[p]
        ap = a0 if p < 0 else a[p]
        v[i] = Xup[i] @ vp + vJ
        a[i] = Xup[i] @ ap + S @ qdd[m.dofs(i)] + crm(v[i]) @ vJ
        f[i] = m.I[i] @ a[i] + crf(v[i]) @ (m.I[i] @ v[i])
        if fext is not None:
            # fext is given in the body-fixed frame: bring it to the after-joint frame with the force transform of X_com
            Xc = np.asarray(m.d.X_com).reshape(n, 12)[i]
            Xf = np.linalg.inv(plucker_motion(Xc[:9].reshape(3, 3), Xc[9:])).T
            f[i] = f[i] - Xf @ fext[i]
    tau = np.zeros(m.nv)
    for i in range(n - 1, -1, -1):
        tau[m.dofs(i)] = m.S(i).T @ f[i]
        p = m.parent[i]
        if p >= 0:
            f[p] = f[p] + Xup[i].T @ f[i]
    if return_wrenches:
        return tau, np.array(f)  # wrench every joint transmits, after-joint frames
    return tau


def crba(m: Model, q):
    n = m.n
    Xup = [np.linalg.inv(m.X_child_to_parent(i, q)) for i in range(n)]
    Ic = [I.copy() for I in m.I]
    for i in range(n - 1, -1, -1):
        p = m.parent[i]
        if p >= 0:
            Ic[p] = Ic[p] + Xup[i].T @ Ic[i] @ Xup[i]
    H = np.zeros((m.nv, m.nv))
    for i in range(n):
        S = m.S(i)
        F = Ic[i] @ S
        di = m.dofs(i)
        H[np.ix_(di, di)] = S.T @ F
        j = i
        while m.parent[j] >= 0:
            F = Xup[j].T @ F
            j = m.parent[j]
            dj = m.dofs(j)
            H[np.ix_(dj, di)] = m.S(j).T @ F
            H[np.ix_(di, dj)] = (m.S(j).T @ F).T
    return H


def aba(m: Model, q, qd, tau, g, fext=None):
    n = m.n
    Xup, v, c, IA, pA = [None] * n, [None] * n, [None] * n, [None] * n, [None] * n
    for i in range(n):
        Xup[i] = np.linalg.inv(m.X_child_to_parent(i, q))
        S = m.S(i)
        vJ = S @ qd[m.dofs(i)]
        p = m.parent[i]
        v[i] = Xup[i] @ (np.zeros(6) if p < 0 else v[p]) + vJ
        c[i] = crm(v[i]) @ vJ
        IA[i] = m.I[i].copy()
        pA[i] = crf(v[i]) @ (m.I[i] @ v[i])
        if fext is not None:
            Xc = np.asarray(m.d.X_com).reshape(n, 12)[i]
            Xf = np.linalg.inv(plucker_motion(Xc[:9].reshape(3, 3), Xc[9:])).T
            pA[i] = pA[i] - Xf @ fext[i]
    U, Dinv, u = [None] * n, [None] * n, [None] * n
    for i in range(n - 1, -1, -1):
        S = m.S(i)
        U[i] = IA[i] @ S
        D = S.T @ U[i]
        Dinv[i] = np.linalg.inv(D) if D.size else D
        u[i] = tau[m.dofs(i)] - S.T @ pA[i]
        p = m.parent[i]
        if p >= 0:
            Ia = IA[i] - U[i] @ Dinv[i] @ U[i].T
            pa = pA[i] + Ia @ c[i] + U[i] @ Dinv[i] @ u[i]
            IA[p] = IA[p] + Xup[i].T @ Ia @ Xup[i]
            pA[p] = pA[p] + Xup[i].T @ pa
    a = [None] * n
    qdd = np.zeros(m.nv)
    a0 = _root_acceleration(g)
    for i in range(n):
        p = m.parent[i]
        ap = Xup[i] @ (a0 if p < 0 else a[p]) + c[i]
        qi = Dinv[i] @ (u[i] - U[i].T @ ap)
        qdd[m.dofs(i)] = qi
        a[i] = ap + m.S(i) @ qi
    return qdd


# ---------------------------------------------------------------------------------------------------------------------
# Coriolis matrix and centroidal momentum from dense body Jacobians (independent of the recursive composite / factorised
# inertia sweep of CompositeRigidBodyMassMatrixCalculator): with J_k the 6 x nv Jacobian of body k in its own after-joint
# coordinates and Jd_k its derivative (column j = kXj (v_j x S_j) for every ancestor-or-self joint j with a constant S_j),
#     C = sum_k J_k^T (I_k Jd_k + B_k J_k),     B_k = crf(v_k) I_k     (Echeandia & Wensing 2021, eq. 23 with B = v x* I)
#     A = sum_k cmX*_k I_k J_k,                  b = sum_k cmX*_k (I_k Jd_k qd + crf(v_k) I_k v_k)
def _body_jacobians(m: Model, q, qd):
    n = m.n
    Xup = [np.linalg.inv(m.X_child_to_parent(i, q)) for i in range(n)]  # parent -> child motion transforms
    v, J, Jd, X0 = [None] * n, [None] * n, [None] * n, [None] * n
    for i in range(n):
        p = m.parent[i]
        S = m.S(i)
        di = m.dofs(i)
        v[i] = Xup[i] @ (np.zeros(6) if p < 0 else v[p]) + S @ qd[di]
        J[i] = np.zeros((6, m.nv)) if p < 0 else Xup[i] @ J[p]
        Jd[i] = np.zeros((6, m.nv)) if p < 0 else Xup[i] @ Jd[p]
        J[i][:, di] = S
        Jd[i][:, di] = crm(v[i]) @ S
        X0[i] = Xup[i] @ (np.eye(6) if p < 0 else X0[p])  # root -> body i motion transform
    return v, J, Jd, X0


def coriolis_dense(m: Model, q, qd):
    v, J, Jd, _ = _body_jacobians(m, q, qd)
    C = np.zeros((m.nv, m.nv))
    for k in range(m.n):
        C += J[k].T @ (m.I[k] @ Jd[k] + crf(v[k]) @ m.I[k] @ J[k])
    return C


def centroidal_dense(m: Model, q, qd, frame=None, at_com=False):
    """(A [6, nv], b [6], origin of the centroidal frame in ``frame`` coordinates)."""
    v, J, Jd, X0 = _body_jacobians(m, q, qd)
    R, p = (np.eye(3), np.zeros(3)) if frame is None else (np.asarray(frame[:9], dtype=float).reshape(3, 3), np.asarray(frame[9:], dtype=float))
    origin = np.zeros(3)
    if at_com:
        mass, first = 0.0, np.zeros(3)
        for k in range(m.n):
            Xk0 = np.linalg.inv(X0[k])  # body k -> root: E = rotation of the body frame in the root, r = its position
            E, r = Xk0[:3, :3], None
            rx = Xk0[3:, :3] @ E.T
            r = np.array([rx[2, 1], rx[0, 2], rx[1, 0]])
            mk = m.I[k][3, 3]
            ck = np.array([m.I[k][2, 4], m.I[k][0, 5], m.I[k][1, 3]]) / mk  # from the m [c]x block
            first += mk * (E @ ck + r)
            mass += mk
        com_root = first / mass
        origin = R.T @ (com_root - p)
        p = com_root
    Xf = plucker_motion(R, p).T  # force transform root -> centroidal frame
    A, b = np.zeros((6, m.nv)), np.zeros(6)
    for k in range(m.n):
        to_root = X0[k].T  # force transform body k -> root = (motion transform root -> k)^T
        A += Xf @ to_root @ m.I[k] @ J[k]
        b += Xf @ to_root @ (m.I[k] @ (Jd[k] @ qd) + crf(v[k]) @ (m.I[k] @ v[k]))
    return A, b, origin


def relative_acceleration_dense(m: Model, q, qd, qdd, g, base, body):
    """Acceleration of body `body` relative to body `base` (indices of listed joints, -1 = the root body), expressed in the body-fixed
    frame of `body`, in the reference's convention (component-wise derivative of the relative twist expressed in the moving body frame),
    from the dense body Jacobians: with Featherstone's spatial accelerations a_i = J_i qdd + Jd_i qd + X0_i a0 (true derivatives, which
    subtract like vectors),  A_rel = a_2 - X_{1->2} a_1 + v_2 x (X_{1->2} v_1).  Independent of the cross-product recipe the C oracle
    restates from SpatialAccelerationBasics.changeFrame."""
    v, J, Jd, X0 = _body_jacobians(m, q, qd)
    a0 = _root_acceleration(g)
    n = m.n
    Xc = np.asarray(m.d.X_com).reshape(n, 12)

    def quantities(i):
        if i < 0:
            return np.eye(6), np.zeros(6), a0
        return X0[i], v[i], J[i] @ qdd + Jd[i] @ qd + X0[i] @ a0

    X1, v1, a1 = quantities(base)
    X2, v2, a2 = quantities(body)
    X12 = X2 @ np.linalg.inv(X1)  # frame of base -> frame of body (after-joint frames)
    rel = a2 - X12 @ a1 + crm(v2) @ (X12 @ v1)
    if body < 0:
        return rel
    return np.linalg.inv(plucker_motion(Xc[body, :9].reshape(3, 3), Xc[body, 9:])) @ rel  # after-joint -> body-fixed frame
