"""Host-side mirror of Mecano's multi-body API: iteration order, index maps, model flattening."""
import numpy as np
import pytest

from mecano_amd import random_tools as rt
from mecano_amd.multibody import (FixedJoint, JointMatrixIndexProvider, MultiBodySystem, PrismaticJoint, RevoluteJoint, RigidBody,
                                  SixDoFJoint)


def build_small_tree():
    root = RigidBody("elevator")
    a = RevoluteJoint("a", root, None, (0, 0, 1))
    A = RigidBody("A", a, np.eye(3), 1.0, centerOfMassOffset=(0, 0, 0.1))
    b = PrismaticJoint("b", A, (np.eye(3), (0.1, 0, 0)), (1, 0, 0))
    B = RigidBody("B", b, np.eye(3), 2.0)
    c = SixDoFJoint("c", A)
    C = RigidBody("C", c, np.eye(3), 3.0)
    d = RevoluteJoint("d", B, None, (0, 1, 0))
    RigidBody("D", d, np.eye(3), 4.0)
    e = FixedJoint("e", C)
    RigidBody("E", e, np.eye(3), 5.0)
    return root, (a, b, c, d, e)


def test_depth_first_preorder_children_in_creation_order():
    """iterators/JointIterator.java:130-177: a, b, d (subtree of b first), then c, e."""
    root, (a, b, c, d, e) = build_small_tree()
    assert [j.name for j in root.subtreeJointList()] == ["a", "b", "d", "c", "e"]


def test_index_provider_running_sums():
    """JointMatrixIndexProvider.java:71-123."""
    root, (a, b, c, d, e) = build_small_tree()
    sys_ = MultiBodySystem.toMultiBodySystemInput(root)
    p = sys_.getJointMatrixIndexProvider()
    assert p.getJointDoFIndices(a) == [0] and p.getJointDoFIndices(b) == [1] and p.getJointDoFIndices(d) == [2]
    assert p.getJointDoFIndices(c) == [3, 4, 5, 6, 7, 8] and p.getJointDoFIndices(e) == []
    assert p.getJointConfigurationIndices(c) == [3, 4, 5, 6, 7, 8, 9]
    assert sys_.getNumberOfDoFs() == 9 and sys_.getConfigurationSize() == 10


def test_model_desc_flattening():
    root, (a, b, c, d, e) = build_small_tree()
    desc = MultiBodySystem.toMultiBodySystemInput(root).toModelDesc()
    assert desc.n_joints == 5 and desc.nv == 9 and desc.nq == 10
    assert list(desc.parent) == [-1, 0, 1, 0, 3]
    assert list(desc.joint_type) == [0, 1, 0, 2, 3]
    assert np.allclose(desc.X_before.reshape(5, 12)[1, 9:], (0.1, 0, 0))
    assert np.allclose(desc.X_com.reshape(5, 12)[0, 9:], (0, 0, 0.1))
    assert np.allclose(desc.inertia_mass, [1, 2, 4, 3, 5])


def test_joints_to_ignore_take_their_subtree():
    """MultiBodySystemReadOnly.java:167-171."""
    root, (a, b, c, d, e) = build_small_tree()
    sys_ = MultiBodySystem.toMultiBodySystemInput(root, [b])
    assert [j.name for j in sys_.getJointsToConsider()] == ["a", "c", "e"]
    assert [j.name for j in sys_.getJointsToIgnore()] == ["b", "d"]
    desc = sys_.toModelDesc()
    assert desc.n_joints == 3 and desc.nv == 7


def test_humanoid_shape():
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    d = sys_.toModelDesc()
    assert (d.n_joints, d.nv, d.nq) == (25, 30, 31)
    depth = np.zeros(25, dtype=int)
    for i in range(25):
        depth[i] = 1 if d.parent[i] < 0 else depth[d.parent[i]] + 1
    assert depth.max() == 8  # pelvis -> hand


def test_topology_key_depends_on_structure_only():
    rng = np.random.default_rng(1)
    a = rt.nextHumanoid(rng).toModelDesc()
    b = rt.nextHumanoid(rng).toModelDesc()
    assert a.topology_key() == b.topology_key()
    c = MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(rng, 7)[0].getPredecessor()).toModelDesc()
    assert c.topology_key() != a.topology_key()


def test_ignored_subtree_lumping_equals_welding():
    """considerIgnoredSubtreesInertia (InverseDynamicsCalculator.java:832-860): ignoring a joint and lumping its subtree into the
    parent body is the same mechanical system as welding that subtree (fixed joints) at the zero configuration."""
    from oracle.cpu_oracle import OracleModel
    rng = np.random.default_rng(12)

    from helpers import build_lump_pair as build

    root_w, _ = build(True)
    welded = MultiBodySystem.toMultiBodySystemInput(root_w)
    root_i, k0 = build(False)
    ignoring = MultiBodySystem.toMultiBodySystemInput(root_i, [k0])
    d_w = welded.toModelDesc()
    d_i = ignoring.toModelDesc(considerIgnoredSubtreesInertia=True)
    assert d_i.n_joints == 3 and d_w.n_joints == 5 and d_i.nv == d_w.nv == 3
    assert np.abs(d_i.inertia_com.reshape(-1, 3)[0]).max() > 1e-3  # the lump moves the CoM off the body-fixed origin
    q, qd, qdd, tau = rt.nextState(rng, ignoring, 4)
    om_w, om_i = OracleModel(d_w), OracleModel(d_i)
    g = (0.1, 0.2, -9.81)
    assert np.allclose(om_i.rnea(q, qd, qdd, g), om_w.rnea(q, qd, qdd, g), rtol=0, atol=1e-11)
    assert np.allclose(om_i.aba(q, qd, tau, g), om_w.aba(q, qd, tau, g), rtol=0, atol=1e-10)
    assert np.allclose(om_i.crba(q), om_w.crba(q), rtol=0, atol=1e-11)
    # without lumping the ignored subtree's inertia is simply gone: a different system
    d_n = ignoring.toModelDesc(considerIgnoredSubtreesInertia=False)
    assert np.abs(OracleModel(d_n).rnea(q, qd, qdd, g) - om_w.rnea(q, qd, qdd, g)).max() > 1e-3


def test_tree_split_plans_of_the_registered_code_objects():
    """The compile-time partition each specialised code object was built with (mh_spec_split_plan, host-only): the humanoid and the
    fixed-base torso get the staged trunk (a sub-trunk folded between two barriers), the quadruped the plain split (no sub-trunk), the
    7-joint arm none (a chain).  Checks the invariants the kernels rely on: one late limb per wave at most, a free wave for the sub-trunk,
    every limb owned, the cut inside its wave's late limb."""
    import ctypes
    from mecano_amd import build as b
    try:
        b.build_all(jobs=5)  # no-op when the driver's build() has run; otherwise the objects are compiled side by side (minutes)
        paths = {name: b.build_spec(desc) for name, desc in b.registered_models().items()}
    except Exception as e:  # no hipcc and no prebuilt objects
        pytest.skip(f"specialised code objects unavailable: {e}")
    expect = {"humanoid30": (1, 1), "arm7": (0, 0), "quadruped18": (1, 0), "torso13": (1, 1), "centaur20": (1, 1)}
    for name, path in paths.items():
        lib = ctypes.CDLL(path)
        buf = (ctypes.c_int * 256)()
        n = lib.mh_spec_split_plan(buf, 256)
        v = list(buf[:n])
        usable, staged, n_limbs, n_sub, root = v[:5]
        assert (usable, staged) == expect[name], (name, v)
        limbs = [v[5 + 5 * k: 10 + 5 * k] for k in range(n_limbs)]
        cuts = v[5 + 5 * n_limbs: 9 + 5 * n_limbs]
        if not usable:
            continue
        assert all(0 <= l[2] < 4 and 0 <= l[3] < 4 for l in limbs)
        if staged:
            late_owners = [l[2] for l in limbs if l[4]]
            assert len(late_owners) == len(set(late_owners)) <= 3 and n_sub >= 1 and root >= 0
            for w, c in enumerate(cuts):
                if c >= 0:  # the cut body belongs to a late limb of wave w
                    assert any(l[4] and l[2] == w and l[0] <= c < l[0] + l[1] for l in limbs), (name, w, c, limbs)
        else:
            assert all(l[2] == l[3] for l in limbs)
        print(name, "staged" if staged else "plain", limbs, cuts)


def test_fused_kernel_capabilities_of_the_registered_code_objects():
    """What the host asks a code object before it sends device-filling batches to the fused forward-dynamics kernel (host-only exports):
    mh_spec_zvf_usable (joints below the root revolute / fixed, two workgroups per CU fit) and -- round 5 -- mh_spec_zvf_pair_usable (the same
    launch writes tau = h + M(q) qdd for mh_rnea_aba_f64).  The humanoid has both; a chain has neither; whoever serves the pair serves the
    forward dynamics; the LDS of a workgroup never exceeds half a CU's."""
    import ctypes
    from mecano_amd import build as b
    try:
        b.build_all(jobs=5)
        paths = {name: b.build_spec(desc) for name, desc in b.registered_models().items()}
    except Exception as e:  # no hipcc and no prebuilt objects
        pytest.skip(f"specialised code objects unavailable: {e}")
    seen = {}
    for name, path in paths.items():
        lib = ctypes.CDLL(path)
        lib.mh_spec_zvf_lds_bytes.restype = ctypes.c_long
        zvf, pair, lds = lib.mh_spec_zvf_usable(), lib.mh_spec_zvf_pair_usable(), lib.mh_spec_zvf_lds_bytes()
        seen[name] = (zvf, pair, lds)
        assert pair <= zvf, (name, zvf, pair)
        if zvf:
            assert 0 < lds * 2 <= 160 * 1024, (name, lds)
    assert seen["humanoid30"][:2] == (1, 1) and seen["arm7"][:2] == (0, 0), seen
    assert seen["humanoid30"][2] == 160 * 64 * 8  # 157 slots of both phases + the mailed head's three (tests/test_split_plan.py)


def test_committed_humanoid_model_is_what_the_generator_produces():
    """mecano_amd/models/humanoid30.json (the benchmark model) against nextHumanoid(default_rng(43)): identical."""
    from mecano_amd import random_tools as rt
    a, b = rt.humanoid30Desc(), rt.nextHumanoid(np.random.default_rng(43)).toModelDesc()
    assert (a.n_joints, a.nq, a.nv) == (b.n_joints, b.nq, b.nv) == (25, 31, 30)
    for f in ("parent", "joint_type", "axis", "X_before", "X_com", "inertia_J", "inertia_mass", "inertia_com", "dof_indices", "cfg_indices"):
        assert np.array_equal(np.asarray(getattr(a, f)).reshape(-1), np.asarray(getattr(b, f)).reshape(-1)), f


def test_joint_state_and_extract_insert_follow_the_reference():
    """JointBasics.setJointConfiguration / ...Velocity / ...Acceleration / ...Tau (multiBodySystem/interfaces/JointBasics.java:150-224) and
    MultiBodySystemTools.extractJointsState / insertJointsState (tools/MultiBodySystemTools.java:1433-1491, 1578-1637): joints in index
    provider order, each consuming getConfigurationMatrixSize() / getDegreesOfFreedom() rows; a fresh floating joint sits at the identity."""
    import numpy as np
    from mecano_amd import random_tools as rt
    from mecano_amd.multibody import JointStateType, MultiBodySystemTools
    sys_ = rt.nextHumanoid(np.random.default_rng(1))
    joints = sys_.getJointMatrixIndexProvider().getIndexedJointsInOrder()
    nq, nv = sys_.getConfigurationSize(), sys_.getNumberOfDoFs()
    q0 = np.zeros((nq, 1))
    assert MultiBodySystemTools.extractJointsState(joints, JointStateType.CONFIGURATION, q0) == nq
    assert q0[3, 0] == 1.0 and np.count_nonzero(q0) == 1  # identity quaternion (x, y, z, s) of the pelvis, everything else zero
    q, qd, qdd, tau = rt.nextState(np.random.default_rng(2), sys_, 1)
    for kind, row, size in ((JointStateType.CONFIGURATION, q[0], nq), (JointStateType.VELOCITY, qd[0], nv),
                            (JointStateType.ACCELERATION, qdd[0], nv), (JointStateType.EFFORT, tau[0], nv)):
        assert MultiBodySystemTools.insertJointsState(joints, kind, row.reshape(-1, 1)) == size
        back = np.zeros((size, 1))
        assert MultiBodySystemTools.extractJointsState(joints, kind, back) == size
        assert np.array_equal(back[:, 0], row)
    # per-joint access with a row offset, and the one-DoF convenience setters
    provider = sys_.getJointMatrixIndexProvider()
    knee = joints[4]
    assert knee.getDegreesOfFreedom() == 1 and knee.getQd() == qd[0, provider.getJointDoFIndices(knee)[0]]
    knee.setQdd(0.25)
    m = np.zeros(5)
    assert knee.getJointAcceleration(3, m) == 4 and m[3] == 0.25
