"""Diagnostic: where does mh_crba_f32 differ from the fp64 oracle on the all-joint-kinds family (tests/test_gpu_known_answers.py)?"""
import os, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem
from oracle.cpu_oracle import OracleModel

kinds = ("revolute", "prismatic", "sixdof", "fixed", "planar", "spherical")
rng = np.random.default_rng(zlib.crc32(("f32" + "all_kinds_tree").encode()))
dev = lambda x, dt=torch.float64: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dt)
for it in range(4):
    n = int(rng.integers(1, 41))
    joints = rt.nextJointTree(rng, n, kinds)
    sys_ = MultiBodySystem.toMultiBodySystemInput(joints[0].getPredecessor())
    d = sys_.toModelDesc()
    hm, om = HipModel(d), OracleModel(d)
    B = int(rng.integers(1, 150))
    q, qd, qdd, tau = rt.nextState(rng, sys_, B)
    if it % 2:
        rng.uniform(-1, 1, (B, d.n_joints, 6))
    H_ref = om.crba(q)
    H64 = hm.crba(dev(q)).cpu().numpy()
    H32 = hm.crba(dev(q, torch.float32)).cpu().numpy().astype(np.float64)
    e64, e32 = np.abs(H64 - H_ref), np.abs(H32 - H_ref)
    print(f"it {it}: n {d.n_joints} nv {d.nv} B {B} max|H| {np.abs(H_ref).max():.3e}  err64 {e64.max():.3e}  err32 {e32.max():.3e}")
    # which DoF belongs to which joint / kind
    owner, kind_of = np.zeros(d.nv, int), np.zeros(d.nv, int)
    ofs = 0
    ndof = {0: 1, 1: 1, 2: 6, 3: 0, 4: 3, 5: 3}
    for j in range(d.n_joints):
        for k in range(ndof[int(d.joint_type[j])]):
            owner[d.dof_indices[ofs]] = j
            kind_of[d.dof_indices[ofs]] = int(d.joint_type[j])
            ofs += 1
    bad = np.argwhere(e32 > 1e-3 * np.abs(H_ref).max())
    print("   bad entries:", len(bad), " bad configurations:", len(set(bad[:, 0].tolist())), "of", B)
    seen = set()
    for b, r, c in bad[:4000]:
        key = (kind_of[r], kind_of[c])
        if key not in seen:
            seen.add(key)
            print(f"   cfg {b} H[{r},{c}] joints ({owner[r]} kind {kind_of[r]}, {owner[c]} kind {kind_of[c]}): f32 {H32[b, r, c]:.6e} ref {H_ref[b, r, c]:.6e}")
    if len(bad):
        bcfg = bad[0][0]
        qq = q[bcfg]
        print("   q of first bad cfg: max |q|", np.abs(qq).max())
        # same configuration alone
        H1 = hm.crba(dev(q[bcfg:bcfg + 1], torch.float32)).cpu().numpy().astype(np.float64)
        print("   alone: err", np.abs(H1[0] - H_ref[bcfg]).max())
        break
