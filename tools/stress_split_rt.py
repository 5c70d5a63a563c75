"""One-off stress of the run-time tree split planner and kernels: many random trees / forests of every joint kind, split forced on against off."""
import os, sys
os.environ["MH_DISABLE_SPEC"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem, RigidBody
kinds_all = ("revolute", "prismatic", "sixdof", "fixed", "planar", "spherical")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = (0.1, -0.3, -9.81)
n_split = worst_t = worst_a = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    root = RigidBody("root")
    for k in range(int(rng.integers(1, 5))):  # a forest of 1..4 trees
        n = int(rng.integers(1, 70))
        kinds = kinds_all if rng.random() < 0.5 else ("revolute", "prismatic")
        (rt.nextJointTree if rng.random() < 0.8 else rt.nextJointChain)(rng, n, kinds, rootBody=root, prefix=f"t{k}_")
    sys_ = MultiBodySystem.toMultiBodySystemInput(root)
    d = sys_.toModelDesc()
    os.environ["MH_SPLIT_RT"] = "1"; on = HipModel(d)
    os.environ["MH_SPLIT_RT"] = "0"; off = HipModel(d)
    used = "run-time tree split" in on.kernel_variant
    n_split += used
    B = int(rng.integers(1, 400))
    q, qd, qdd, tau = (torch.tensor(np.ascontiguousarray(x), device="cuda") for x in rt.nextState(rng, sys_, B))
    fext = torch.tensor(rng.uniform(-1, 1, (B, d.n_joints, 6)), device="cuda")
    t, t0 = on.rnea(q, qd, qdd, g, fext), off.rnea(q, qd, qdd, g, fext)
    a, a0 = on.aba(q, qd, tau, g, fext), off.aba(q, qd, tau, g, fext)
    tp, ap = on.rnea_aba(q, qd, qdd, tau, g, fext)  # one launch while 2 * groups <= CUs (pair_split_kernel): bit for bit the two single calls
    assert torch.equal(tp, t) and torch.equal(ap, a), (it, d.n_joints, B, float((tp - t).abs().max()), float((ap - a).abs().max()))
    et = float((t - t0).abs().max() / max(1.0, float(t0.abs().max()))); ea = float((a - a0).abs().max() / max(1.0, float(a0.abs().max())))
    back = float((on.rnea(q, qd, a, g, fext) - tau).abs().max() / max(1.0, float(tau.abs().max())))  # ABA then RNEA returns the efforts
    worst_t, worst_a = max(worst_t, et), max(worst_a, back)
    flag = "" if (et < 1e-11 and back < 1e-7) else "   <-- CHECK"
    print(f"{it:3d} n={d.n_joints:3d} B={B:3d} split={'yes' if used else 'no '} rnea {et:.1e} aba {ea:.1e} round trip {back:.1e}{flag}  {on.kernel_variant[40:]}", flush=True)
print("models with a split:", n_split, "worst rnea diff", worst_t, "worst ABA round trip", worst_a)
