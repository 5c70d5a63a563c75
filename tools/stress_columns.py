"""Stress of the kernels that spread bodies' columns over several waves (mass matrix, Coriolis matrix, centroidal momentum, regressor):
random trees of every joint kind, random batch sizes around the wave / part boundaries, against the oracle.  python tools/stress_columns.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem
from oracle.cpu_oracle import OracleModel

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(20261004)
t0, n, worst = time.time(), 0, 0.0
kinds = ("revolute", "prismatic", "sixdof", "fixed", "planar", "spherical")
while time.time() - t0 < budget:
    nj = int(rng.integers(1, 40))
    joints = rt.nextJointTree(rng, nj, kinds) if rng.random() < 0.7 else rt.nextFloatingChain(rng, nj, ("revolute", "prismatic"), tree=bool(rng.integers(0, 2)))
    sys_ = MultiBodySystem.toMultiBodySystemInput(joints[0].getPredecessor())
    d = sys_.toModelDesc()
    if d.nv == 0:
        continue
    hm, om = HipModel(d), OracleModel(d)
    B = int(rng.choice([1, 63, 64, 65, 127, 129, 500, 1000, 2047, 4097, int(rng.integers(1, 6000))]))
    q, qd, qdd, _ = rt.nextState(rng, sys_, B)
    tq, tqd, tqdd = (torch.tensor(x, device="cuda") for x in (q, qd, qdd))
    idx = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, 6)]))
    ti = torch.as_tensor(idx, device="cuda")
    def chk(name, got, ref):
        global worst
        err = float(np.abs(got - ref).max()) if ref.size else 0.0
        scale = max(1.0, float(np.abs(ref).max()) if ref.size else 0.0)
        worst = max(worst, err / scale)
        assert err <= 1e-9 * scale, (name, nj, B, err, scale)
    H, C = hm.crba_coriolis(tq, tqd)
    Ho, Co = om.crba_coriolis(q[idx], qd[idx])
    chk("H", H[ti].cpu().numpy(), Ho), chk("C", C[ti].cpu().numpy(), Co)
    chk("crba", hm.crba(tq)[ti].cpu().numpy(), Ho)
    A, b, com = hm.centroidal(tq, tqd, at_com=True)
    Ao, bo, como = om.centroidal(q[idx], qd[idx], at_com=True)
    chk("A", A[ti].cpu().numpy(), Ao), chk("b", b[ti].cpu().numpy(), bo), chk("com", com[ti].cpu().numpy(), como)
    s = idx[:3]
    Y = hm.regressor(tq, tqd, tqdd)
    chk("Y", Y[torch.as_tensor(s, device="cuda")].cpu().numpy(), om.regressor(q[s], qd[s], qdd[s]))
    n += 1
print(f"{n} random systems in {time.time() - t0:.0f} s, worst scaled error {worst:.2e}  [{hm.kernel_variant[:30]}]")
