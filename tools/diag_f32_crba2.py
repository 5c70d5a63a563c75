import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt, _lib
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem, RigidBody, SixDoFJoint, SphericalJoint, PlanarJoint
from oracle.cpu_oracle import OracleModel
np.set_printoptions(precision=4, linewidth=200, suppress=True)
dev = lambda x, dt=torch.float64: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dt)
rng = np.random.default_rng(1)
for name, cls in (("sixdof", SixDoFJoint), ("spherical", SphericalJoint), ("planar", PlanarJoint)):
    root = RigidBody("root")
    j = cls("j", root)
    rt.nextRigidBody(rng, "b", j)
    sys_ = MultiBodySystem.toMultiBodySystemInput(root)
    d = sys_.toModelDesc()
    hm, om = HipModel(d), OracleModel(d)
    q, _, _, _ = rt.nextState(rng, sys_, 2)
    Hr = om.crba(q)
    H32 = hm.crba(dev(q, torch.float32)).cpu().numpy()
    H32s = hm.crba(dev(q, torch.float32).t().contiguous(), layout=_lib.LAYOUT_SOA).cpu().numpy().T.reshape(Hr.shape)
    print(name, "err aos", np.abs(H32 - Hr).max(), "err soa", np.abs(H32s - Hr).max())
    print(H32[0]); print(Hr[0])
