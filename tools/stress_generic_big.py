"""Stress of the run-time-topology RNEA / ABA at device-filling batches (depth-first kernels, sweep ABA, bushy-tree routing, transposed copies,
row windows): random trees of every joint kind, fp64 and fp32, AoS and SoA, against the oracle on a sample.
MH_DISABLE_SPEC=1 python tools/stress_generic_big.py [seconds]"""
import os, sys, time
os.environ.setdefault("MH_DISABLE_SPEC", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel
from mecano_amd.multibody import MultiBodySystem
from oracle.cpu_oracle import OracleModel

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(555)
kinds = ("revolute", "prismatic", "sixdof", "fixed", "planar", "spherical")
g = (0.0, 0.0, -9.81)
t0, n, worst64, worst32 = time.time(), 0, 0.0, 0.0
while time.time() - t0 < budget:
    nj = int(rng.integers(2, 45))
    joints = rt.nextJointTree(rng, nj, kinds) if rng.random() < 0.6 else (rt.nextJointChain(rng, nj, ("revolute", "prismatic")) if rng.random() < 0.5 else rt.nextFloatingChain(rng, nj, ("revolute", "prismatic"), tree=True))
    sys_ = MultiBodySystem.toMultiBodySystemInput(joints[0].getPredecessor())
    d = sys_.toModelDesc()
    if d.nv == 0:
        continue
    hm, om = HipModel(d), OracleModel(d)
    B = int(rng.choice([33000, 40001, 70000]))
    base = 2048
    st = rt.nextState(rng, sys_, base)
    idx = np.unique(rng.integers(0, B, 5))
    sq, sqd, sqdd, stau = (x[idx % base] for x in st)
    ref_t, ref_a = om.rnea(sq, sqd, sqdd, g), om.aba(sq, sqd, stau, g)
    # the oracle's own backward error on this sample: what the conditioning of the system allows
    eo = float(np.abs(om.rnea(sq, sqd, ref_a, g) - stau).max()) / max(1.0, float(np.abs(stau).max()), float(np.abs(ref_t).max()))
    for dt in (torch.float64, torch.float32):
        q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=dt).repeat((B + base - 1) // base, 1)[:B].contiguous() for x in st)
        ti = torch.as_tensor(idx, device="cuda")
        for layout in (_lib.LAYOUT_AOS, _lib.LAYOUT_SOA):
            if layout == _lib.LAYOUT_SOA:
                a = [x.t().contiguous() for x in (q, qd, qdd, tau)]
                tt, aa = hm.rnea(a[0], a[1], a[2], g, layout=layout).t(), hm.aba(a[0], a[1], a[3], g, layout=layout).t()
            else:
                tt, aa = hm.rnea(q, qd, qdd, g), hm.aba(q, qd, tau, g)
            et = float(np.abs(tt[ti].double().cpu().numpy() - ref_t).max()) / max(1.0, float(np.abs(ref_t).max()))
            # forward dynamics by its backward error (random mixed trees can be ill-conditioned): inverse dynamics of the result = the efforts
            back = om.rnea(sq, sqd, aa[ti].double().cpu().numpy(), g)
            ea = float(np.abs(back - stau).max()) / max(1.0, float(np.abs(stau).max()), float(np.abs(ref_t).max()))
            if dt == torch.float64:
                worst64 = max(worst64, et, ea)
                assert et <= 1e-9 and ea <= max(1e-9, 50 * eo), (nj, B, layout, et, ea, eo)
            else:
                worst32 = max(worst32, et)
                # fp32 forward dynamics is conditioning-bound (tests/test_gpu_parity.py): the oracle's fp64 backward error scaled by u32 / u64
                assert np.isfinite(ea) and ea <= max(256 * nj * 2.0 ** -24, 4e9 * eo) and et <= 64 * nj * 2.0 ** -24, (nj, B, layout, et, ea, eo)
    n += 1
print(f"{n} random systems in {time.time() - t0:.0f} s; worst scaled error fp64 {worst64:.2e}, fp32 RNEA {worst32:.2e}")
