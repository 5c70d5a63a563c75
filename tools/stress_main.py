"""Stress of the main entry points on the humanoid with its code object: random batch sizes (1 .. 70 000, biased to wave / work-group /
dispatch boundaries), every call against the oracle on a sample and against its sibling calls bit for bit.
python tools/stress_main.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from oracle.cpu_oracle import OracleModel

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(77)
sys_ = rt.nextHumanoid(np.random.default_rng(43))
d = rt.humanoid30Desc()
hm, om = HipModel(d), OracleModel(d)
g = (0.0, 0.0, -9.81)
edges = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16385, 32767, 32768, 32769, 65537]
t0, n, worst = time.time(), 0, 0.0
def chk(name, got, ref, B):
    global worst
    err = float(np.abs(got - ref).max())
    scale = max(1.0, float(np.abs(ref).max()))
    worst = max(worst, err / scale)
    assert err <= 1e-9 * scale, (name, B, err, scale)
while time.time() - t0 < budget:
    B = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 70000))
    base = min(B, 4096)
    st = rt.nextState(rng, sys_, base)
    q, qd, qdd, tau = (torch.tensor(x, device="cuda").repeat((B + base - 1) // base, 1)[:B].contiguous() for x in st)
    fx = torch.tensor(rng.uniform(-5, 5, (base, d.n_joints, 6)), device="cuda").repeat((B + base - 1) // base, 1, 1)[:B].contiguous() if rng.random() < 0.3 else None
    t1, a1 = hm.rnea(q, qd, qdd, g, f_ext=fx), hm.aba(q, qd, tau, g, f_ext=fx)
    t2, a2 = hm.rnea_aba(q, qd, qdd, tau, g, f_ext=fx)
    t3, H3 = hm.rnea_crba(q, qd, qdd, g, f_ext=fx)
    H1 = hm.crba(q)
    assert torch.equal(t1, t2) and torch.equal(t1, t3) and torch.equal(H1, H3), B
    # (forward dynamics: the bias split serves the single call up to 128 groups, the pair call up to 85 -- same q̈ to rounding in between)
    assert float((a2 - a1).abs().max()) <= 1e-10 * max(1.0, float(a1.abs().max())), B
    idx = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, 4)]))
    ti = torch.as_tensor(idx, device="cuda")
    sq, sqd, sqdd, stau = (x[ti].cpu().numpy() for x in (q, qd, qdd, tau))
    sf = fx[ti].cpu().numpy() if fx is not None else None
    chk("rnea", t1[ti].cpu().numpy(), om.rnea(sq, sqd, sqdd, g, sf), B)
    chk("aba", a1[ti].cpu().numpy(), om.aba(sq, sqd, stau, g, sf), B)
    chk("crba", H1[ti].cpu().numpy(), om.crba(sq), B)
    qn, vn, acc = hm.step(1e-3, q, qd, tau, g, f_ext=fx)
    # (the fused step runs the one-job tree-split forward dynamics, mh_aba_f64 the bias split at small batches: same q̈ to rounding)
    assert float((acc - a1).abs().max()) <= 1e-10 * max(1.0, float(a1.abs().max())), B
    n += 1
print(f"{n} random batches in {time.time() - t0:.0f} s, worst scaled error {worst:.2e}  [{hm.kernel_variant[:30]}]")
