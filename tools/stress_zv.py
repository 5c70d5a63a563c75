"""Stress of the bias-split forward dynamics (mh_zv_kernels.h): the four registered tree shapes with fresh physical parameters, random
batch sizes (biased to wave / work-group / dispatch boundaries), random 6-D root accelerations and external wrenches; the bias-split
launch (as dispatched, and forced at every batch size with MH_ZV=2) against the tree-split kernels on EVERY row (a race shows as whole
groups of 64 rows going wrong) and against the oracle on a sample.  python tools/stress_zv.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel
from oracle.cpu_oracle import OracleModel

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(2026)
edges = [1, 2, 63, 64, 65, 127, 128, 129, 511, 512, 513, 4095, 4096, 4097, 5440, 5441, 8191, 8192, 8193, 16384, 16385, 16448, 20000, 23999]
shapes = {"humanoid": rt.nextHumanoid, "torso": rt.nextFixedBaseTorso, "centaur": rt.nextCentaur, "quadruped": rt.nextQuadruped}
t0, n, worst, worst_pair = time.time(), 0, 0.0, 0.0
while time.time() - t0 < budget:
    name = list(shapes)[n % 4]
    sys_ = shapes[name](rng)
    d = sys_.toModelDesc()
    om = OracleModel(d)
    models = {}
    for mode in ("0", "1", "2"):
        os.environ["MH_ZV"] = mode
        models[mode] = HipModel(d)
        assert models[mode].kernel_variant.startswith("topo:"), (name, mode, models[mode].kernel_variant)
    for rep in range(6):
        B = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 24000))
        base = min(B, 2048)
        st = rt.nextState(rng, sys_, base)
        q, qd, qdd, tau = (torch.tensor(x, device="cuda").repeat((B + base - 1) // base, 1)[:B].contiguous() for x in st)
        fx = torch.tensor(rng.uniform(-5, 5, (base, d.n_joints, 6)), device="cuda").repeat((B + base - 1) // base, 1, 1)[:B].contiguous() if rng.random() < 0.4 else None
        g = rng.uniform(-3, 3, 6) if rng.random() < 0.5 else (0.0, 0.0, -9.81)
        a0 = models["0"].aba(q, qd, tau, g, f_ext=fx)
        t0_, a0p = models["0"].rnea_aba(q, qd, qdd, tau, g, f_ext=fx)
        scale = max(1.0, float(a0.abs().max()))
        for mode in ("1", "2"):
            for k in range(3):  # back to back: flags of the previous launch must never be taken for this one's
                a = models[mode].aba(q, qd, tau, g, f_ext=fx)
                t, ap = models[mode].rnea_aba(q, qd, qdd, tau, g, f_ext=fx)
            e1, e2 = float((a - a0).abs().max()) / scale, float((ap - a0p).abs().max()) / scale
            worst_pair = max(worst_pair, e1, e2)
            assert e1 <= 1e-9 and e2 <= 1e-9 and torch.equal(t, t0_), (name, mode, B, e1, e2)
        # the pair call's efforts against the inverse dynamics' own call (beyond one group per CU they are h + M qdd of the fused kernel)
        t_one = models["0"].rnea(q, qd, qdd, g, f_ext=fx)
        e3 = float((t0_ - t_one).abs().max()) / max(1.0, float(t_one.abs().max()))
        worst_pair = max(worst_pair, e3)
        assert e3 <= 1e-12, (name, B, e3)
        idx = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, 4)]))
        ti = torch.as_tensor(idx, device="cuda")
        sf = fx[ti].cpu().numpy() if fx is not None else None
        ref = om.aba(q[ti].cpu().numpy(), qd[ti].cpu().numpy(), tau[ti].cpu().numpy(), g, sf)
        err = float(np.abs(a[ti].cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max()))
        worst = max(worst, err)
        assert err <= 1e-9, (name, B, err)
    n += 1
    if n % 25 == 0:
        print(f"... {n} models, {time.time() - t0:.0f} s, worst so far {worst:.2e} / {worst_pair:.2e}", flush=True)
os.environ.pop("MH_ZV", None)
print(f"{n} models x 6 batches x 2 modes x 3 launches in {time.time() - t0:.0f} s: worst scaled error against the oracle {worst:.2e}, "
      f"worst bias-split vs tree-split over all rows {worst_pair:.2e}")
