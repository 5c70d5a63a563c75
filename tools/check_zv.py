"""Quick parity check of the bias-split forward dynamics (mh_aba_f64 and mh_rnea_aba_f64, MH_ZV=2: at every batch size) against the oracle,
humanoid and the other registered tree shapes; and repeated launches + a graph replay (the flags must come back to zero)."""
import os, sys
os.environ.setdefault("MH_ZV", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mecano_amd import build as b, random_tools as rt
from mecano_amd.engine import HipModel
from oracle.cpu_oracle import OracleModel
g = (0.3, -0.2, -9.81)
worst = 0.0
models = b.registered_models() if "all" in sys.argv else {"humanoid30": b.registered_models()["humanoid30"]}
for name, d in models.items():
    hm, om = HipModel(d), OracleModel(d)
    print(name, hm.kernel_variant[:40], flush=True)
    for B in (1, 63, 64, 65, 100, 4096, 5000, 12000):
        rng = np.random.default_rng(B)
        q = rng.uniform(-3, 3, (B, d.nq)); qd, qdd, tau = (rng.uniform(-1, 1, (B, d.nv)) for _ in range(3))
        for j, t in enumerate(d.joint_type):
            if t == 2:
                c = int(d.cfg_indices[sum((7 if tt == 2 else (0 if tt == 3 else 1)) for tt in d.joint_type[:j])]) if False else None
        # unit quaternions for 6-DoF joints (the first four configuration entries of each)
        ofs = 0
        for t in d.joint_type:
            if t == 2:
                q[:, ofs:ofs + 4] /= np.linalg.norm(q[:, ofs:ofs + 4], axis=1, keepdims=True)
            ofs += 7 if t == 2 else (0 if t == 3 else 1)
        dev = lambda x: torch.tensor(x, device="cuda")
        tq, tqd, tqdd, ttau = dev(q), dev(qd), dev(qdd), dev(tau)
        idx = np.unique(np.concatenate([np.arange(0, B, max(1, B // 200)), [B - 1]]))
        a_ref, t_ref = om.aba(q[idx], qd[idx], tau[idx], g), om.rnea(q[idx], qd[idx], qdd[idx], g)
        for rep in range(3):
            a = hm.aba(tq, tqd, ttau, g).cpu().numpy()[idx]
            t2, a2 = hm.rnea_aba(tq, tqd, tqdd, ttau, g)
            e = max(np.abs(a - a_ref).max() / max(1, np.abs(a_ref).max()), np.abs(a2.cpu().numpy()[idx] - a_ref).max() / max(1, np.abs(a_ref).max()),
                    np.abs(t2.cpu().numpy()[idx] - t_ref).max() / max(1, np.abs(t_ref).max()))
            worst = max(worst, e)
            assert e < 1e-8 and np.isfinite(e), (name, B, rep, e)
        hm.check()
    # a captured pair launch replayed: same epoch every time
    B = 4096
    rng = np.random.default_rng(7)
    st = [torch.tensor(x, device="cuda") for x in (rng.uniform(-1, 1, (B, d.nq)), rng.uniform(-1, 1, (B, d.nv)), rng.uniform(-1, 1, (B, d.nv)), rng.uniform(-1, 1, (B, d.nv)))]
    o1, o2 = torch.empty_like(st[1]), torch.empty_like(st[1])
    fn = hm.bind_rnea_aba(st[0], st[1], st[2], st[3], o1, o2, g)
    fn(); torch.cuda.synchronize()
    r1, r2 = o1.clone(), o2.clone()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        hm.bind_rnea_aba(st[0], st[1], st[2], st[3], o1, o2, g)()  # (bound inside: the binding takes the current -- capturing -- stream)
    for _ in range(5):
        o1.zero_(), o2.zero_()
        gr.replay(); torch.cuda.synchronize()
        if not (torch.equal(o1, r1) and torch.equal(o2, r2)):
            d1, d2 = (o1 - r1).abs(), (o2 - r2).abs()
            print("replay differs:", name, "tau rows", int((d1.amax(1) > 0).sum()), "max", float(d1.max()), "| qdd rows", int((d2.amax(1) > 0).sum()), "max", float(d2.max()),
                  "nan", int(torch.isnan(o2).sum()), "first bad rows", torch.nonzero(d2.amax(1) > 0)[:8].flatten().tolist(), flush=True)
            raise SystemExit(1)
    hm.check()
print("worst scaled error %.3e" % worst)
