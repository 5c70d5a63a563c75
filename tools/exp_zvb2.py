"""Timing of mh_aba_f64 on the humanoid at the batch sizes given, under whatever MH_* environment the caller set (one line per size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mecano_amd import random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from oracle.cpu_oracle import OracleModel
sys_ = rt.nextHumanoid(np.random.default_rng(43))
desc = sys_.toModelDesc()
hm, om = HipModel(desc), OracleModel(desc)
stream = torch.cuda.current_stream().cuda_stream
g = (0.3, -0.2, -9.81)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MH_"))
for B in [int(a) for a in sys.argv[1:]] or [32768, 262144]:
    q, qd, qdd, tau = rt.nextState(np.random.default_rng(B), sys_, min(B, 32768 + 77))
    rep = (B + q.shape[0] - 1) // q.shape[0]
    dq, dqd, dtau = (torch.tensor(x, device="cuda").repeat(rep, 1)[:B].contiguous() for x in (q, qd, tau))
    n = min(B, 200)
    a = hm.aba(dq, dqd, dtau, g)
    torch.cuda.synchronize()
    ah = a.cpu().numpy()
    tq, tqd, ttau = (x[-n:].cpu().numpy() for x in (dq, dqd, dtau))
    err = max(np.abs(ah[:n] - om.aba(q[:n], qd[:n], tau[:n], g)).max(), np.abs(ah[-n:] - om.aba(tq, tqd, ttau, g)).max())
    fn = lambda: hm.aba(dq, dqd, dtau, g)
    for _ in range(5): fn()
    best = 1e9
    for r in range(4):
        t = HipTimer(); t.start(stream)
        for _ in range(20): fn()
        t.stop(stream)
        best = min(best, t.elapsed_ms() / 20 * 1e3)
    print(f"[{tag}] {hm.kernel_variant[:40]} B={B}: err {err:.2e} nan {int(np.isnan(ah).sum())}  {best:.1f} us  {B / best / 1e3:.3f} G/s", flush=True)
