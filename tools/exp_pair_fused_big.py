"""mh_rnea_aba_f64 at device-filling batches: ONE launch (the fused forward-dynamics kernel with the M(q) qdd phase: MH_ZVF_PAIR=1, default)
against the two launches (MH_ZVF_PAIR=0), with the results compared to the single calls and to the oracle."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    from oracle.cpu_oracle import OracleModel
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    d = sys_.toModelDesc()
    hm, om = HipModel(d), OracleModel(d)
    stream = torch.cuda.current_stream().cuda_stream
    g = (0.3, -0.2, -9.81)
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MH_Z"))
    for B in [int(a) for a in sys.argv[2:]]:
        base = min(B, 8192)
        st = rt.nextState(np.random.default_rng(B), sys_, base)
        rep = (B + base - 1) // base
        q, qd, qdd, tau = (torch.tensor(x, device="cuda").repeat(rep, 1)[:B].contiguous() for x in st)
        o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
        fn = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, g)
        fn(); torch.cuda.synchronize()
        t1, a1 = hm.rnea(q, qd, qdd, g), hm.aba(q, qd, tau, g)
        idx = np.arange(0, min(B, base), 97)
        tr, ar = om.rnea(st[0][idx], st[1][idx], st[2][idx], g), om.aba(st[0][idx], st[1][idx], st[3][idx], g)
        e_t = float((o1 - t1).abs().max() / t1.abs().max()); e_a = float((o2 - a1).abs().max() / a1.abs().max())
        o_t = float(np.abs(o1.cpu().numpy()[idx] - tr).max() / np.abs(tr).max()); o_a = float(np.abs(o2.cpu().numpy()[idx] - ar).max() / np.abs(ar).max())
        for _ in range(5): fn()
        best = 1e9
        for r in range(4):
            t = HipTimer(); t.start(stream)
            for _ in range(20): fn()
            t.stop(stream)
            best = min(best, t.elapsed_ms() / 20 * 1e3)
        print(f"[{tag}] pair B={B}: {best:.1f} us  {B / best / 1e3:.3f} G pairs/s | vs single calls tau {e_t:.1e} qdd {e_a:.1e} | vs oracle tau {o_t:.1e} qdd {o_a:.1e}", flush=True)
else:
    for env in ({}, {"MH_ZVF_PAIR": "0"}):
        subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:], env=dict(os.environ, **env))
