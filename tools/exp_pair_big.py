"""mh_rnea_aba_f64 on the humanoid at batches between the headline and device-filling sizes: the one-launch forms against two calls
(MH_DISABLE_FUSED=1: mh_rnea_f64, then mh_aba_f64 -- which takes the fused forward-dynamics kernel beyond one group per CU)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(sys_.toModelDesc())
    stream = torch.cuda.current_stream().cuda_stream
    g = (0.0, 0.0, -9.81)
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MH_"))
    for B in [int(a) for a in sys.argv[2:]]:
        q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(B), sys_, B))
        o1, o2 = torch.empty_like(qd), torch.empty_like(qd)
        fn = hm.bind_rnea_aba(q, qd, qdd, tau, o1, o2, g)
        for _ in range(5): fn()
        best = 1e9
        for r in range(4):
            t = HipTimer(); t.start(stream)
            for _ in range(30): fn()
            t.stop(stream)
            best = min(best, t.elapsed_ms() / 30 * 1e3)
        print(f"[{tag}] pair B={B}: {best:.1f} us  {B / best / 1e3:.3f} G pairs/s", flush=True)
else:
    for env in ({}, {"MH_DISABLE_FUSED": "1"}):
        subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:], env=dict(os.environ, **env))
