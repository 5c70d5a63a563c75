"""mh_rnea_f64 on the humanoid at device-filling batches: the persistent loop that requests the next group's rows behind the trunk pass
(default from three groups of 64 configurations per CU) against the tree-split kernel's own loop (MH_RNEA_AHEAD=0)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(sys_.toModelDesc())
    print(hm.kernel_variant, flush=True)
    stream = torch.cuda.current_stream().cuda_stream
    g = (0.0, 0.0, -9.81)
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("MH_") and k != "MH_SPEC_DIR")
    for B in [int(a) for a in sys.argv[2:]]:
        q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(B), sys_, B))
        fn = lambda: hm.rnea(q, qd, qdd, g)
        ref = None
        for _ in range(5): fn()
        best = 1e9
        for r in range(4):
            t = HipTimer(); t.start(stream)
            for _ in range(30): fn()
            t.stop(stream)
            best = min(best, t.elapsed_ms() / 30 * 1e3)
        chk = float(hm.rnea(q, qd, qdd, g).double().abs().sum())
        print(f"[{tag}] sum|tau| {chk:.12e} RNEA B={B}: {best:.1f} us  {B / best / 1e3:.3f} G/s", flush=True)
else:
    for env in ({}, {"MH_RNEA_AHEAD": "0"}):
        subprocess.run([sys.executable, __file__, "child"] + sys.argv[1:], env=dict(os.environ, **env))
