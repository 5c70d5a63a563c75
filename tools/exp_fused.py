"""Experiment: one fused RNEA+ABA launch vs two launches over batch sizes (MH_FUSED_FACTOR widens the fused range)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    sys_ = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(sys_.toModelDesc())
    stream = torch.cuda.current_stream().cuda_stream
    g = (0, 0, -9.81)
    out = []
    for B in (4096, 8192, 12288, 16384, 24576, 32768, 65536):
        q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(1), sys_, B))
        for _ in range(3): hm.rnea_aba(q, qd, qdd, tau, g)
        t = HipTimer(); t.start(stream)
        for _ in range(20): hm.rnea_aba(q, qd, qdd, tau, g)
        t.stop(stream)
        out.append("%d: %.1f us" % (B, t.elapsed_ms() / 20 * 1e3))
    print("MH_FUSED_FACTOR =", os.environ.get("MH_FUSED_FACTOR"), " | ".join(out), flush=True)
else:
    for f in (1, 2, 4, 8):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, MH_FUSED_FACTOR=str(f)))
