"""Config 5 (128-body mixed tree, fp32, B = 131072): AoS forward dynamics through transposed scratch copies vs through LDS row windows
(MH_DFS_TRANSPOSE=0), and the SoA figures beside them."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mecano_amd import random_tools as rt
    from mecano_amd.engine import HipModel, HipTimer
    from mecano_amd.multibody import MultiBodySystem
    def timeit(fn, stream, iters=5, warm=2):
        for _ in range(warm): fn()
        t = HipTimer(); t.start(stream)
        for _ in range(iters): fn()
        t.stop(stream)
        return t.elapsed_ms() / iters * 1e3
    stream = torch.cuda.current_stream().cuda_stream
    tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
    hm = HipModel(tree.toModelDesc())
    B, base = 131072, 8192
    f32 = torch.float32
    q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=f32).repeat(B // base, 1).contiguous() for x in rt.nextState(np.random.default_rng(1), tree, base))
    T = lambda x: x.t().contiguous()
    qs, qds, qdds, taus = T(q), T(qd), T(qdd), T(tau)
    g = (0, 0, -9.81)
    a1, a2 = hm.aba(q, qd, tau, g), hm.aba(qs, qds, taus, g, layout=1).t()
    print({k: os.environ.get(k) for k in ("MH_DFS_TRANSPOSE",)}, "aos == soa:", bool(torch.equal(a1, a2)), "rnea aos %.0f soa %.0f | aba aos %.0f soa %.0f us" % (
        timeit(lambda: hm.rnea(q, qd, qdd, g), stream), timeit(lambda: hm.rnea(qs, qds, qdds, g, layout=1), stream),
        timeit(lambda: hm.aba(q, qd, tau, g), stream), timeit(lambda: hm.aba(qs, qds, taus, g, layout=1), stream)), flush=True)
else:
    for env in ({}, {"MH_DFS_TRANSPOSE": "0"}, {"MH_DFS_TRANSPOSE": "1"}):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env))
