"""Experiment: config 5 (128-body mixed tree, fp32) generic kernels vs resident waves per CU (MH_WAVES_PER_CU)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mecano_amd import _lib, random_tools as rt
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
    B = 131072
    f32 = torch.float32
    q, qd, qdd, tau = (torch.tensor(x, device="cuda", dtype=f32) for x in rt.nextState(np.random.default_rng(1), tree, B))
    T = lambda x: x.t().contiguous()
    qs, qds, qdds, taus = T(q), T(qd), T(qdd), T(tau)
    g = (0, 0, -9.81)
    print(os.environ.get("MH_WAVES_PER_CU"), "rnea aos %.0f soa %.0f | aba aos %.0f soa %.0f us" % (
        timeit(lambda: hm.rnea(q, qd, qdd, g), stream), timeit(lambda: hm.rnea(qs, qds, qdds, g, layout=1), stream),
        timeit(lambda: hm.aba(q, qd, tau, g), stream), timeit(lambda: hm.aba(qs, qds, taus, g, layout=1), stream)), flush=True)
else:
    for lds in (0, 1):
        print("MH_GENERIC_LDS =", lds, flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, MH_GENERIC_LDS=str(lds)))
