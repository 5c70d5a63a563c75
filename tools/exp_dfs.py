"""Measurement helper: the depth-first run-time-topology kernels (mh_dfs_kernels.h) against the sweep kernels they replace, per memory
placement.  Config 5 (128-body tree, fp32), the humanoid without its code object, the reference's 30-joint benchmark shapes.
Usage: python tools/exp_dfs.py [B5]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem

stream = torch.cuda.current_stream().cuda_stream
g = (0.0, 0.0, -9.81)
dev = lambda x, dt=torch.float64: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dt)
T = lambda x: x.t().contiguous()


def timeit(fn, iters=5, warm=3):
    for _ in range(warm):
        fn()
    t = HipTimer()
    t.start(stream)
    for _ in range(iters):
        fn()
    t.stop(stream)
    return t.elapsed_ms() / iters * 1e-3


def model(desc, **env):
    keys = ("MH_DFS", "MH_DFS_PLACE", "MH_DISABLE_SPEC", "MH_WAVES_PER_CU", "MH_DFS_WIN", "MH_DFS_BUDGET", "MH_DFS_ABA64", "MH_DFS_TRANSPOSE")
    for k in keys:
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ[k] = str(v)
    m = HipModel(desc)
    for k in keys:
        os.environ.pop(k, None)
    return m


def row(name, B, secs, bytes_per_eval):
    print(f"{name:64s} B={B:7d} {secs * 1e6:9.1f} us {B / secs / 1e6:8.2f} M/s {B * bytes_per_eval / secs / 1e9:7.1f} GB/s ({B * bytes_per_eval / secs / 8e12 * 100:5.2f} %)", flush=True)


B5 = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
ONLY5 = len(sys.argv) > 2 and sys.argv[2] == "c5"
tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
d5 = tree.toModelDesc()
f32 = torch.float32
q, qd, qdd, tau = (dev(x, f32) for x in rt.nextState(np.random.default_rng(1), tree, B5))
qs, qds, qdds, taus = T(q), T(qd), T(qdd), T(tau)
by = 4 * (d5.nq + 3 * d5.nv)
for label, env in (("sweep kernels (round 1)", dict(MH_DFS=0)), ("dfs auto (frames near the leaves in LDS)", {}), ("dfs auto, no row windows", dict(MH_DFS_WIN=0)),
                   ("dfs auto, AoS always via transposed copies", dict(MH_DFS_TRANSPOSE=1)), ("dfs auto, AoS never via transposed copies", dict(MH_DFS_TRANSPOSE=0)),
                   ("dfs all LDS", dict(MH_DFS_PLACE=0)), ("dfs all global", dict(MH_DFS_PLACE=2)), ("dfs all global, no row windows", dict(MH_DFS_PLACE=2, MH_DFS_WIN=0)),
                   ("dfs LDS budget 16 slots", dict(MH_DFS_BUDGET=16)), ("dfs LDS budget 32 slots", dict(MH_DFS_BUDGET=32)), ("dfs LDS budget 48 slots", dict(MH_DFS_BUDGET=48)),
                   ("dfs LDS budget 64 slots, 6 waves per CU", dict(MH_DFS_BUDGET=64, MH_WAVES_PER_CU=6)), ("dfs 4 waves per CU", dict(MH_WAVES_PER_CU=4))):
    hm = model(d5, **env)
    row(f"C5 RNEA fp32 AoS  {label}", B5, timeit(lambda: hm.rnea(q, qd, qdd, g)), by)
    row(f"C5 RNEA fp32 SoA  {label}", B5, timeit(lambda: hm.rnea(qs, qds, qdds, g, layout=_lib.LAYOUT_SOA)), by)
    row(f"C5 ABA  fp32 AoS  {label}", B5, timeit(lambda: hm.aba(q, qd, tau, g)), by)
    row(f"C5 ABA  fp32 SoA  {label}", B5, timeit(lambda: hm.aba(qs, qds, taus, g, layout=_lib.LAYOUT_SOA)), by)

if ONLY5:
    sys.exit(0)
hum = rt.nextHumanoid(np.random.default_rng(43))
dh = hum.toModelDesc()
for B in (4096, 32768, 262144):
    q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(2342), hum, B))
    by = 8 * (dh.nq + 3 * dh.nv)
    for label, env in (("specialised", {}), ("sweep kernels", dict(MH_DISABLE_SPEC=1, MH_DFS=0)), ("auto (RNEA depth-first, fp64 ABA sweep)", dict(MH_DISABLE_SPEC=1)), ("dfs auto, ABA too", dict(MH_DISABLE_SPEC=1, MH_DFS_ABA64=1)),
                       ("dfs all LDS", dict(MH_DISABLE_SPEC=1, MH_DFS_PLACE=0)), ("dfs stack LDS", dict(MH_DISABLE_SPEC=1, MH_DFS_PLACE=1)),
                       ("dfs global", dict(MH_DISABLE_SPEC=1, MH_DFS_PLACE=2))):
        hm = model(dh, **env)
        tr, ta = timeit(lambda: hm.rnea(q, qd, qdd, g), 20, 5), timeit(lambda: hm.aba(q, qd, tau, g), 20, 5)
        row(f"humanoid RNEA fp64 {label} [{hm.kernel_variant[:12]}]", B, tr, by)
        row(f"humanoid ABA  fp64 {label}", B, ta, by)
        print(f"      RNEA + ABA = {(tr + ta) * 1e6:.1f} us", flush=True)

for name, sys_ in rt.referenceBenchmarkSystems().items():
    d = sys_.toModelDesc()
    for B in (4096, 262144):
        q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(2342), sys_, B))
        by = 8 * (d.nq + 3 * d.nv)
        for label, env in (("sweep kernels", dict(MH_DFS=0)), ("auto", {}), ("dfs ABA too", dict(MH_DFS_ABA64=1))):
            hm = model(d, **env)
            row(f"{name} RNEA fp64 {label}", B, timeit(lambda: hm.rnea(q, qd, qdd, g), 10, 3), by)
            row(f"{name} ABA  fp64 {label}", B, timeit(lambda: hm.aba(q, qd, tau, g), 10, 3), by)
