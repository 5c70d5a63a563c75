"""Measurement helper (not part of the product): BASELINE.json's five configurations on ONE GPU at their per-GPU shard sizes, plus the
Coriolis / centroidal kernels.  HIP events on the launch stream, inputs device-resident; prints time per call, evaluations / s and the
algorithmic GB/s (inputs read once + outputs written once).  Usage: python tools/bench_configs.py [quick]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer
from mecano_amd.multibody import MultiBodySystem


def timeit(fn, stream, iters=20, warm=10):
    """Seconds per call: `iters` calls between two HIP events.  Calls that take less than the host needs to issue one (the Python mirror
    allocates the output and marshals the arguments: 3-15 us depending on the box) are timed a second time as ONE replay of a captured
    graph of the same `iters` calls (the entry points are capturable: tests/test_gpu_parity.py::test_entry_points_are_graph_capturable),
    which takes the host out of the loop; the smaller figure is the device's."""
    for _ in range(warm):
        fn()
    t = HipTimer()
    t.start(stream)
    for _ in range(iters):
        fn()
    t.stop(stream)
    eager = t.elapsed_ms() / iters * 1e-3
    if eager > 60e-6:
        return eager
    try:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        t = HipTimer()
        t.start(stream)
        graph.replay()
        t.stop(stream)
        return min(eager, t.elapsed_ms() / iters * 1e-3)
    except Exception as e:  # a path that cannot be captured: the eager figure stands
        print(f"   (graph replay unavailable: {type(e).__name__}: {e})", flush=True)
        torch.cuda.synchronize()
        return eager


def report(name, B, secs, bytes_per_eval):
    print(f"{name:58s} B={B:8d}  {secs * 1e6:9.1f} us  {B / secs / 1e6:9.2f} M/s  {B * bytes_per_eval / secs / 1e9:8.1f} GB/s "
          f"({B * bytes_per_eval / secs / 8e12 * 100:5.2f} % of 8 TB/s)", flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    stream = torch.cuda.current_stream().cuda_stream
    g = (0.0, 0.0, -9.81)
    dev = lambda x, dt=torch.float64: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=dt)
    T = lambda x: x.t().contiguous()
    # ---- C2: 7-DoF arm, RNEA fp64, B = 1024
    arm = MultiBodySystem.toMultiBodySystemInput(rt.nextJointChain(np.random.default_rng(43), 7, ("revolute",))[0].getPredecessor())
    hm = HipModel(arm.toModelDesc())
    for B in (1024, 262144):
        q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(2342), arm, B))
        report(f"C2 arm7 RNEA fp64 [{hm.kernel_variant}]", B, timeit(lambda: hm.rnea(q, qd, qdd, g), stream), 8 * (7 * 4))
        report(f"   arm7 ABA  fp64", B, timeit(lambda: hm.aba(q, qd, tau, g), stream), 8 * (7 * 4))
    # ---- C3 / C4: humanoid
    hum = rt.nextHumanoid(np.random.default_rng(43))
    hm = HipModel(hum.toModelDesc())
    nq, nv = hm.nq, hm.nv
    for B in (4096, 32768) if quick else (4096, 32768, 262144):
        q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(2342), hum, B))
        report(f"C3 humanoid RNEA fp64 [{hm.kernel_variant}]", B, timeit(lambda: hm.rnea(q, qd, qdd, g), stream), 8 * (nq + 3 * nv))
        report(f"C3 humanoid CRBA fp64", B, timeit(lambda: hm.crba(q), stream, iters=10), 8 * (nq + nv * nv))
        report(f"C4 humanoid ABA fp64", B, timeit(lambda: hm.aba(q, qd, tau, g), stream), 8 * (nq + 3 * nv))
        report(f"N3 humanoid mass + Coriolis matrix fp64 (specialised)", B, timeit(lambda: hm.crba_coriolis(q, qd), stream, iters=5), 8 * (nq + nv + 2 * nv * nv))
        report(f"N3 humanoid centroidal A, b at CoM fp64 (specialised)", B, timeit(lambda: hm.centroidal(q, qd, None, True), stream, iters=5), 8 * (nq + nv + 6 * nv + 9))
    # ---- C5: random 128-body tree, mixed joints, fp32, per-GPU shard of B = 1M over 8 GPUs
    tree = MultiBodySystem.toMultiBodySystemInput(rt.nextJointTree(np.random.default_rng(128), 128, ("revolute", "prismatic", "sixdof"))[0].getPredecessor())
    hm = HipModel(tree.toModelDesc())
    nq, nv = hm.nq, hm.nv
    f32 = torch.float32
    for B in (4096, 32768) if quick else (4096, 131072):
        q, qd, qdd, tau = (dev(x, f32) for x in rt.nextState(np.random.default_rng(1), tree, B))
        report(f"C5 tree128 (nq={nq}, nv={nv}) RNEA fp32 AoS [{hm.kernel_variant}]", B, timeit(lambda: hm.rnea(q, qd, qdd, g), stream, iters=5), 4 * (nq + 3 * nv))
        report(f"C5 tree128 ABA fp32 AoS", B, timeit(lambda: hm.aba(q, qd, tau, g), stream, iters=5), 4 * (nq + 3 * nv))
        qs, qds, qdds, taus = T(q), T(qd), T(qdd), T(tau)
        report(f"C5 tree128 RNEA fp32 SoA", B, timeit(lambda: hm.rnea(qs, qds, qdds, g, layout=_lib.LAYOUT_SOA), stream, iters=5), 4 * (nq + 3 * nv))
        report(f"C5 tree128 ABA fp32 SoA", B, timeit(lambda: hm.aba(qs, qds, taus, g, layout=_lib.LAYOUT_SOA), stream, iters=5), 4 * (nq + 3 * nv))


if __name__ == "__main__":
    main()
