"""BASELINE.json config 2 (7-DoF arm, RNEA, fp64, B = 1024): where do the ~10.5 us per call go?  

  eager        back-to-back calls on one stream, HIP events around N of them: device time + whatever of the host's launch path the GPU waits for
  graph        K captured launches replayed as one graph: the host is out of the loop, what is left is device time + the kernel boundary
  empty        the same two measurements for a kernel that does nothing (one wave, one store): the floor of a dependent launch on this stack
  stamps       (code object built with -DMH_PROBE) real-time stamps inside the kernel: entry -> rows staged -> walk done -> rows written

python tools/exp_c2_floor.py [B ...]        MH_SPEC_DIR=exp_probe_arm python tools/exp_c2_floor.py stamps [B]
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, random_tools as rt
from mecano_amd.engine import HipModel, HipTimer

arm = rt.committedBenchmarkSystems()["arm7"]
desc = rt.modelDescFromJson("arm7")
hm = HipModel(desc)
g = (0.0, 0.0, -9.81)
stream = torch.cuda.Stream()


def per_call_eager(fn, n=400):
    with torch.cuda.stream(stream):
        for _ in range(50):
            fn()
        t = HipTimer()
        t.start(stream.cuda_stream)
        for _ in range(n):
            fn()
        t.stop(stream.cuda_stream)
        return t.elapsed_ms() / n * 1e3


def per_call_graph(fn, k=50, replays=40):
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            for _ in range(k):
                fn()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        t = HipTimer()
        t.start(stream.cuda_stream)
        for _ in range(replays):
            gr.replay()
        t.stop(stream.cuda_stream)
        return t.elapsed_ms() / (k * replays) * 1e3


if len(sys.argv) > 1 and sys.argv[1] == "stamps":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    lib = _lib.load()
    q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(2342), arm, B))
    nslice = (B + 63) // 64
    for algo, fn, x in (("rnea", lib.mh_rnea_f64, qdd), ("aba", lib.mh_aba_f64, tau)):
        out = torch.zeros(B * desc.nv + nslice * 32, device="cuda", dtype=torch.float64)
        gv = (ctypes.c_double * 3)(*g)
        opts = hm._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream().cuda_stream)
        for rep in range(6):
            _lib.check(fn(hm._h, B, q.data_ptr(), qd.data_ptr(), x.data_ptr(), gv, None, ctypes.byref(opts), out.data_ptr()))
            torch.cuda.synchronize()
        st = out[B * desc.nv:].cpu().numpy().view(np.uint64).reshape(nslice, 32)[:, :5].astype(np.int64)
        t0 = st[:, 0].min()
        d = lambda a, z: "%.2f" % (np.median(st[:, z] - st[:, a]) / 100.0)
        extra = f"  inward {d(1, 4)}  outward {d(4, 2)}" if algo == "aba" else ""
        print(f"{hm.kernel_variant} {algo} B={B}: us per slice (median over {nslice} waves): stage {d(0, 1)}  walk {d(1, 2)}{extra}  copy-out {d(2, 3)}  "
              f"| entry skew {(st[:, 0].max() - t0) / 100.0:.2f}  kernel span {(st[:, 3].max() - t0) / 100.0:.2f}")
    sys.exit(0)

print("variant", hm.kernel_variant)
e = torch.zeros(64, device="cuda")
empty = lambda: e.add_(1.0)  # one wave, one load, one store
print(f"empty kernel (torch add_ on 64 floats): eager {per_call_eager(empty):.2f} us / launch, graph of 50 {per_call_graph(empty):.2f} us / launch")
for B in [int(a) for a in sys.argv[1:]] or [64, 1024, 4096, 16384]:
    q, qd, qdd, tau = (torch.tensor(x, device="cuda") for x in rt.nextState(np.random.default_rng(2342), arm, B))
    with torch.cuda.stream(stream):
        hm.reserve(B)
    o = torch.empty_like(qd)
    lib = _lib.load()
    gv = (ctypes.c_double * 3)(*g)
    opts = hm._options(_lib.LAYOUT_AOS, True, True, stream.cuda_stream)
    rnea = lambda: lib.mh_rnea_f64(hm._h, B, q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), gv, None, ctypes.byref(opts), o.data_ptr())
    aba = lambda: lib.mh_aba_f64(hm._h, B, q.data_ptr(), qd.data_ptr(), tau.data_ptr(), gv, None, ctypes.byref(opts), o.data_ptr())
    for name, fn in (("RNEA", rnea), ("ABA", aba)):
        print(f"arm7 {name} B={B}: eager {per_call_eager(fn):.2f} us / call, graph of 50 launches {per_call_graph(fn):.2f} us / launch", flush=True)
