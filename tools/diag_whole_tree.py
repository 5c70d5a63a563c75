"""Diagnosis helper for DESIGN.md section 10 (the whole-tree ABA that returned wrong results in round 1): build the humanoid's code
object WITH the whole-tree ABA (-DMH_FORCE_WHOLE_TREE_ABA) into exp_build/wt (python tools/diag_whole_tree.py build [flags]), then on a
GPU box create the model with the self-check in verbose mode (python tools/diag_whole_tree.py check): every plan's error against the
run-time-topology kernels is printed, a failing plan refuses the object."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from mecano_amd import build as b

desc = b.registered_models()["humanoid30"]
key, parents, kinds = b.topology_of(desc)
out_dir = os.path.join("exp_build", "wt")
if sys.argv[1] == "build":
    os.makedirs(out_dir, exist_ok=True)
    defs = [f"-DMH_TOPO_N={len(parents)}", "-DMH_TOPO_PARENTS=" + ",".join(str(int(x)) for x in parents), "-DMH_TOPO_TYPES=" + ",".join(str(int(x)) for x in kinds),
            "-DMH_FORCE_WHOLE_TREE_ABA"]
    t = time.time()
    subprocess.check_call([b.hipcc()] + b.FLAGS + defs + sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(out_dir, os.path.basename(b.spec_path(key))), b.SPEC_SOURCE])
    print("built in %.0f s" % (time.time() - t))
else:
    import numpy as np, torch
    from mecano_amd import _lib, random_tools as rt
    from mecano_amd.engine import HipModel
    hum = rt.nextHumanoid(np.random.default_rng(0))
    g = (0.0, 0.0, -9.81)
    dev = lambda x: torch.tensor(np.ascontiguousarray(x), device="cuda", dtype=torch.float64)

    def model(**env):
        keys = ("MH_SPEC_DIR", "MH_SPEC_SELFCHECK", "MH_SPEC_SPLIT", "MH_SPEC_IO", "MH_SPEC_ST", "MH_DISABLE_SPEC", "MH_DISABLE_FUSED")
        for k in keys:
            os.environ.pop(k, None)
        for k, v in env.items():
            os.environ[k] = str(v)
        m = HipModel(desc)
        for k in keys:
            os.environ.pop(k, None)
        return m

    ref = model(MH_DISABLE_SPEC=1)
    for B in (64, 197, 4096, 40000):
        q, qd, qdd, tau = (dev(x) for x in rt.nextState(np.random.default_rng(B), hum, B))
        for zero_v in (False, True):
            qdv = torch.zeros_like(qd) if zero_v else qd
            want = ref.aba(q, qdv, tau, g)
            for io in (0, 1):
                for st in (0, 1):
                    hm = model(MH_SPEC_DIR=out_dir, MH_SPEC_SELFCHECK=0, MH_SPEC_SPLIT=0, MH_SPEC_IO=io, MH_SPEC_ST=st, MH_DISABLE_FUSED=1)
                    for layout, name in ((_lib.LAYOUT_AOS, "AoS"), (_lib.LAYOUT_SOA, "SoA")):
                        if layout == _lib.LAYOUT_SOA:
                            got = hm.aba(q.t().contiguous(), qdv.t().contiguous(), tau.t().contiguous(), g, layout=layout).t()
                        else:
                            got = hm.aba(q, qdv, tau, g)
                        err = float((got - want).abs().max())
                        print(f"whole-tree ABA [{hm.kernel_variant[:24]}] B={B:6d} qd={'0' if zero_v else 'random'} rows-in-LDS={io} hand-over-in-LDS={st} {name}: max |err| = {err:.3e}"
                              f"  ({'OK' if err < 1e-8 else 'WRONG'})", flush=True)
