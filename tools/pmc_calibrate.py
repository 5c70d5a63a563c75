"""Counter calibration workload: mh_integrate_f64, SoA, B = 1M -- an 8-byte-per-lane streaming kernel with exactly known traffic
(reads (nq + 2 nv) * 8 B, writes (nq + nv) * 8 B per configuration, far beyond the 256 MiB Infinity Cache).  Run under
rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) to learn how those counters tally 8-byte-per-lane accesses."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from mecano_amd import _lib, build as b
from mecano_amd.engine import HipModel
desc = b.registered_models()["humanoid30"]
hm = HipModel(desc)
B = 1 << 20
q = torch.randn((desc.nq, B), device="cuda", dtype=torch.float64); qd = torch.randn((desc.nv, B), device="cuda", dtype=torch.float64); qdd = torch.randn_like(qd)
qo, vo = torch.empty_like(q), torch.empty_like(qd)
for _ in range(10):
    hm.integrate(1e-3, q, qd, qdd, _lib.LAYOUT_SOA, out=(qo, vo))
torch.cuda.synchronize()
print("true bytes per launch: read", B * (desc.nq + 2 * desc.nv) * 8, "written", B * (desc.nq + desc.nv) * 8)
