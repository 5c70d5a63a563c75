"""Per-phase s_memtime stamps of the tree-split kernels (needs the code object built with -DMH_PROBE: python tools/isa.py --so -DMH_PROBE).
Stamps per 64-configuration slice and wave: 0 entry, 1 rows staged, 2 limbs done, 3 past barrier, 4 trunk inward done (ABA),
5 trunk outward done, 6 past barrier, 7 results copied out."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, build as b
from mecano_amd.engine import HipModel

algo = sys.argv[1] if len(sys.argv) > 1 else "aba"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
desc = b.registered_models()["humanoid30"]
hm = HipModel(desc)
lib = _lib.load()
q = torch.randn(B, desc.nq, device="cuda", dtype=torch.float64); q[:, :4] /= q[:, :4].norm(dim=1, keepdim=True)
qd = torch.randn(B, desc.nv, device="cuda", dtype=torch.float64); x = torch.randn_like(qd)
nslice = (B + 63) // 64
out = torch.zeros(B * desc.nv + nslice * 32, device="cuda", dtype=torch.float64)
g = (ctypes.c_double * 3)(0, 0, -9.81)
opts = hm._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream().cuda_stream)
fn = lib.mh_rnea_f64 if algo == "rnea" else lib.mh_aba_f64
for rep in range(6):
    _lib.check(fn(hm._h, B, q.data_ptr(), qd.data_ptr(), x.data_ptr(), g, None, ctypes.byref(opts), out.data_ptr()))
    torch.cuda.synchronize()
st = out[B * desc.nv:].cpu().numpy().view(np.uint64).reshape(nslice, 4, 8).astype(np.int64)
t0 = st[:, :, 0].min()
print("variant", hm.kernel_variant, "algo", algo, "B", B, " (ticks relative to the first wave's entry; 100 ticks = 1 us if the counter runs at 100 MHz)")
d = st - st[:, :, :1]
names = ["entry", "staged", "limbs", "barrier1", "trunk_in", "trunk_out", "barrier2", "copied"]
for w in range(4):
    print(f"wave {w}: " + "  ".join(f"{n}={np.median(d[:, w, k]):7.0f}" for k, n in enumerate(names)))
print("entry skew across slices (ticks): min %d  median %d  max %d" % ((st[:, 0, 0] - t0).min(), np.median(st[:, 0, 0] - t0), (st[:, 0, 0] - t0).max()))
print("kernel span (last stamp - first entry): %d ticks" % (st[:, :, 7].max() - t0))
for sl in (0, nslice // 2, nslice - 1):
    print("slice", sl, "wave0 stamps:", (st[sl, 0] - t0).tolist(), " wave1:", (st[sl, 1] - t0).tolist())
