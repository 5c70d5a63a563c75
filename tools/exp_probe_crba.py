"""Phase stamps of the tree-split CRBA kernel (code object built with -DMH_PROBE): 0 entry, 1 context, 2 limbs, 3 barrier, 4 trunk (wave 0), 5 barrier, 6 written, 7 barrier."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from mecano_amd import _lib, build as b
from mecano_amd.engine import HipModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
os.environ["MH_SPEC_SPLIT"] = "1"
desc = b.registered_models()["humanoid30"]
q = torch.randn(B, desc.nq, device="cuda", dtype=torch.float64); q[:, :4] /= q[:, :4].norm(dim=1, keepdim=True)
lpg = int(os.environ.get("MH_CRBA_LPG", "16"))  # the stamps are indexed by workgroup: one per lpg configurations
os.environ["MH_CRBA_LPG"] = str(lpg)
ns = (B + lpg - 1) // lpg
hm = HipModel(desc); lib = _lib.load()
print(hm.kernel_variant)
out = torch.zeros(B * desc.nv * desc.nv + ns * 32, device="cuda", dtype=torch.float64)
opts = hm._options(_lib.LAYOUT_AOS, True, True, torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    _lib.check(lib.mh_crba_f64(hm._h, B, q.data_ptr(), ctypes.byref(opts), out.data_ptr())); torch.cuda.synchronize()
torch.cuda.synchronize()
st = out[B * desc.nv * desc.nv:].cpu().numpy().view(np.uint64).reshape(ns, 4, 8).astype(np.int64)
d = (st - st[:, :, :1]) * (1.0 if os.environ.get("STAMP_TICKS") else 0.01)  # 100 MHz ticks -> us
for w in range(4):
    print(f"wave {w}: " + "  ".join(f"s{k}={np.median(d[:, w, k]):7.2f}" for k in range(8)))
