"""Diagnostic: which HIP runtimes does the process hold and what do they see (run on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
print({k: v for k, v in os.environ.items() if "VISIBLE" in k or "HSA" in k or "HIP" in k or "ROC" in k})
from mecano_amd import _lib
lib = _lib.load()
n = ctypes.c_int32(-1)
print("before torch: status", lib.mh_device_count(ctypes.byref(n)), "count", n.value, lib.mh_last_error())
import torch
print("torch", torch.cuda.is_available(), torch.cuda.device_count())
print("after torch: status", lib.mh_device_count(ctypes.byref(n)), "count", n.value, lib.mh_last_error())
print([l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l][::4])
