"""Diagnostic: create the registered models repeatedly and print what the create-time self-check says."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from mecano_amd import build as b
from mecano_amd.engine import HipModel
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
models = b.registered_models()
for name, desc in models.items():
    bad = {}
    for i in range(n):
        v = HipModel(desc).kernel_variant
        if not v.startswith("topo:"):
            bad[v] = bad.get(v, 0) + 1
    print(name, "refused", sum(bad.values()), "of", n)
    for v, c in bad.items():
        print("   ", c, "x", v)
