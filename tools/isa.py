"""Experiment helper: compile the humanoid's specialised code object with -DMH_SPEC_MINIMAL (only the ident + LDS-IO tree-split kernels)
either to ISA text (default) or to the .so bench.py loads (--so).  python tools/isa.py [--so] [extra hipcc flags]"""
import subprocess, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from mecano_amd import build as b

name = os.environ.get("ISA_MODEL", "humanoid30")  # ISA_MODEL=arm7 ISA_FULL=1: the chain's whole-tree kernels (not in a minimal build)
desc = b.registered_models()[name]
key, parents, kinds = b.topology_of(desc)
extra = [a for a in sys.argv[1:] if a != "--so"]
if not os.environ.get("ISA_FULL"):
    extra.insert(0, "-DMH_SPEC_MINIMAL")
defs, extra = b.spec_defines(parents, kinds, extra), []  # (tree, source hash the library checks at load, the extra flags and their hash)
t = time.time()
if "--so" in sys.argv:
    out_dir = os.environ.get("EXP_DIR", "build/exp")  # git-ignored, travels to the GPU box (gpurun_out/ does not)
    os.makedirs(out_dir, exist_ok=True)  # never clobbers the shipped code object: run with MH_SPEC_DIR=build/exp
    cmd = [b.hipcc()] + b.SPEC_FLAGS + defs + extra + ["-o", os.path.join(out_dir, os.path.basename(b.spec_path(key))), b.SPEC_SOURCE]
else:
    os.makedirs("gpurun_out/isa", exist_ok=True)
    cmd = [b.hipcc()] + [f for f in b.SPEC_FLAGS if f not in ("-shared", "-fPIC")] + defs + extra + ["--cuda-device-only", "-S", "-o", "gpurun_out/isa/min.s", b.SPEC_SOURCE]
subprocess.check_call(cmd)
print("built in %.0f s" % (time.time() - t))
