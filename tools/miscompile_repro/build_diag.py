import sys, time, subprocess, os
sys.path.insert(0, '.')
from mecano_amd import build as b
d = b.registered_models()['humanoid30']
key, parents, kinds = b.topology_of(d)
out = b.spec_path(key)
if os.environ.get('DIAG_OUT'):
    os.makedirs(os.environ['DIAG_OUT'], exist_ok=True)
    out = os.path.join(os.environ['DIAG_OUT'], os.path.basename(out))
flags = sys.argv[1:]
cmd = [b.hipcc()] + b.FLAGS + [f"-DMH_TOPO_N={len(parents)}", "-DMH_TOPO_PARENTS=" + ",".join(str(int(x)) for x in parents),
      "-DMH_TOPO_TYPES=" + ",".join(str(int(x)) for x in kinds), "-DMH_DIAG_ONLY"] + (flags if any(f.startswith("-DMH_DIAG_FLAGS") for f in flags) else flags + ["-DMH_DIAG_FLAGS=3"]) + ["-o", out, b.SPEC_SOURCE]
t = time.time(); subprocess.check_call(cmd); print("built %s in %.0f s" % (out, time.time() - t))
