#!/bin/bash
# usage (GPU box, repo root): tools/pmc_kernels.sh <B> "<counter>" ["<counter>" ...]   one rocprofv3 --pmc pass per counter
B=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rocprofv3 --pmc $c -d $out/k_$c -o pmc --output-format csv -- python3 $root/tools/pmc_kernels.py $B > /dev/null 2>&1
done
cd $root
python3 - "$out" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/k_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{c:12s} {sum(v) / len(v):12.1f} per launch ({len(v):3d} launches)  {k}")
PY
