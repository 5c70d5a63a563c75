#!/bin/bash
# usage (GPU box, repo root): tools/pmc_zvf.sh <B> "<counter>" ...   one rocprofv3 --pmc pass per counter over tools/prof_zvb.py (forward dynamics, 30 calls)
B=$1; shift
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rocprofv3 --pmc $c -d $out/kz_$c -o pmc --output-format csv -- python3 $root/tools/prof_zvb.py $B > /dev/null 2>&1
done
cd $root
python3 - "$out" "$@" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for c in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/kz_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and int(r["Grid_Size"]) >= 512 * 256:
                acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{c:22s} {sum(v) / len(v):14.1f} per launch ({len(v):3d} launches, grid >= 512 workgroups)  {k}")
PY
