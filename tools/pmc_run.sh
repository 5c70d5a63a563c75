#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <outdir> <B> "<counters pass 1>" "<counters pass 2>" ...
# one rocprofv3 --pmc pass per counter group (never combined with trace flags), then a per-kernel sum table
out=$1; B=$2; shift 2
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $root/$out/p$i -o pmc --output-format csv -- python3 $root/tools/pmc_rnea.py $B > /dev/null 2>&1
done
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, k, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] == list(acc[k])[0]:
            seen.add(key)
    for k in acc:
        pass
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v:16.0f}")
PY
