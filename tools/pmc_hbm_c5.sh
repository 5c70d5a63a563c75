#!/bin/bash
# usage (GPU box, repo root): tools/pmc_hbm_c5.sh   FETCH_SIZE / WRITE_SIZE (separate passes) + kernel stats of config 5's RNEA / ABA (fp32, B = 131072, AoS)
root=$(pwd); out=$root/gpurun_out/pmc_c5hbm; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/$c -o pmc --output-format csv -- python3 $root/tools/pmc_dfs.py ${LAYOUT:-c5aos} > $out/$c.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/tools/pmc_dfs.py ${LAYOUT:-c5aos} > $out/trace.log 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
B, nq, nv = 131072, 362, 323
alg = B * 4 * (nq + 3 * nv)
for k, cs in acc.items():
    if "dfs" not in k:
        continue
    f = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])); w = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"]))
    res[k] = {"FETCH_SIZE_KB_per_launch_mean": f, "WRITE_SIZE_KB_per_launch_mean": w, "FETCH_SIZE_correction": 2.0,
              "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024, "algorithmic_bytes_per_launch": alg, "ratio": (2 * f + w) * 1024 / alg}
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dfs" in r["Name"]:
            res.setdefault(r["Name"][:48], {})["avg_ns"] = float(r["AverageNs"]); res[r["Name"][:48]]["calls"] = int(r["Calls"])
print(json.dumps(res, indent=1))
PY
