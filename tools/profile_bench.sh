#!/bin/bash
# usage (GPU box, repo root): tools/profile_bench.sh <tag>
# rocprofv3 kernel-trace statistics of the default bench.py run, then the HBM counters in their own passes (never combined with
# trace flags).  Summaries land in gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=${1:-prof}
export MH_BENCH_NO_PMC=1   # bench.py must not start profiler children of its own from under a profiler
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o bench --output-format csv -- python3 $root/bench.py > $out/${tag}_bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/${tag}_pmc_$c -o pmc --output-format csv -- python3 $root/bench.py --steps 20 --warmup 5 > /dev/null 2>&1
done
cd $root
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, shutil
out, tag = sys.argv[1], sys.argv[2]
stats = glob.glob(f"{out}/{tag}_trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"{out}/{tag}_kernel_stats.csv")
    print(open(stats[0]).read()[:1500])
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"{out}/{tag}_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("fused_split" in r["Kernel_Name"] or "zv_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == c and int(r["Grid_Size"]) >= 128 * 256:
                vals.append(float(r["Counter_Value"]))
    if vals:
        res[f"{c}_KB_per_launch_mean"] = sum(vals) / len(vals)
        res[f"{c}_launches"] = len(vals)
json.dump(res, open(f"{out}/{tag}_hbm_pmc.json", "w"), indent=1)
print(res)
PY
