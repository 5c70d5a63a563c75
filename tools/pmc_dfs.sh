#!/bin/bash
# usage (GPU box, repo root): tools/pmc_dfs.sh <hum|c5> <tag>    SQ counters of the run-time-topology kernels, one rocprofv3 --pmc pass per group
which=$1; tag=$2
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
groups=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE")
i=0
for grp in "${groups[@]}"; do
  rocprofv3 --pmc $grp -d $out/g$i -o pmc --output-format csv -- python3 $root/tools/pmc_dfs.py $which > $out/g$i.log 2>&1
  i=$((i+1))
done
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/tools/pmc_dfs.py $which > $out/trace.log 2>&1
cd $root
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "dfs" not in k and "rnea_kernel" not in k and "aba_kernel" not in k and "regressor" not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:22s} {sum(v) / len(v):14.1f} per launch ({len(v)} launches)")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("stats", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
