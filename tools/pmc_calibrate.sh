#!/bin/bash
# usage (GPU box, repo root): tools/pmc_calibrate.sh  -> gpurun_out/pmc_calibration.txt
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $out/cal_$c -o pmc --output-format csv -- python3 $root/tools/pmc_calibrate.py > $out/cal_$c.log 2>&1
done
cd $root
python3 - "$out" <<'PY' | tee $out/pmc_calibration.txt
import csv, glob, sys
out = sys.argv[1]
B, nq, nv = 1 << 20, 31, 30
true = {"FETCH_SIZE": B * (nq + 2 * nv) * 8, "WRITE_SIZE": B * (nq + nv) * 8}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = [float(r["Counter_Value"]) for f in glob.glob(f"{out}/cal_{c}/**/*counter_collection.csv", recursive=True)
            for r in csv.DictReader(open(f)) if "integrate_soa" in r["Kernel_Name"] and r["Counter_Name"] == c]
    if vals:
        m = sum(vals) / len(vals)
        print(f"{c}: counter {m:.1f} KB per launch over {len(vals)} launches; true {true[c] / 1024:.1f} KB; counter / true = {m * 1024 / true[c]:.4f}")
PY
