#!/bin/bash
# usage (GPU box, repo root): tools/pmc_c5_pair.sh [B]   FETCH_SIZE / WRITE_SIZE (separate passes) and kernel times of config 5's pair call
# (fp32, AoS and SoA, fused walk and two launches); summary on stdout
root=$(pwd); B=${1:-131072}; out=$root/gpurun_out/pmc_c5pair; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for pair in 1 0; do for lay in aos soa; do
  export MH_DFS_PAIR=$pair
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d $out/p${pair}_${lay}_$c -o pmc --output-format csv -- python3 $root/tools/pmc_c5_pair.py $B $lay > $out/p${pair}_${lay}_$c.log 2>&1
  done
  rocprofv3 --kernel-trace --stats -d $out/p${pair}_${lay}_trace -o t --output-format csv -- python3 $root/tools/pmc_c5_pair.py $B $lay > $out/p${pair}_${lay}_trace.log 2>&1
  echo "done pair=$pair $lay"
done; done
cd $root
python3 - "$out" "$B" <<'PY'
import csv, glob, sys, collections
out, B = sys.argv[1], int(sys.argv[2])
nq, nv = 362, 323
alg = B * 4 * 2 * (nq + 3 * nv)  # the pair: RNEA + ABA, inputs read once + outputs written once each
print(f"# config 5 pair call, B = {B}, algorithmic bytes per call {alg / 1e6:.1f} MB (2 x 4 (nq + 3 nv) per configuration, nq {nq}, nv {nv})")
print("# HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH_SIZE doubled on gfx950: MI355X_MICROARCH.md), per call = sum over the call's kernels")
for pair in (1, 0):
    for lay in ("aos", "soa"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(f"{out}/p{pair}_{lay}_{c}/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        times = {}
        for f in glob.glob(f"{out}/p{pair}_{lay}_trace/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                times[r["Name"][:60]] = (float(r["AverageNs"]), int(r["Calls"]))
        total_b, total_t = 0.0, 0.0
        print(f"## {'fused walk' if pair else 'two launches'}, {lay.upper()}")
        for k, cs in sorted(acc.items()):
            if not any(s in k for s in ("dfs", "rows_to", "columns_to", "transpose")):
                continue
            n = 5.0
            fb = 2 * sum(cs["FETCH_SIZE"]) * 1024 / n
            wb = sum(cs["WRITE_SIZE"]) * 1024 / n
            t = times.get(k, (0.0, 0))
            tt = t[0] * t[1] / n
            total_b += fb + wb
            total_t += tt
            print(f"   {k:60s} per call: {(fb + wb) / 1e6:9.1f} MB HBM ({fb / 1e6:.1f} read + {wb / 1e6:.1f} written), {tt / 1e3:9.1f} us")
        print(f"   TOTAL per call {total_b / 1e6:.1f} MB = {total_b / alg:.2f} x algorithmic, {total_t / 1e3:.1f} us of kernels -> {B / total_t * 1e3:.1f} M configs/s")
PY
